"""Gate QC filters -- host-side mirror of ``radar_grid/filters.py:12-598`` (``GateFilter``,
``create_mask_from_filter``).

A filter is a mask *producer*: the hot path only consumes the resulting ``bool[n_gates]``
(``True`` = gate excluded), which is OR-ed with the field's own mask and folded into the packed field
values on the device (``rg_pack_fields_f32``).  The predicates are the reference's, including its NaN
behaviour: comparisons with NaN are False, so threshold filters never exclude NaN gates
(``filters.py:134``).  For volumes that already live in HBM, :func:`device_gate_mask` evaluates the same
predicates with ``rg_gate_mask_f32`` without a host round trip.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _native

logger = logging.getLogger("radar_grid.filters")


def _flat_f32(field) -> np.ndarray:
    """Raw float32 values of a 2-D radar field, flattened ray-major (``filters.py:99-102``)."""
    return np.ma.getdata(np.ma.masked_invalid(field)).ravel().astype("float32")


class GateFilter:
    """Accumulates exclusion predicates over the gates of one radar volume (OR logic).

    ``radar`` is duck-typed like ``pyart.core.Radar``: ``nrays``, ``ngates``, ``fields[name]['data']``,
    and for the geometric predicates ``gate_altitude``, ``range``, ``elevation``.
    """

    def __init__(self, radar):
        self.radar = radar
        self.n_gates = radar.nrays * radar.ngates
        self._gate_excluded = np.zeros(self.n_gates, dtype=bool)
        self._filter_history: List[str] = []

    # ---- state -----------------------------------------------------------------------------------
    @property
    def gate_excluded(self) -> np.ndarray:
        return self._gate_excluded

    @property
    def gate_included(self) -> np.ndarray:
        return ~self._gate_excluded

    def n_excluded(self) -> int:
        return self._gate_excluded.sum()

    def n_included(self) -> int:
        return (~self._gate_excluded).sum()

    def summary(self) -> str:
        n_ex, n_in = self.n_excluded(), self.n_included()
        head = [
            "GateFilter Summary:",
            f"  Total gates: {self.n_gates:,}",
            f"  Excluded: {n_ex:,} ({100 * n_ex / self.n_gates:.1f}%)",
            f"  Included: {n_in:,} ({100 * n_in / self.n_gates:.1f}%)",
            f"  Filters applied ({len(self._filter_history)}):",
        ]
        return "\n".join(head + [f"    - {h}" for h in self._filter_history])

    def __repr__(self) -> str:
        return f"GateFilter(excluded={self.n_excluded():,}/{self.n_gates:,}, filters={len(self._filter_history)})"

    # ---- plumbing --------------------------------------------------------------------------------
    def _get_field_data(self, field_name: str) -> np.ndarray:
        return _flat_f32(self.radar.fields[field_name]["data"])

    def _add_filter(self, mask: np.ndarray, description: str) -> "GateFilter":
        self._gate_excluded = self._gate_excluded | mask
        self._filter_history.append(description)
        return self

    def _field_or_warn(self, field_name: str) -> Optional[np.ndarray]:
        """Missing field => warning + no-op (``filters.py:130-132``)."""
        if field_name not in self.radar.fields:
            logger.warning(f"Field '{field_name}' not found in radar. No gates excluded.")
            return None
        return self._get_field_data(field_name)

    # ---- threshold predicates (filters.py:114-233) -------------------------------------------------
    def exclude_below(self, field_name: str, threshold: float) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter(d < threshold, f"{field_name} < {threshold}")

    def exclude_above(self, field_name: str, threshold: float) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter(d > threshold, f"{field_name} > {threshold}")

    def exclude_between(self, field_name: str, low: float, high: float) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter((d > low) & (d < high), f"{low} < {field_name} < {high}")

    def exclude_outside(self, field_name: str, low: float, high: float) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter((d < low) | (d > high), f"{field_name} outside [{low}, {high}]")

    def exclude_equal(self, field_name: str, value: float, atol: float = 1e-5) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter(np.abs(d - value) < atol, f"{field_name} == {value}")

    # ---- invalid-data predicates (filters.py:239-306) ----------------------------------------------
    def exclude_invalid(self, field_name: str) -> "GateFilter":
        d = self._field_or_warn(field_name)
        return self if d is None else self._add_filter(np.isnan(d) | np.isinf(d), f"{field_name} invalid (NaN/Inf)")

    def exclude_masked(self, field_name: str) -> "GateFilter":
        if self._field_or_warn(field_name) is None:
            return self
        field = self.radar.fields[field_name]["data"]
        if isinstance(field, np.ma.MaskedArray):
            mask = np.ma.getmaskarray(field).ravel()
        else:
            mask = np.zeros(self.n_gates, dtype=bool)
        return self._add_filter(mask, f"{field_name} masked")

    def exclude_all_invalid(self, field_name: str) -> "GateFilter":
        if self._field_or_warn(field_name) is None:
            return self
        field = self.radar.fields[field_name]["data"]
        mask = np.ma.getmaskarray(np.ma.masked_invalid(field)).ravel()
        return self._add_filter(mask, f"{field_name} all invalid (NaN/Inf/masked)")

    # ---- geometric predicates (filters.py:312-469) -------------------------------------------------
    def _altitude(self) -> np.ndarray:
        return self.radar.gate_altitude["data"].ravel()

    def _range_per_gate(self) -> np.ndarray:
        ranges = self.radar.range["data"]
        return np.broadcast_to(ranges, (self.radar.nrays, self.radar.ngates)).ravel()

    def _elevation_per_gate(self) -> np.ndarray:
        return np.repeat(self.radar.elevation["data"], self.radar.ngates)

    def exclude_below_altitude(self, altitude: float) -> "GateFilter":
        return self._add_filter(self._altitude() < altitude, f"altitude < {altitude}m")

    def exclude_above_altitude(self, altitude: float) -> "GateFilter":
        return self._add_filter(self._altitude() > altitude, f"altitude > {altitude}m")

    def exclude_below_range(self, range_min: float) -> "GateFilter":
        return self._add_filter(self._range_per_gate() < range_min, f"range < {range_min}m")

    def exclude_above_range(self, range_max: float) -> "GateFilter":
        return self._add_filter(self._range_per_gate() > range_max, f"range > {range_max}m")

    def exclude_below_elevation_angle(self, min_elev: float) -> "GateFilter":
        return self._add_filter(self._elevation_per_gate() < min_elev, f"elevation angle < {min_elev}°")

    def exclude_above_elevation_angle(self, max_elev: float) -> "GateFilter":
        return self._add_filter(self._elevation_per_gate() > max_elev, f"elevation angle > {max_elev}°")

    def exclude_outside_elevation_range(self, min_elev: float, max_elev: float) -> "GateFilter":
        e = self._elevation_per_gate()
        return self._add_filter((e < min_elev) | (e > max_elev), f"elevation angle outside [{min_elev}°, {max_elev}°]")

    # ---- custom predicates (filters.py:475-530) ----------------------------------------------------
    def exclude_where(self, mask: np.ndarray, description: str = "custom") -> "GateFilter":
        flat = mask.ravel().astype(bool)
        if len(flat) != self.n_gates:
            raise ValueError(f"Mask size {len(flat)} doesn't match n_gates {self.n_gates}")
        return self._add_filter(flat, description)

    def exclude_by_function(self, field_name: str, func: Callable[[np.ndarray], np.ndarray],
                            description: str = "custom function") -> "GateFilter":
        return self._add_filter(func(self._get_field_data(field_name)), f"{field_name}: {description}")

    # ---- utilities (filters.py:536-557) ------------------------------------------------------------
    def copy(self) -> "GateFilter":
        dup = GateFilter(self.radar)
        dup._gate_excluded = self._gate_excluded.copy()
        dup._filter_history = self._filter_history.copy()
        return dup

    def reset(self) -> "GateFilter":
        self._gate_excluded = np.zeros(self.n_gates, dtype=bool)
        self._filter_history = []
        return self

    def include_all(self) -> "GateFilter":
        return self.reset()

    def exclude_all(self) -> "GateFilter":
        self._gate_excluded = np.ones(self.n_gates, dtype=bool)
        self._filter_history.append("exclude all")
        return self


def create_mask_from_filter(radar, field_name: str,
                            gatefilter: Optional[GateFilter] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Flattened float32 values plus the combined (invalid | filter) mask (``filters.py:560-598``)."""
    masked = np.ma.masked_invalid(radar.fields[field_name]["data"])
    values = np.ma.getdata(masked).ravel().astype("float32")
    mask = np.ma.getmaskarray(masked).ravel()
    if gatefilter is not None:
        mask = mask | gatefilter.gate_excluded
    return values, mask


def device_gate_mask(data, op: str, a: float = 0.0, b: float = 0.0, mask=None):
    """Evaluate one GateFilter predicate on a device-resident float32 field (``rg_gate_mask_f32``).

    ``data``: cuda float32 tensor ``[G]``; ``mask``: optional uint8 tensor to OR into (allocated when
    omitted).  ``op`` is one of below/above/between/outside/equal/invalid.  Returns the uint8 mask tensor.
    """
    torch = _native.torch_mod()
    lib = _native.load_library()
    if op not in _native.GATE_OPS:
        raise ValueError(f"unknown gate predicate: {op}")
    if not (data.is_cuda and data.dtype == torch.float32 and data.is_contiguous()):
        raise ValueError("data must be a contiguous cuda float32 tensor")
    if mask is None:
        mask = torch.zeros(data.numel(), dtype=torch.uint8, device=data.device)
    with torch.cuda.device(data.device):
        _native.check(lib.rg_gate_mask_f32(_native.ptr(data), data.numel(), _native.GATE_OPS[op], float(a), float(b),
                                           _native.ptr(mask), _native.stream_ptr()), "rg_gate_mask_f32")
    return mask


# ============================================================================
# GridFilter -- applied after interpolation on 2-D projections (radar_grid/filters.py:609-780)
# ============================================================================
class GridFilter:
    """Value filters on a product plane (``radar_grid/filters.py:609-780``, same methods): pixels below / above a
    threshold, outside a range, NaN / Inf, or flagged by a custom function are set to ``fill_value`` (NaN by
    default); the input is never modified.  NumPy arrays (float32 or float64, any shape) are staged through HBM and
    come back as NumPy; cuda tensors stay on the device.  The comparison runs in the array's dtype, exactly as
    NumPy compares an array with a Python number (``rg_grid_filter``)."""

    @staticmethod
    def _run(grid, flags: int, lo: float = 0.0, hi: float = 0.0, mask=None, fill_value: float = np.nan):
        torch = _native.torch_mod()
        lib = _native.load_library()
        is_tensor = type(grid).__module__.startswith("torch")
        if is_tensor:
            if not grid.is_cuda:
                raise _native.NativeUnavailable("GridFilter runs on the GPU: pass a cuda tensor or a NumPy array")
            src = grid.contiguous()
        else:
            arr = np.asarray(grid)
            if arr.dtype not in (np.float32, np.float64):
                arr = arr.astype(np.float64)        # the reference would fail to store NaN in an integer grid
            src = torch.from_numpy(np.ascontiguousarray(arr)).to(_native.device())
        if src.dtype not in (torch.float32, torch.float64):
            raise ValueError("grid must be float32 or float64")
        mask_t = None
        if mask is not None:
            if type(mask).__module__.startswith("torch"):
                mask_t = mask.to(device=src.device, dtype=torch.uint8).contiguous()
            else:
                mask_t = torch.from_numpy(np.ascontiguousarray(np.asarray(mask, dtype=bool).astype(np.uint8))).to(src.device)
            if mask_t.numel() != src.numel():
                raise ValueError("the custom mask must have the shape of the grid")
        out = torch.empty_like(src)
        with torch.cuda.device(src.device):
            _native.check(lib.rg_grid_filter(_native.ptr(src), int(src.dtype == torch.float64), src.numel(), flags,
                                             float(lo), float(hi), _native.ptr(mask_t), float(fill_value),
                                             _native.ptr(out), _native.stream_ptr()), "rg_grid_filter")
        return out if is_tensor else out.cpu().numpy()

    def apply_below(self, grid, threshold: float, fill_value: float = np.nan):
        """Set values ``< threshold`` to ``fill_value`` (``filters.py:631-658``)."""
        return self._run(grid, _native.RG_TEST_LO, lo=threshold, fill_value=fill_value)

    def apply_above(self, grid, threshold: float, fill_value: float = np.nan):
        """Set values ``> threshold`` to ``fill_value`` (``filters.py:660-687``)."""
        return self._run(grid, _native.RG_TEST_HI, hi=threshold, fill_value=fill_value)

    def apply_outside_range(self, grid, vmin: float, vmax: float, fill_value: float = np.nan):
        """Set values outside ``[vmin, vmax]`` to ``fill_value`` (``filters.py:689-720``)."""
        return self._run(grid, _native.RG_TEST_LO | _native.RG_TEST_HI, lo=vmin, hi=vmax, fill_value=fill_value)

    def apply_invalid(self, grid, fill_value: float = np.nan):
        """Set NaN and Inf values to ``fill_value`` (``filters.py:722-748``)."""
        return self._run(grid, _native.RG_TEST_NONFINITE, fill_value=fill_value)

    def apply_custom(self, grid, func: Callable, fill_value: float = np.nan):
        """``func(grid)`` returns a boolean mask, True = set to ``fill_value`` (``filters.py:750-780``).  The
        function itself is the caller's code and runs wherever ``grid`` lives; only the masked assignment is ours."""
        if type(grid).__module__.startswith("torch"):
            return self._run(grid, 0, mask=func(grid.clone()), fill_value=fill_value)
        return self._run(grid, 0, mask=func(np.array(grid, copy=True)), fill_value=fill_value)
