"""Adaptor for the seam where a native gridder drops in "behind the 3-D grid cache tier" of
``radar_processor.process_radar_to_cog`` -- ``_get_or_build_grid3d`` (``src/radar_processor/processor.py:33-181``).

That function grids with PyART (``grid_from_radars(..., 'map_gates_to_grid', weighting_function='nearest',
roi_func='constant', constant_roi=...)``, processor.py:152-163) and stores a plain dict in ``GRID3D_CACHE``
(processor.py:170-179).  :func:`build_grid3d_package` produces that dict with the fused HIP gridder
(``rg_roi_grid_f32``, closest-gate selection, constant ROI), so a maintainer replaces the PyART call by

    pkg = build_grid3d_package(radar_to_use, field_to_use, z_grid_limits, y_grid_limits, x_grid_limits,
                               grid_resolution, gate_excluded=gf.gate_excluded)
    GRID3D_CACHE[cache_key] = pkg

**Parity unpinned**: PyART (arm-pyart >= 2.1.1) is not in the reference tree and not installable here, so the
selection rule is restated from its documented behaviour (the value of the closest non-excluded gate within the
constant ROI wins) and checked only against this build's own brute-force oracle.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import _native
from .geometry_builder import RoiSearch
from .radar_adaptors import get_gate_coordinates
from .roi_grid import roi_grid_fields_device


def constant_roi_for(grid_resolution: float, y_grid_limits: Sequence[float]) -> float:
    """``max(1.5 * res, 800 + 400 * range_max / 100 km)`` (processor.py:134-139)."""
    range_max_m = (y_grid_limits[1] - y_grid_limits[0]) / 2
    return max(grid_resolution * 1.5, 800 + (range_max_m / 100000) * 400)


def grid3d_shape(z_grid_limits, y_grid_limits, x_grid_limits, grid_resolution: float):
    """``(ceil(z_top / res) + 1, int(dy / res), int(dx / res))`` (processor.py:141-143)."""
    return (int(np.ceil(z_grid_limits[1] / grid_resolution)) + 1,
            int((y_grid_limits[1] - y_grid_limits[0]) / grid_resolution),
            int((x_grid_limits[1] - x_grid_limits[0]) / grid_resolution))


def build_grid3d_package(radar, field_name: str, z_grid_limits, y_grid_limits, x_grid_limits, grid_resolution: float,
                         gate_excluded: Optional[np.ndarray] = None, device=None) -> dict:
    """Grid ``field_name`` of a (duck-typed) PyART radar the way ``_get_or_build_grid3d`` does and return the
    cache package: ``arr3d`` (masked float32 ``[nz, ny, nx]``), ``x`` / ``y`` / ``z`` axes, ``projection``,
    ``field_name``, ``field_metadata``.  ``gate_excluded``: the QC GateFilter's boolean mask (True = drop)."""
    torch = _native.torch_mod()
    shape = grid3d_shape(z_grid_limits, y_grid_limits, x_grid_limits, grid_resolution)
    limits = (tuple(z_grid_limits), tuple(y_grid_limits), tuple(x_grid_limits))
    roi = constant_roi_for(grid_resolution, y_grid_limits)
    gx, gy, gz = get_gate_coordinates(radar)
    search = RoiSearch(gx, gy, gz, shape, limits, min_radius=roi, beam_factor=0.0, toa=float("inf"), device=device)
    raw = radar.fields[field_name]["data"]
    masked = np.ma.masked_invalid(raw)
    values = np.ascontiguousarray(np.ma.getdata(masked), dtype=np.float32).ravel()
    mask = np.ma.getmaskarray(masked).ravel()
    if gate_excluded is not None:
        mask = mask | np.asarray(gate_excluded, dtype=bool).ravel()
    f_t = torch.from_numpy(values).to(search.dev)
    m_t = torch.from_numpy(mask.astype(np.uint8)).to(search.dev)
    arr = roi_grid_fields_device(search, [f_t], [m_t], weighting="closest")[0].cpu().numpy()
    lat = float(radar.latitude["data"][0])
    lon = float(radar.longitude["data"][0])
    return {
        "arr3d": np.ma.masked_invalid(arr),
        "x": np.linspace(x_grid_limits[0], x_grid_limits[1], shape[2]),
        "y": np.linspace(y_grid_limits[0], y_grid_limits[1], shape[1]),
        "z": np.linspace(z_grid_limits[0], z_grid_limits[1], shape[0]),
        "projection": {"proj": "pyart_aeqd", "lat_0": lat, "lon_0": lon, "_include_lon_0_lat_0": True},
        "field_name": field_name,
        "field_metadata": {k: v for k, v in radar.fields[field_name].items() if k != "data"},
    }
