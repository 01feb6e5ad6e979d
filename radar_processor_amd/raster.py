"""The 2-D raster stage behind the 3-D grid cache, on the device:

* :func:`collapse_field_3d_to_2d` / :func:`collapse_grid_to_2d` -- ``radar_processor/utils.py:336-387`` and
  ``radar_processor/processor.py:480-551``: 'ppi' (nearest level to the beam height), 'cappi' (nearest level to a
  height), 'colmax', plus the per-field re-mask against ``vmin``;
* :func:`apply_filter_masks` -- ``radar_processor/processor.py:802-886`` (visual + QC threshold filters);
* :func:`apply_colormap_to_array` -- ``radar_grid/geotiff.py:70-145`` (matplotlib Normalize + colormap -> RGBA uint8).

Same names and argument meaning as the reference.  NumPy / masked-array inputs are staged to HBM and come back as
NumPy; the ``*_device`` functions take and return cuda tensors, where **NaN stands for "masked"** (the cache package
holds ``masked_invalid`` data, so the two coincide).  One deliberate consequence: an *unmasked* NaN inside a plain
ndarray is ignored by 'colmax' here, while ``ndarray.max`` would propagate it.

Nothing of this is pinned by golden vectors (``radar_processor`` and ``geotiff`` do not import in the build
container): parity rests on the reference's own test expectations and a restatement, see oracle/radar_grid_oracle.py.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence

import numpy as np

from . import _native

PROCESSOR_EARTH_RADIUS = 8.49e6   # processor.py:517 / utils.py:369
REFLECTIVITY_FIELDS = ("filled_DBZH", "DBZH", "DBZV", "DBZHF", "composite_reflectivity")   # processor.py:543
STRICT_FIELDS = ("KDP", "ZDR")                                                              # processor.py:545


class PlaneTest(NamedTuple):
    """One threshold test of ``rg_plane_filter_f32``: drop a pixel when the tested plane (``None`` = the source plane
    itself) is ``< lo`` (``<= lo`` with ``lo_inclusive``), ``> hi``, or -- with ``nonfinite`` -- NaN / +-inf."""
    plane: object = None
    lo: Optional[float] = None
    hi: Optional[float] = None
    lo_inclusive: bool = False
    nonfinite: bool = False


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch")


def _require_cuda(t, what: str):
    if not t.is_cuda:
        raise _native.NativeUnavailable(f"{what} runs on the GPU: pass a cuda tensor or a NumPy array")


def _nan_filled_f32(a) -> np.ndarray:
    """float32 copy of a (masked) array with NaN where it is masked."""
    if isinstance(a, np.ma.MaskedArray):
        data = np.array(np.ma.getdata(a), dtype=np.float32)   # a copy: the caller's array is never written
        mask = np.ma.getmaskarray(a)
        if mask.any():
            data[mask] = np.nan
        return np.ascontiguousarray(data)
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------------------------------
# collapse
# --------------------------------------------------------------------------------------------------
def collapse_plane_device(grid, product: str, *, x_coords=None, y_coords=None, z_levels=None, elevation_deg=None,
                          target_height_m=None, return_level: bool = False):
    """``grid``: cuda float32 ``[nz, ny, nx]`` (NaN = masked) -> cuda float32 ``[ny, nx]``.  With
    ``return_level`` the 'ppi' product also returns the int32 level index it picked per pixel."""
    torch = _native.torch_mod()
    lib = _native.load_library()
    _require_cuda(grid, "collapse")
    if grid.dtype != torch.float32 or grid.dim() != 3:
        raise ValueError("device grids must be float32 [nz, ny, nx]")
    grid = grid.contiguous()
    nz, ny, nx = (int(s) for s in grid.shape)
    out = torch.empty((ny, nx), dtype=torch.float32, device=grid.device)
    with torch.cuda.device(grid.device):
        stream = _native.stream_ptr()
        if product == "ppi":
            assert elevation_deg is not None and x_coords is not None and y_coords is not None and z_levels is not None
            tables = []
            for coords, n in ((x_coords, nx), (y_coords, ny), (z_levels, nz)):
                host = np.ascontiguousarray(np.asarray(coords, dtype=np.float64))
                if host.shape != (n,):
                    raise ValueError(f"coordinate table of length {host.shape} does not match the grid ({n})")
                tables.append(torch.from_numpy(host).to(grid.device))
            level = torch.empty((ny, nx), dtype=torch.int32, device=grid.device) if return_level else None
            sin_elev = float(np.sin(np.deg2rad(elevation_deg)))                    # processor.py:520
            _native.check(lib.rg_collapse_ppi_f32(_native.ptr(grid), _native.ptr(tables[0]), _native.ptr(tables[1]),
                                                  _native.ptr(tables[2]), nz, ny, nx, sin_elev,
                                                  2.0 * PROCESSOR_EARTH_RADIUS, _native.ptr(out), _native.ptr(level),
                                                  stream), "rg_collapse_ppi_f32")
            return (out, level) if return_level else out
        if product == "cappi":
            assert target_height_m is not None and z_levels is not None
            iz = int(np.abs(np.asarray(z_levels) - float(target_height_m)).argmin())   # processor.py:532
            lo = hi = iz
        elif product == "colmax":
            lo, hi = 0, nz - 1
        else:
            raise ValueError("Producto inválido")
        _native.check(lib.rg_column_reduce_f32(_native.ptr(grid), nz, ny * nx, lo, hi, _native.COLUMN_OPS["max"],
                                               _native.ptr(out), 0, stream), "rg_column_reduce_f32")
    return out


def collapse_field_3d_to_2d(data3d, product: str, *, x_coords=None, y_coords=None, z_levels=None, elevation_deg=None,
                            target_height_m=None):
    """Collapse a 3-D field to the 2-D product plane without touching a Grid object
    (``radar_processor/utils.py:336-387``, same keywords).  NumPy / masked input -> float32 masked array;
    cuda tensor -> cuda tensor with NaN where masked."""
    if _is_tensor(data3d):
        if data3d.dim() == 2:
            return data3d
        return collapse_plane_device(data3d, product, x_coords=x_coords, y_coords=y_coords, z_levels=z_levels,
                                     elevation_deg=elevation_deg, target_height_m=target_height_m)
    if data3d.ndim == 2:
        return np.ma.array(np.asarray(np.ma.getdata(data3d)).astype(np.float32), mask=np.ma.getmaskarray(data3d))
    if product not in ("ppi", "cappi", "colmax"):
        raise ValueError("Producto inválido")
    torch = _native.torch_mod()
    dev = _native.device()
    grid = torch.from_numpy(_nan_filled_f32(data3d)).to(dev)
    plane = collapse_plane_device(grid, product, x_coords=x_coords, y_coords=y_coords, z_levels=z_levels,
                                  elevation_deg=elevation_deg, target_height_m=target_height_m).cpu().numpy()
    if isinstance(data3d, np.ma.MaskedArray):
        return np.ma.array(plane, mask=np.isnan(plane))
    return np.ma.array(plane, mask=np.zeros(plane.shape, dtype=bool))


def remask_tests(field: str, vmin: float):
    """The re-mask of collapse_grid_to_2d as plane tests (processor.py:541-546): invalid values always, ``<= vmin``
    for the reflectivity fields, ``< vmin`` for KDP / ZDR."""
    if field in REFLECTIVITY_FIELDS:
        return [PlaneTest(lo=float(vmin), lo_inclusive=True, nonfinite=True)]      # masked_less_equal
    if field in STRICT_FIELDS:
        return [PlaneTest(lo=float(vmin), nonfinite=True)]                         # masked_less
    return [PlaneTest(nonfinite=True)]


def collapse_grid_to_2d(grid, field: str, product: str, *, elevation_deg=None, target_height_m=None, vmin=-30.0):
    """In-place collapse of a (duck-typed) ``pyart.core.Grid`` to a single level
    (``radar_processor/processor.py:480-551``): collapse, mask invalid values, mask ``<= vmin`` for the
    reflectivity fields / ``< vmin`` for KDP and ZDR, store as ``(1, ny, nx)`` with ``_FillValue = -9999`` and
    ``z = [0.0]``."""
    data3d = grid.fields[field]["data"]
    if data3d.ndim != 2 and product not in ("ppi", "cappi", "colmax"):
        raise ValueError("Producto inválido")
    torch = _native.torch_mod()
    dev = _native.device()
    plane_t = torch.from_numpy(_nan_filled_f32(data3d)).to(dev)
    if data3d.ndim != 2:
        plane_t = collapse_plane_device(plane_t, product, x_coords=grid.x["data"], y_coords=grid.y["data"],
                                        z_levels=grid.z["data"], elevation_deg=elevation_deg,
                                        target_height_m=target_height_m)
    _, mask_t = plane_filter_device(plane_t, remask_tests(field, vmin), want_values=False, want_mask=True)
    plane = np.ma.array(plane_t.cpu().numpy(), mask=mask_t.cpu().numpy().astype(bool))
    grid.fields[field]["data"] = plane[np.newaxis, ...]
    grid.fields[field]["_FillValue"] = -9999.0
    grid.z["data"] = np.array([0.0], dtype=float)


# --------------------------------------------------------------------------------------------------
# threshold masks
# --------------------------------------------------------------------------------------------------
def plane_filter_device(src, tests: Sequence[PlaneTest], src_mask=None, want_values: bool = True, want_mask: bool = False):
    """OR of threshold tests over a cuda float32 plane (``rg_plane_filter_f32``).

    ``tests``: :class:`PlaneTest` entries.  Returns ``(values, mask)``: values with NaN where dropped or already
    masked, mask uint8 (either may be ``None``).  "Already masked" means ``src_mask != 0`` when a mask is given and
    ``isnan(src)`` otherwise."""
    torch = _native.torch_mod()
    lib = _native.load_library()
    _require_cuda(src, "plane filters")
    if src.dtype != torch.float32:
        raise ValueError("planes must be float32")
    src = src.contiguous()
    n = src.numel()
    if len(tests) > _native.RG_MAX_PLANE_TESTS:
        raise ValueError(f"at most {_native.RG_MAX_PLANE_TESTS} tests per pass")
    arr = (_native.PlaneTest * max(len(tests), 1))()
    keep = []   # keeps contiguous copies of the tested planes alive until the launch is enqueued
    for i, test in enumerate(tests):
        plane, lo, hi = test.plane, test.lo, test.hi
        flags = _native.RG_TEST_NONFINITE if test.nonfinite else 0
        if lo is not None:
            flags |= _native.RG_TEST_LO | (_native.RG_TEST_LO_INCLUSIVE if test.lo_inclusive else 0)
        if hi is not None:
            flags |= _native.RG_TEST_HI
        if plane is not None:
            _require_cuda(plane, "plane filters")
            if plane.dtype != torch.float32 or plane.numel() != n:
                raise ValueError("a tested plane must be float32 with the shape of the source plane")
            plane = plane.contiguous()
            keep.append(plane)
        arr[i].plane = _native.ptr(plane)
        arr[i].lo = 0.0 if lo is None else float(lo)
        arr[i].hi = 0.0 if hi is None else float(hi)
        arr[i].flags = flags
    if src_mask is not None:
        src_mask = src_mask.to(torch.uint8).contiguous()
    out = torch.empty_like(src) if want_values else None
    out_mask = torch.empty(src.shape, dtype=torch.uint8, device=src.device) if want_mask else None
    with torch.cuda.device(src.device):
        _native.check(lib.rg_plane_filter_f32(_native.ptr(src), _native.ptr(src_mask), n, arr, len(tests),
                                              _native.ptr(out), _native.ptr(out_mask), _native.stream_ptr()),
                      "rg_plane_filter_f32")
    return out, out_mask


def _filter_tests(visual_filters, qc_filters, field_to_use, qc_planes, upload):
    """Turn the reference's filter objects into plane tests (processor.py:836-884).  ``qc_planes``: dict of 2-D QC
    planes; ``upload(plane)`` -> cuda tensor."""
    tests = []
    cache = {}

    def dev_plane(name):
        if name not in cache:
            cache[name] = upload(qc_planes[name])
        return cache[name]

    for flt in visual_filters or ():
        name = str(getattr(flt, "field", None) or "").upper()
        if not name:
            continue
        lo, hi = getattr(flt, "min", None), getattr(flt, "max", None)
        if name == str(field_to_use).upper():
            if lo is not None and lo <= 0.3 and field_to_use == "RHOHV":   # processor.py:849
                lo = None
            if lo is not None or hi is not None:
                tests.append(PlaneTest(None, lo, hi))
        elif qc_planes.get(name) is not None:
            if lo is not None or hi is not None:
                tests.append(PlaneTest(dev_plane(name), lo, hi))
    for flt in qc_filters or ():
        name = str(getattr(flt, "field", "") or "").upper()
        if qc_planes.get(name) is None:
            continue
        lo, hi = getattr(flt, "min", None), getattr(flt, "max", None)
        if lo is not None or hi is not None:
            tests.append(PlaneTest(dev_plane(name), lo, hi))
    return tests


def apply_filter_masks(masked_arr, visual_filters, qc_filters, field_to_use, pkg_cached):
    """Phases 10 and 11 of ``process_radar_to_cog`` (``radar_processor/processor.py:802-886``, there
    ``_apply_filter_masks``): mask the plane where a visual filter on the plotted field, a cross-field visual filter
    or a QC filter on a cached QC plane (``pkg_cached['qc']``) fails.  The data values are left untouched, only the
    mask grows; without filters the input is returned as is."""
    if not visual_filters and not qc_filters:
        return masked_arr
    torch = _native.torch_mod()
    dev = _native.device()
    qc_planes = pkg_cached.get("qc", {}) or {}

    def upload(plane):
        # the reference compares the raw data of a masked QC plane (its mask does not take part)
        return torch.from_numpy(np.ascontiguousarray(np.ma.getdata(plane), dtype=np.float32)).to(dev)

    data = np.ma.getdata(masked_arr)
    src = upload(masked_arr)
    tests = _filter_tests(visual_filters, qc_filters, field_to_use, qc_planes, upload)
    mask_in = torch.from_numpy(np.ma.getmaskarray(masked_arr).astype(np.uint8)).to(dev)
    out_mask = None
    for i in range(0, max(len(tests), 1), _native.RG_MAX_PLANE_TESTS):
        _, out_mask = plane_filter_device(src, tests[i:i + _native.RG_MAX_PLANE_TESTS],
                                          src_mask=mask_in if out_mask is None else out_mask,
                                          want_values=False, want_mask=True)
    return np.ma.array(np.array(data, copy=True), mask=out_mask.cpu().numpy().astype(bool))


# --------------------------------------------------------------------------------------------------
# colormap -> RGBA
# --------------------------------------------------------------------------------------------------
def colormap_lut(cmap) -> np.ndarray:
    """uint8 ``[N + 3, 4]`` table: the colormap's N entries, then its under, over and bad colours, each
    ``(rgba * 255)`` truncated like ``geotiff.py:139``.  ``cmap``: matplotlib name or Colormap, or an ``(N, 4)`` /
    ``(N, 3)`` float array of colours in 0..1 (under = first, over = last, bad = transparent black -- matplotlib's
    defaults for a listed colormap)."""
    if isinstance(cmap, (str, bytes)) or hasattr(cmap, "get_bad"):
        if isinstance(cmap, (str, bytes)):
            import matplotlib.pyplot as plt
            cmap = plt.get_cmap(cmap)
        n = int(cmap.N)
        table = np.vstack([np.asarray(cmap(np.arange(n)), dtype=np.float64).reshape(n, 4),
                           np.asarray(cmap.get_under(), dtype=np.float64).reshape(1, 4),
                           np.asarray(cmap.get_over(), dtype=np.float64).reshape(1, 4),
                           np.asarray(cmap.get_bad(), dtype=np.float64).reshape(1, 4)])
    else:
        colours = np.asarray(cmap, dtype=np.float64)
        if colours.ndim != 2 or colours.shape[1] not in (3, 4) or len(colours) < 1:
            raise ValueError("a colour table must have shape (N, 3) or (N, 4)")
        if colours.shape[1] == 3:
            colours = np.hstack([colours, np.ones((len(colours), 1))])
        table = np.vstack([colours, colours[:1], colours[-1:], np.zeros((1, 4))])
    if len(table) - 3 > _native.RG_MAX_LUT:
        raise ValueError(f"colormaps of more than {_native.RG_MAX_LUT} entries are not supported")
    return np.ascontiguousarray((table * 255).astype(np.uint8))


def colormap_rgba_device(data, lut, vmin=None, vmax=None, fill_value=None):
    """cuda float32 / float64 tensor of any shape -> cuda uint8 tensor ``shape + (4,)``.  ``lut``: the table of
    :func:`colormap_lut` (NumPy) or the same already on the device."""
    torch = _native.torch_mod()
    lib = _native.load_library()
    _require_cuda(data, "apply_colormap_to_array")
    if data.dtype not in (torch.float32, torch.float64):
        raise ValueError("data must be float32 or float64")
    data = data.contiguous()
    is_f64 = data.dtype == torch.float64
    n = data.numel()
    if not _is_tensor(lut):
        lut = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.uint8)).to(data.device)
    n_lut = int(lut.shape[0]) - 3
    has_fill = fill_value is not None
    fill = float(fill_value) if has_fill else 0.0
    with torch.cuda.device(data.device):
        stream = _native.stream_ptr()
        if vmin is None or vmax is None:
            ws = torch.empty(_native.RG_MINMAX_WORKSPACE_BYTES, dtype=torch.uint8, device=data.device)
            stats = torch.empty(4, dtype=torch.float64, device=data.device)
            _native.check(lib.rg_nan_minmax(_native.ptr(data), int(is_f64), n, int(has_fill), fill, _native.ptr(ws),
                                            _native.ptr(stats), stream), "rg_nan_minmax")
            lo, hi, count, kept = (float(v) for v in stats.cpu())
            # geotiff.py:118-130: nanmin / nanmax of the valid pixels (NaN when they are all NaN, as np.nanmin
            # returns), 0.0 / 1.0 when there is no valid pixel at all
            if vmin is None:
                vmin = 0.0 if kept == 0 else lo if count > 0 else float("nan")
            if vmax is None:
                vmax = 1.0 if kept == 0 else hi if count > 0 else float("nan")
        # Normalize keeps its limits as Python floats (matplotlib's _sanitize_extrema), whatever was passed in
        vmin, vmax = float(vmin), float(vmax)
        if vmin > vmax:
            raise ValueError("minvalue must be less than or equal to maxvalue")
        out = torch.empty(tuple(data.shape) + (4,), dtype=torch.uint8, device=data.device)
        _native.check(lib.rg_colormap_rgba(_native.ptr(data), int(is_f64), n, vmin, vmax, int(has_fill), fill,
                                           _native.ptr(lut), n_lut, _native.ptr(out), stream), "rg_colormap_rgba")
    return out


def apply_colormap_to_array(data, cmap, vmin: Optional[float] = None, vmax: Optional[float] = None,
                            fill_value: Optional[float] = None):
    """Convert a 2-D product to an RGBA uint8 image (``radar_grid/geotiff.py:70-145``, same signature): pixels equal
    to ``fill_value`` (NaN when it is ``None``) become transparent, the rest is normalised to ``[vmin, vmax]``
    (default: the valid pixels' min / max) with clipping and pushed through the colormap.  NumPy in -> NumPy out
    ``(ny, nx, 4)``; cuda tensor in -> cuda tensor out."""
    lut = colormap_lut(cmap)
    if _is_tensor(data):
        return colormap_rgba_device(data, lut, vmin, vmax, fill_value)
    torch = _native.torch_mod()
    dev = _native.device()
    if isinstance(data, np.ma.MaskedArray):
        data = data.astype(np.float64 if data.dtype == np.float64 else np.float32).filled(np.nan)
    arr = np.asarray(data)
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    return colormap_rgba_device(t, lut, vmin, vmax, fill_value).cpu().numpy()
