"""2-D products collapsed from a 3-D grid -- mirror of ``radar_grid/products.py``:
``constant_altitude_ppi`` (CAPPI, :317-415), ``column_max`` / ``column_min`` / ``column_mean`` (:420-580) and,
new in this build, ``column_argmax``.

The scalar control flow (level search, altitude -> index conversion, error handling) stays on the host exactly
as in the reference; the per-pixel arithmetic runs in ``rg_cappi_lerp_f32`` / ``rg_column_reduce_f32``
(csrc/rg_products.hip).  Inputs may be NumPy arrays (staged to HBM and back, NumPy result) or cuda float32
tensors (result stays in HBM).
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np

from . import _native
from .grid_geometry import GridGeometry

logger = logging.getLogger("radar_grid.products")

EARTH_RADIUS = 6371000.0            # radar_grid/products.py:19
EFFECTIVE_RADIUS_FACTOR = 4.0 / 3.0  # radar_grid/products.py:20


# --------------------------------------------------------------------------------------------------
# beam-height helpers (radar_grid/products.py:23-165) -- tiny host-side formulas, NumPy like the reference
# --------------------------------------------------------------------------------------------------
def compute_beam_height(horizontal_distance, elevation_angle: float, radar_altitude: float = 0.0,
                        ke: float = EFFECTIVE_RADIUS_FACTOR, re: float = EARTH_RADIUS):
    """Beam height above sea level with the 4/3 effective-earth-radius model
    ``h = sqrt(r^2 + (ke Re)^2 + 2 r ke Re sin(el)) - ke Re + h0`` where the slant range is approximated from the
    ground range as ``r = s / max(cos(el), 0.01)`` (``radar_grid/products.py:70-89``)."""
    el = np.radians(elevation_angle)
    r_eff = ke * re
    slant = horizontal_distance / np.maximum(np.cos(el), 0.01)
    return np.sqrt(slant**2 + r_eff**2 + 2 * slant * r_eff * np.sin(el)) - r_eff + radar_altitude


def compute_beam_height_simple(horizontal_distance, elevation_angle: float, radar_altitude: float = 0.0,
                               ke: float = EFFECTIVE_RADIUS_FACTOR, re: float = EARTH_RADIUS):
    """Second-order approximation ``h = r sin(el) + r^2 / (2 ke Re) + h0`` (``radar_grid/products.py:123-136``)."""
    el = np.radians(elevation_angle)
    r_eff = ke * re
    slant = horizontal_distance / np.maximum(np.cos(el), 0.01)
    return slant * np.sin(el) + (slant**2) / (2 * r_eff) + radar_altitude


def compute_beam_height_flat(horizontal_distance, elevation_angle: float, radar_altitude: float = 0.0):
    """Flat-earth beam height ``s tan(el) + h0`` (``radar_grid/products.py:164-165``)."""
    return horizontal_distance * np.tan(np.radians(elevation_angle)) + radar_altitude


def _ground_range(geometry: GridGeometry, dtype=None):
    """Horizontal distance of every (y, x) pixel from the radar at the grid origin."""
    _, ny, nx = geometry.grid_shape
    (y_lo, y_hi), (x_lo, x_hi) = geometry.grid_limits[1], geometry.grid_limits[2]
    yy, xx = np.meshgrid(np.linspace(y_lo, y_hi, ny, dtype=dtype), np.linspace(x_lo, x_hi, nx, dtype=dtype), indexing="ij")
    return np.sqrt(xx**2 + yy**2)


def get_beam_height_difference(geometry: GridGeometry, elevation_angle: float, radar_altitude: float = 0.0,
                               ke: float = EFFECTIVE_RADIUS_FACTOR):
    """Curved-minus-flat beam height on the grid's (y, x) plane, float64 ``(ny, nx)``
    (``radar_grid/products.py:583-626``)."""
    dist = _ground_range(geometry)
    return (compute_beam_height(dist, elevation_angle, radar_altitude, ke=ke)
            - compute_beam_height_flat(dist, elevation_angle, radar_altitude))


def get_elevation_from_z_level(z_level: float, geometry: GridGeometry, radar_altitude: float = 0.0,
                               earth_curvature: bool = True, ke: float = EFFECTIVE_RADIUS_FACTOR):
    """Elevation angle (degrees) whose beam reaches ``z_level`` at every (y, x) pixel
    (``radar_grid/products.py:629-697``): flat-earth arctangent, refined by five fixed-point steps on the
    4/3-earth height when ``earth_curvature``."""
    dist = np.maximum(_ground_range(geometry), 1.0)          # avoid the singularity at the radar
    el = np.arctan((z_level - radar_altitude) / dist)
    if earth_curvature:
        r_eff = ke * EARTH_RADIUS
        for _ in range(5):
            slant = dist / np.maximum(np.cos(el), 0.01)
            height = np.sqrt(slant**2 + r_eff**2 + 2 * slant * r_eff * np.sin(el)) - r_eff + radar_altitude
            el = np.clip(el + (z_level - height) / (slant + 1), -np.pi / 2, np.pi / 2)
    return np.degrees(el)


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch")


def _to_device_grid(grid):
    """Returns (cuda float32 contiguous tensor [nz, ny, nx], came_from_numpy)."""
    torch = _native.torch_mod()
    if _is_tensor(grid):
        if not grid.is_cuda:
            raise _native.NativeUnavailable("products run on the GPU: pass a cuda tensor or a NumPy array")
        if grid.dtype != torch.float32:
            raise ValueError("device grids must be float32")
        return grid.contiguous(), False
    dev = _native.device()
    if isinstance(grid, np.ma.MaskedArray):
        grid = np.ma.getdata(grid)   # the reference's arithmetic also acts on the raw data (products.py:407-411)
    arr = np.ascontiguousarray(grid, dtype=np.float32)
    if arr.ndim != 3:
        raise ValueError("grid must have shape (nz, ny, nx)")
    return torch.from_numpy(arr).to(dev), True


def _finish(t, as_numpy: bool):
    return t.cpu().numpy() if as_numpy else t


def cappi_plan(z_limits, nz: int, altitude: float, interpolation: str = "linear"):
    """The scalar decisions of a CAPPI, taken once on the host (``radar_grid/products.py:361-404``), as a tuple:

    * ``("outside",)`` -- ``altitude`` is not within ``[z_min, z_max]``: all-NaN plane;
    * ``("level", k)`` -- the plane IS level ``k`` (nearest-level mode, an exact level hit within ``rtol=1e-6``, or a
      fractional position that falls off either end);
    * ``("blend", k, w_k, w_k1)`` -- float32 blend of levels ``k`` and ``k + 1``.

    Level altitudes are the float32 ``linspace`` of the limits; the fractional position uses the float64 step.
    """
    lo, hi = z_limits
    if not lo <= altitude <= hi:
        return ("outside",)
    levels = np.linspace(lo, hi, nz, dtype="float32")
    if interpolation == "nearest":
        return ("level", int(np.abs(levels - altitude).argmin()))
    if interpolation != "linear":
        raise ValueError(f"Unknown interpolation method: {interpolation}")
    exact = np.flatnonzero(np.isclose(levels, altitude, rtol=1e-6))
    if exact.size:
        return ("level", int(exact[0]))
    spacing = (hi - lo) / (nz - 1) if nz > 1 else 1.0
    position = (altitude - lo) / spacing
    k = int(np.floor(position))
    if k < 0 or k + 1 >= nz:
        return ("level", min(max(k, 0), nz - 1))
    upper = position - k
    return ("blend", k, 1.0 - upper, upper)


def constant_altitude_ppi(grid, geometry: GridGeometry, altitude: float, interpolation: str = "linear"):
    """CAPPI at ``altitude`` metres (``radar_grid/products.py:317-415``).

    Out of ``[z_min, z_max]`` -> warning + all-NaN float32; ``'nearest'`` and an exact level hit return that
    level of ``grid`` (a view, like the reference); otherwise the float32 lerp of the two bracketing levels.
    """
    nz, ny, nx = geometry.grid_shape
    plan = cappi_plan(geometry.grid_limits[0], nz, altitude, interpolation)
    if plan[0] == "outside":
        z_min, z_max = geometry.grid_limits[0]
        logger.warning(f"Altitude {altitude}m is outside grid range [{z_min}, {z_max}]m")
        if _is_tensor(grid):
            return grid.new_full((ny, nx), float("nan"))
        return np.full((ny, nx), np.nan, dtype="float32")
    if plan[0] == "level":
        return grid[plan[1], :, :]
    _, k, w_k, w_k1 = plan
    torch = _native.torch_mod()
    lib = _native.load_library()
    g, as_numpy = _to_device_grid(grid)
    out = torch.empty((ny, nx), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        # weak Python-float weights act as float32 under NumPy >= 2
        _native.check(lib.rg_cappi_lerp_f32(_native.ptr(g), ny * nx, k, float(np.float32(w_k)), float(np.float32(w_k1)),
                                            _native.ptr(out), _native.stream_ptr()), "rg_cappi_lerp_f32")
    return _finish(out, as_numpy)


def constant_elevation_ppi(grid, geometry: GridGeometry, elevation_angle: float, interpolation: str = "linear",
                           earth_curvature: bool = True, ke: float = EFFECTIVE_RADIUS_FACTOR):
    """PPI at a constant elevation angle sampled from the 3-D grid (``radar_grid/products.py:168-314``).

    For every (y, x) the beam height of ``elevation_angle`` (4/3-earth model, or flat earth) is the target
    altitude; ``'linear'`` interpolates between the bracketing levels and returns float64 (NaN outside
    ``[z_min, z_max]``), ``'nearest'`` samples the nearest level and returns float32.  Runs in
    ``rg_elevation_ppi_f32``; bit-identical to the reference's NumPy evaluation.
    """
    if interpolation not in ("linear", "nearest"):
        raise ValueError(f"Unknown interpolation method: {interpolation}")
    nz, ny, nx = (int(v) for v in geometry.grid_shape)
    z_min, z_max = geometry.grid_limits[0]
    (y_lo, y_hi), (x_lo, x_hi) = geometry.grid_limits[1], geometry.grid_limits[2]
    z_step = (z_max - z_min) / (nz - 1) if nz > 1 else 1.0
    el = np.radians(elevation_angle)
    r_eff = ke * EARTH_RADIUS
    torch = _native.torch_mod()
    lib = _native.load_library()
    g, as_numpy = _to_device_grid(grid)
    xc = torch.from_numpy(np.linspace(x_lo, x_hi, nx, dtype="float32")).to(g.device)
    yc = torch.from_numpy(np.linspace(y_lo, y_hi, ny, dtype="float32")).to(g.device)
    linear = interpolation == "linear"
    out = torch.empty((ny, nx), dtype=torch.float64 if linear else torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _native.check(lib.rg_elevation_ppi_f32(
            _native.ptr(g), _native.ptr(xc), _native.ptr(yc), nz, ny, nx, float(np.maximum(np.cos(el), 0.01)),
            float(np.sin(el)), float(np.tan(el)), float(r_eff), float(r_eff**2), float(z_min), float(z_max),
            float(z_step), int(bool(earth_curvature)), int(linear), _native.ptr(out), _native.stream_ptr()),
            "rg_elevation_ppi_f32")
    return _finish(out, as_numpy)


def _level_window(nz, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry):
    """Index window of the column products (``products.py:462-485``)."""
    if z_min_alt is not None or z_max_alt is not None:
        if geometry is None:
            raise ValueError("geometry is required when using altitude-based limits")
        z_lo, z_hi = geometry.grid_limits[0]
        z_coords = np.linspace(z_lo, z_hi, nz)
        if z_min_alt is not None:
            z_min_idx = int(np.searchsorted(z_coords, z_min_alt))
        if z_max_alt is not None:
            z_max_idx = int(np.searchsorted(z_coords, z_max_alt, side="right")) - 1
    if z_min_idx is None:
        z_min_idx = 0
    if z_max_idx is None:
        z_max_idx = nz - 1
    return max(0, z_min_idx), min(nz - 1, z_max_idx)


def _column(op: str, grid, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry, want_arg=False):
    nz = int(grid.shape[0])
    lo, hi = _level_window(nz, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry)
    if lo > hi:
        raise ValueError(f"empty level window [{lo}, {hi}]")   # np.nanmax raises on a zero-size slice too
    torch = _native.torch_mod()
    lib = _native.load_library()
    g, as_numpy = _to_device_grid(grid)
    ny, nx = int(g.shape[1]), int(g.shape[2])
    out = torch.empty((ny, nx), dtype=torch.float32, device=g.device)
    arg = torch.empty((ny, nx), dtype=torch.int32, device=g.device) if want_arg else None
    with torch.cuda.device(g.device):
        _native.check(lib.rg_column_reduce_f32(_native.ptr(g), nz, ny * nx, lo, hi, _native.COLUMN_OPS[op],
                                               _native.ptr(out), _native.ptr(arg), _native.stream_ptr()),
                      "rg_column_reduce_f32")
    if want_arg:
        return _finish(out, as_numpy), _finish(arg, as_numpy)
    return _finish(out, as_numpy)


def column_max(grid, z_min_idx: Optional[int] = None, z_max_idx: Optional[int] = None,
               z_min_alt: Optional[float] = None, z_max_alt: Optional[float] = None,
               geometry: Optional[GridGeometry] = None):
    """COLMAX: NaN-ignoring maximum of every vertical column (``radar_grid/products.py:420-490``); an all-NaN
    column stays NaN."""
    return _column("max", grid, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry)


def column_min(grid, z_min_idx: Optional[int] = None, z_max_idx: Optional[int] = None,
               z_min_alt: Optional[float] = None, z_max_alt: Optional[float] = None,
               geometry: Optional[GridGeometry] = None):
    """NaN-ignoring column minimum (``radar_grid/products.py:493-535``)."""
    return _column("min", grid, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry)


def column_mean(grid, z_min_idx: Optional[int] = None, z_max_idx: Optional[int] = None,
                z_min_alt: Optional[float] = None, z_max_alt: Optional[float] = None,
                geometry: Optional[GridGeometry] = None):
    """NaN-ignoring column mean (``radar_grid/products.py:538-580``)."""
    return _column("mean", grid, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry)


def column_argmax(grid, z_min_idx: Optional[int] = None, z_max_idx: Optional[int] = None,
                  z_min_alt: Optional[float] = None, z_max_alt: Optional[float] = None,
                  geometry: Optional[GridGeometry] = None):
    """``(colmax, level)``: the column maximum and the int32 index (into the full grid) of the FIRST level that
    attains it, ``-1`` where the column is all NaN.  Not in the reference: ``np.nanargmax``
    semantics on the same 3-D grid define the contract."""
    return _column("max", grid, z_min_idx, z_max_idx, z_min_alt, z_max_alt, geometry, want_arg=True)
