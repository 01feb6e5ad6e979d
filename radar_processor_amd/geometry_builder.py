"""GPU geometry builder -- mirror of ``radar_grid/compute.py`` (``compute_grid_geometry`` :106-284,
``_process_single_level`` :18-103).

The reference builds a cKDTree per z-level and walks every voxel in a Python loop, exchanging levels through
temp ``.npz`` files.  Here the whole build is three kernels on the device (csrc/rg_geometry.hip):
bucket + radix-sort the gates into a cell grid, count every voxel's neighbours, prefix-sum, fill.  Membership
(``z_rel <= toa`` and ``d2 < r2``) and the weights are evaluated in the reference's float64 arithmetic from the
same float32 inputs, so the neighbour sets are identical and the weights agree to the last float32 bit
(Barnes: up to 1 ulp where the two ``exp`` implementations round differently).  Row order inside a voxel is
(cell row, gate index) instead of KD-tree traversal order.

:class:`RoiSearch` keeps the sorted-gate structure so the same search can also feed the fused on-the-fly
gridder (``roi_grid.py``) without materialising a CSR.
"""
from __future__ import annotations

import logging
import math
import os
from typing import Optional, Tuple

import numpy as np

from . import _native
from .grid_geometry import DeviceCSR, GridGeometry

logger = logging.getLogger("radar_grid.compute")

_INT32_MAX = np.iinfo(np.int32).max
WEIGHTINGS = ("barnes2", "cressman", "nearest")


def _as_device_f32(a, dev):
    torch = _native.torch_mod()
    if type(a).__module__.startswith("torch"):
        return a.to(device=dev, dtype=torch.float32).contiguous().view(-1)
    arr = np.asarray(a)
    if arr.dtype != np.float32:
        logger.warning("gate coordinates are %s; rounding to float32 as get_gate_coordinates() does", arr.dtype)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).ravel()).to(dev)


class RoiSearch:
    """Cell-sorted gates + voxel coordinate tables in HBM: everything the ROI search kernels need."""

    def __init__(self, gate_x, gate_y, gate_z, grid_shape, grid_limits, radar_altitude=0.0, min_radius=250.0,
                 beam_factor=0.01746, toa=17000.0, device=None, cell_size: Optional[float] = None,
                 per_level: Optional[bool] = None):
        """``per_level``: keep one cell-sorted gate list PER GRID LEVEL, each holding only the gates that can reach that level
        (``rg_geom_bin_gates_levels_f32``: |z_gate - z_level| within the largest radius of influence any neighbouring voxel of
        the gate can have) -- a voxel block then streams a third of the candidates.  Default: on where the bound holds
        (0 <= beam_factor < 0.5) and the grid has more than one level; the neighbour sets are the same either way."""
        torch = _native.torch_mod()
        lib = _native.load_library()
        self.dev = _native.canonical_device(device)
        self.grid_shape = tuple(int(s) for s in grid_shape)
        self.grid_limits = grid_limits
        self.min_radius = float(min_radius)
        self.beam_factor = float(beam_factor)
        self.toa = toa
        nz, ny, nx = self.grid_shape

        gx = _as_device_f32(gate_x, self.dev)
        gy = _as_device_f32(gate_y, self.dev)
        gz = _as_device_f32(gate_z, self.dev)
        if not (gx.numel() == gy.numel() == gz.numel()):
            raise ValueError("gate_x, gate_y and gate_z differ in length")
        self.n_gates = int(gx.numel())

        # compute.py:184-186 -- float32 linspace tables (NumPy's own rounding; never re-derived on the device)
        zc = np.linspace(grid_limits[0][0], grid_limits[0][1], nz, dtype="float32")
        yc = np.linspace(grid_limits[1][0], grid_limits[1][1], ny, dtype="float32")
        xc = np.linspace(grid_limits[2][0], grid_limits[2][1], nx, dtype="float32")
        self.zc, self.yc, self.xc = (torch.from_numpy(c).to(self.dev) for c in (zc, yc, xc))

        # largest ROI over the grid (reached at a corner), padded so dropped gates provably cannot be neighbours
        far = math.sqrt(max(abs(float(xc.min())), abs(float(xc.max()))) ** 2
                        + max(abs(float(yc.min())), abs(float(yc.max()))) ** 2
                        + max(abs(float(zc.min())), abs(float(zc.max()))) ** 2)
        r_max = max(self.min_radius, far * abs(self.beam_factor)) * (1.0 + 1e-6) + 1.0
        self.r_max = r_max
        x_lo, x_hi = float(xc.min()) - r_max, float(xc.max()) + r_max
        y_lo, y_hi = float(yc.min()) - r_max, float(yc.max()) + r_max

        want_levels = (nz > 1 and 0.0 <= self.beam_factor < 0.5 and self.min_radius >= 0.0) if per_level is None else bool(per_level)
        if cell_size is None:
            # a level's list holds about a third of the gates: three times the cell size keeps a cell row of a voxel's search
            # box at about one wavefront of candidates (bench grid, ms per K2 pass: x1 13.7, x1.5 12.3, x2 11.9, x3 11.6, x4 11.6)
            cell_size = self._auto_cell(gx, gy, x_lo, x_hi, y_lo, y_hi) * (3.0 if want_levels else 1.0)
        # keep the cell table small (int32 entries): at most ~16 M cells
        span = max(x_hi - x_lo, y_hi - y_lo)
        cell_size = max(float(cell_size), span / 4000.0, 1.0)
        self.cell_size = cell_size
        ncx = max(1, int(math.ceil((x_hi - x_lo) / cell_size)))
        ncy = max(1, int(math.ceil((y_hi - y_lo) / cell_size)))
        self.per_level = bool(want_levels and ncx * ncy * nz < 2 ** 31 - 1)
        self.cells = _native.CellGrid(x0=x_lo, y0=y_lo, inv_cx=1.0 / cell_size, inv_cy=1.0 / cell_size,
                                      z_lo=float(zc.min()) - r_max, z_hi=float(zc.max()) + r_max, ncx=ncx, ncy=ncy,
                                      levels=nz if self.per_level else 0, level0=0)
        alt32, toa32 = float(np.float32(radar_altitude)), float(np.float32(toa))
        if self.per_level:
            # one list per level: count the (gate, level) entries, then bin them
            total = torch.zeros(1, dtype=torch.int64, device=self.dev)
            with torch.cuda.device(self.dev):
                _native.check(lib.rg_geom_bin_levels_count(
                    _native.ptr(gx), _native.ptr(gy), _native.ptr(gz), self.n_gates, alt32, toa32, self.cells,
                    _native.ptr(self.zc), nz, self.min_radius, self.beam_factor, _native.ptr(total), _native.stream_ptr()),
                    "rg_geom_bin_levels_count")
                n_entries = int(total.item())
            if n_entries > 2 ** 31 - 1:
                raise _native.NativeError(f"{n_entries} (gate, level) entries exceed the int32 positions of the cell table; "
                                          "pass per_level=False")
            self.sorted_gates = torch.empty(max(n_entries, 1) * 4, dtype=torch.float32, device=self.dev)
            self.cell_start = torch.empty(ncx * ncy * nz + 1, dtype=torch.int32, device=self.dev)
            ws_bytes = int(lib.rg_geom_bin_levels_workspace_bytes(self.n_gates, n_entries, ncx * ncy * nz))
            if ws_bytes < 0:
                raise _native.NativeError("rg_geom_bin_levels_workspace_bytes rejected its arguments")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.dev)
            with torch.cuda.device(self.dev):
                _native.check(lib.rg_geom_bin_gates_levels_f32(
                    _native.ptr(gx), _native.ptr(gy), _native.ptr(gz), self.n_gates, alt32, toa32, self.cells,
                    _native.ptr(self.zc), nz, self.min_radius, self.beam_factor, n_entries, _native.ptr(self.sorted_gates),
                    _native.ptr(self.cell_start), _native.ptr(ws), ws_bytes, _native.stream_ptr()),
                    "rg_geom_bin_gates_levels_f32")
                self.n_binned = int(self.cell_start[-1].item())      # (gate, level) entries
            del ws
            return
        self.sorted_gates = torch.empty(max(self.n_gates, 1) * 4, dtype=torch.float32, device=self.dev)
        self.cell_start = torch.empty(ncx * ncy + 1, dtype=torch.int32, device=self.dev)
        ws_bytes = int(lib.rg_geom_bin_workspace_bytes(self.n_gates, ncx, ncy))
        if ws_bytes < 0:
            raise _native.NativeError("rg_geom_bin_workspace_bytes rejected its arguments")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.dev)
        with torch.cuda.device(self.dev):
            _native.check(lib.rg_geom_bin_gates_f32(
                _native.ptr(gx), _native.ptr(gy), _native.ptr(gz), self.n_gates, alt32, toa32, self.cells,
                _native.ptr(self.sorted_gates), _native.ptr(self.cell_start), _native.ptr(ws), ws_bytes,
                _native.stream_ptr()), "rg_geom_bin_gates_f32")
            self.n_binned = int(self.cell_start[-1].item())
        del ws

    def cells_from(self, level0: int):
        """The cell grid for a call that covers the levels from ``level0`` on (passing ``zc + level0``): per-level gate lists
        are addressed by the absolute level."""
        if not self.per_level or level0 == 0:
            return self.cells
        c = self.cells
        return _native.CellGrid(x0=c.x0, y0=c.y0, inv_cx=c.inv_cx, inv_cy=c.inv_cy, z_lo=c.z_lo, z_hi=c.z_hi, ncx=c.ncx,
                                ncy=c.ncy, levels=c.levels, level0=int(level0))

    def _auto_cell(self, gx, gy, x_lo, x_hi, y_lo, y_hi) -> float:
        """Cell size for which one cell row of a typical voxel's search box holds about one wavefront (64) of
        candidates.  For polar data the areal gate density falls off as 1/D while the ROI grows as D, so
        ``2*beam_factor*cell * N / (2*pi*R_max)`` candidates per row, independent of range."""
        torch = _native.torch_mod()
        if self.n_gates == 0:
            return max(self.min_radius, 1.0)
        ground = torch.sqrt(gx * gx + gy * gy)
        r_gate = float(torch.nan_to_num(ground, nan=0.0, posinf=0.0).max().item())
        n = self.n_gates
        bf = max(abs(self.beam_factor), 1e-6)
        cell = 64.0 * math.pi * max(r_gate, 1.0) / (bf * n)
        return float(min(max(cell, 0.25 * self.min_radius, 25.0), 8.0 * max(self.min_radius, 1.0) + 2000.0))

    # ------------------------------------------------------------------------------------------------
    def count_pairs(self) -> int:
        """Total number of (voxel, gate) pairs of the geometry (one count pass, nothing is kept)."""
        torch = _native.torch_mod()
        lib = _native.load_library()
        nz, ny, nx = self.grid_shape
        with torch.cuda.device(self.dev):
            counts = torch.zeros(nz * ny * nx + 1, dtype=torch.int32, device=self.dev)
            _native.check(lib.rg_geom_count_f32(
                _native.ptr(self.sorted_gates), _native.ptr(self.cell_start), self.cells, _native.ptr(self.xc),
                _native.ptr(self.yc), _native.ptr(self.zc), nz, ny, nx, self.min_radius, self.beam_factor,
                _native.ptr(counts), _native.stream_ptr()), "rg_geom_count_f32")
            return int(counts.sum(dtype=torch.int64).item())

    def build_csr(self, weighting: str = "barnes2") -> DeviceCSR:
        torch = _native.torch_mod()
        lib = _native.load_library()
        nz, ny, nx = self.grid_shape
        n_vox = nz * ny * nx
        with torch.cuda.device(self.dev):
            stream = _native.stream_ptr()
            counts = torch.zeros(n_vox + 1, dtype=torch.int32, device=self.dev)
            _native.check(lib.rg_geom_count_f32(
                _native.ptr(self.sorted_gates), _native.ptr(self.cell_start), self.cells, _native.ptr(self.xc),
                _native.ptr(self.yc), _native.ptr(self.zc), nz, ny, nx, self.min_radius, self.beam_factor,
                _native.ptr(counts), stream), "rg_geom_count_f32")
            indptr = torch.empty(n_vox + 1, dtype=torch.int64, device=self.dev)
            ws_bytes = int(lib.rg_scan_workspace_bytes(n_vox))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.dev)
            _native.check(lib.rg_scan_counts_i64(_native.ptr(counts), n_vox, _native.ptr(indptr), _native.ptr(ws),
                                                 ws_bytes, stream), "rg_scan_counts_i64")
            n_pairs = int(indptr[-1].item())
            del counts, ws
            gate_idx = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=self.dev)[:n_pairs]
            weights = torch.empty(max(n_pairs, 1), dtype=torch.float32, device=self.dev)[:n_pairs]
            if n_pairs:
                _native.check(lib.rg_geom_fill_f32(
                    _native.ptr(self.sorted_gates), _native.ptr(self.cell_start), self.cells, _native.ptr(self.xc),
                    _native.ptr(self.yc), _native.ptr(self.zc), nz, ny, nx, self.min_radius, self.beam_factor,
                    _native.WEIGHTINGS[weighting], _native.ptr(indptr), _native.ptr(gate_idx), _native.ptr(weights),
                    stream), "rg_geom_fill_f32")
            if n_pairs <= _INT32_MAX:
                indptr = indptr.to(torch.int32)   # the reference's dtype (compute.py:232) whenever it fits
            max_gate = int(gate_idx.max().item()) if n_pairs else -1
        return DeviceCSR(indptr, gate_idx, weights, max_gate)


# weightings whose weights fit the 26-bit code of the packed records, and the code's base: a float32 exponent that is a
# multiple of 8 (the kernels OR it back in) and at most 7 below the largest exponent a weight can have.  barnes2:
# exp(-4)+1e-5 = 2^-6 * 1.17 (exponent 121) .. 1+1e-5 (127); nearest: 1.0 (127)
_PACK_BASE_EXPONENT = {"barnes2": 120, "nearest": 120}
_AUTO_PACKED_MIN_PAIRS = 50_000_000      # layout="auto": from here on the packed layout alone (gridding._COMPACT_MIN_PAIRS)


def _build_compact_only(search: "RoiSearch", weighting: str, pairs_per_slab: int = 1_200_000_000, packed: bool = False):
    """Count -> scan -> per slab of whole grid levels: fill the slab's gate indices into a scratch buffer, derive the
    dictionaries and 16-bit positions of the slab's chunks (chunks never cross a level), drop the scratch.  The int32
    index array of the whole grid -- half of the standard CSR -- never exists, so a geometry of P pairs needs about
    6.2*P bytes instead of 8*P (+2.2*P for the copy).

    ``packed``: additionally fold every slab's positions and weights into the 16-byte records of
    ``rg_csr_compact_pack`` (three pairs each) and keep ONLY those: 5.4*P bytes, and the kernel's fastest stream.  The
    weight code is lossless for weights within 8 binades of the weighting's smallest possible value (Barnes, uniform);
    should a weight fall outside (it cannot for those weightings), the pack kernel flags it and the caller falls back.

    Returns ``(DeviceCSR without gate_indices [and without weights], CompactCSR)`` or ``None`` when a chunk holds more
    than 65536 distinct gates in a single segment (or packing was asked for and is not possible)."""
    from .grid_geometry import CompactCSR
    torch = _native.torch_mod()
    lib = _native.load_library()
    nz, ny, nx = search.grid_shape
    n_xy = ny * nx
    n_vox = nz * n_xy
    dev = search.dev
    if packed and weighting not in _PACK_BASE_EXPONENT:
        return None
    with torch.cuda.device(dev):
        stream = _native.stream_ptr()
        counts = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
        _native.check(lib.rg_geom_count_f32(
            _native.ptr(search.sorted_gates), _native.ptr(search.cell_start), search.cells, _native.ptr(search.xc),
            _native.ptr(search.yc), _native.ptr(search.zc), nz, ny, nx, search.min_radius, search.beam_factor,
            _native.ptr(counts), stream), "rg_geom_count_f32")
        indptr = torch.empty(n_vox + 1, dtype=torch.int64, device=dev)
        ws_bytes = int(lib.rg_scan_workspace_bytes(n_vox))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _native.check(lib.rg_scan_counts_i64(_native.ptr(counts), n_vox, _native.ptr(indptr), _native.ptr(ws), ws_bytes,
                                             stream), "rg_scan_counts_i64")
        del counts, ws
        level_ptr = indptr[::n_xy].cpu().numpy()                 # pair offset at the start of every grid level
        n_pairs = int(level_ptr[-1])
        weights = local = rec = rec_ptr = None
        w_base = 0
        if packed:
            from . import grid_geometry
            rec_order = grid_geometry.DEFAULT_REC_ORDER
            rec_ptr = CompactCSR.record_pointers(indptr, search.grid_shape, rec_order)
            if rec_ptr is None:
                return None
            slots_per_plane = (rec_ptr.numel() - 1) // nz      # chunks never cross a plane: neither do the slots
            n_rec = int(rec_ptr[-1])
            rec = torch.empty((max(n_rec, 1), 4), dtype=torch.int32, device=dev)[:n_rec]
            w_base = _PACK_BASE_EXPONENT[weighting] << 23
            err = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            weights = torch.empty(max(n_pairs, 1), dtype=torch.float32, device=dev)[:n_pairs]
            local = torch.empty(max(n_pairs, 1), dtype=torch.int16, device=dev)[:n_pairs]
        count_parts, dict_parts = [], []
        max_gate = -1
        iz0 = 0
        while iz0 < nz:
            iz1 = iz0 + 1
            while iz1 < nz and level_ptr[iz1 + 1] - level_ptr[iz0] <= pairs_per_slab:
                iz1 += 1
            p0, p1 = int(level_ptr[iz0]), int(level_ptr[iz1])
            scratch = torch.empty(max(p1 - p0, 1), dtype=torch.int32, device=dev)[:p1 - p0]
            if packed:       # the slab's weights and positions are scratch too: only the records survive
                w_slab = torch.empty(max(p1 - p0, 1), dtype=torch.float32, device=dev)[:p1 - p0]
                l_slab = torch.empty(max(p1 - p0, 1), dtype=torch.int16, device=dev)[:p1 - p0]
                w_ptr, l_ptr = _native.ptr(w_slab) - 4 * p0, _native.ptr(l_slab) - 2 * p0
            else:
                w_ptr, l_ptr = _native.ptr(weights), _native.ptr(local)
            if p1 > p0:
                # the fill kernel writes at absolute pair positions: shift the pointers so that the slab's first pair
                # lands at the start of the scratch buffers (the weights go straight to their final place otherwise)
                _native.check(lib.rg_geom_fill_f32(
                    _native.ptr(search.sorted_gates), _native.ptr(search.cell_start), search.cells_from(iz0), _native.ptr(search.xc),
                    _native.ptr(search.yc), _native.ptr(search.zc) + 4 * iz0, iz1 - iz0, ny, nx, search.min_radius,
                    search.beam_factor, _native.WEIGHTINGS[weighting], _native.ptr(indptr) + 8 * iz0 * n_xy,
                    _native.ptr(scratch) - 4 * p0, w_ptr, stream), "rg_geom_fill_f32")
                max_gate = max(max_gate, int(scratch.max().item()))
            built = CompactCSR._planes(indptr[iz0 * n_xy:iz1 * n_xy + 1], _native.ptr(scratch) - 4 * p0, iz1 - iz0, ny, nx,
                                       l_ptr)
            if built is None:
                return None
            count_parts.append(built[0])
            dict_parts.append(built[1])
            if packed and p1 > p0:
                _native.check(lib.rg_csr_compact_pack(
                    _native.ptr(indptr) + 8 * iz0 * n_xy, 1, l_ptr, w_ptr, (iz1 - iz0) * n_xy, nx, ny,
                    _native.ptr(rec_ptr) + 8 * iz0 * slots_per_plane, rec_order, iz0, w_base, _native.ptr(rec),
                    _native.ptr(err), stream), "rg_csr_compact_pack")
                if int(err.item()):
                    logger.info(f"rg_csr_compact_pack flag {int(err.item())} (a weight outside the 26-bit code, or an over-long "
                                "segment): keeping the plain compact layout instead")
                    return None
                del w_slab, l_slab
            del scratch
            iz0 = iz1
        counts_all = torch.cat(count_parts) if count_parts else torch.zeros(0, dtype=torch.int64, device=dev)
        compact = CompactCSR._finish(indptr, search.grid_shape, local, counts_all, dict_parts)
        if packed:
            compact.rec, compact.rec_ptr, compact.rec_order, compact.w_base = rec, rec_ptr, rec_order, w_base
        if n_pairs <= _INT32_MAX:
            indptr = indptr.to(torch.int32)
    return DeviceCSR(indptr, None, weights, max_gate, n_pairs=n_pairs), compact


def compute_grid_geometry(
    gate_x: np.ndarray,
    gate_y: np.ndarray,
    gate_z: np.ndarray,
    grid_shape: Tuple[int, int, int],
    grid_limits: Tuple[Tuple[float, float], ...],
    temp_dir: str,
    radar_altitude: float = 0.0,
    min_radius: float = 250.0,
    beam_factor: float = 0.01746,
    weighting: str = "barnes2",
    toa: float = 17000.0,
    n_workers: Optional[int] = None,
    layout: str = "csr",
) -> GridGeometry:
    """Precompute which gates contribute to each voxel and with what weight
    (``radar_grid/compute.py:106-284``; same signature).

    ``temp_dir`` and ``n_workers`` are accepted for compatibility: the directory must exist (the reference
    raises ``ValueError`` otherwise, compute.py:173-174) but nothing is written to it, and there is no worker
    pool -- the build runs on the GPU.  The result is device resident; ``.indptr`` / ``.gate_indices`` /
    ``.weights`` copy to the host on first access (``save_geometry`` does that).

    ``layout`` (build-specific): ``"csr"`` keeps the reference's three arrays in HBM; ``"auto"`` does so for geometries
    below 50 M pairs and for weights that are not codable (while they fit), builds ``"packed"`` for every larger Barnes /
    uniform geometry, and otherwise falls to ``"compact"``, which keeps
    ``indptr``, ``weights`` and the compact copy of the gate indices only (``grid_geometry.CompactCSR``, 6.2 instead
    of 8 bytes per pair), or to ``"packed"``, which keeps ``indptr`` and the packed pair stream only (positions and
    losslessly coded weights of three pairs per 16-byte record: 5.4 bytes per pair; Barnes and uniform weights) -- for
    geometries too large to hold both; ``.gate_indices`` / ``.weights`` are then rebuilt from the copy when somebody
    asks for them.

    Reference quirks kept on purpose: ``radar_altitude`` is subtracted from ``gate_z``
    in float32 (compute.py:182), the returned geometry does not carry it (compute.py:277-284 => 0.0), and
    ``'nearest'`` means a uniform mean over the ROI, not the nearest gate (compute.py:86-87).
    """
    if not os.path.isdir(temp_dir):
        raise ValueError(f"temp_dir does not exist: {temp_dir}")
    if weighting not in WEIGHTINGS:
        raise ValueError(f"Unknown weighting function: {weighting}")

    search = RoiSearch(gate_x, gate_y, gate_z, grid_shape, grid_limits, radar_altitude=radar_altitude,
                       min_radius=min_radius, beam_factor=beam_factor, toa=toa)
    nz, ny, nx = search.grid_shape
    logger.info(f"Radar altitude: {radar_altitude:.1f} m")
    logger.info(f"TOA filter: {search.n_binned:,} {'(gate, level) entries' if search.per_level else 'gates'} of {search.n_gates:,} gates kept (below {toa}m and within reach "
                f"of the grid); cell size {search.cell_size:.0f} m")
    logger.info(f"Processing {nz} z-levels on {search.dev}...")
    if layout not in ("csr", "compact", "packed", "auto"):
        raise ValueError("layout must be 'csr', 'compact', 'packed' or 'auto'")
    if layout == "auto":
        # Large geometries whose weights fit the 26-bit code (Barnes, uniform) are built in the packed layout ALONE: row
        # pointers + dictionaries + 16-byte records, 5.4 bytes per pair -- it is what every gridding pass of 1-8 fields reads
        # anyway, the reference's index and weight arrays are rebuilt from it on demand, bit for bit (CompactCSR.decode /
        # decode_weights: save_geometry, the CPU baseline, the bench's post-check), and a process touches 47 GB instead of
        # 115 GB for the bench geometry (first-touch allocation is what a build's wall time consists of).  Geometries below
        # 50 M pairs grid in well under a millisecond either way and keep the reference's arrays; non-codable weights
        # (Cressman) keep them too when they and their compact copy fit, and fall to the compact layout otherwise.
        n_pairs = search.count_pairs()
        free_b, _ = _native.torch_mod().cuda.mem_get_info(search.dev)
        if n_pairs >= _AUTO_PACKED_MIN_PAIRS and weighting in _PACK_BASE_EXPONENT:
            layout = "packed"
        else:
            layout = "csr" if 10.4 * n_pairs + (10 << 30) <= free_b else "compact"
        logger.info(f"{n_pairs:,} pairs, {free_b / 1e9:.0f} GB free -> layout '{layout}'")
    if layout == "packed":
        built = _build_compact_only(search, weighting, packed=True)
        if built is not None:
            csr, compact = built
            logger.info(f"Geometry complete ({csr.n_pairs:,} total pairs, packed layout: {compact.rec.shape[0]:,} records, "
                        f"{compact.n_dict:,} dictionary entries).")
            return GridGeometry.from_device(grid_shape, grid_limits, csr, toa, compact=compact)
        logger.info("packed layout not possible for this geometry; building the compact layout")
        layout = "compact"
    if layout == "compact":
        built = _build_compact_only(search, weighting)
        if built is not None:
            csr, compact = built
            logger.info(f"Geometry complete ({csr.n_pairs:,} total pairs, compact layout: {compact.n_dict:,} dictionary "
                        f"entries).")
            return GridGeometry.from_device(grid_shape, grid_limits, csr, toa, compact=compact)
        logger.info("compact layout not possible for this grid; building the standard CSR")
    csr = search.build_csr(weighting)
    logger.info(f"Geometry complete ({csr.n_pairs:,} total pairs).")
    return GridGeometry.from_device(grid_shape, grid_limits, csr, toa)
