"""Field gridding through a precomputed geometry -- mirror of ``radar_grid/interpolate.py``
(``apply_geometry`` :15-104, ``apply_geometry_multi`` :107-142), executed by the HIP kernel
``rg_csr_apply_f32`` (csrc/rg_csr_apply.hip).

Two layers:

* :func:`apply_geometry` / :func:`apply_geometry_multi` keep the reference signatures (NumPy masked arrays in,
  float32 ``ndarray`` of ``geometry.grid_shape`` out).  They stage the inputs into HBM, run the kernels and
  copy the grid back; the geometry itself is uploaded once and cached on the ``GridGeometry`` object.
* :func:`grid_fields_device` is the device-resident form the batch driver and the benchmark use: field
  tensors already in HBM in, grid tensor in HBM out, nothing crosses PCIe.

All fields handed to one call are gridded by ONE pass over the CSR (the reference re-reads the CSR per field,
interpolate.py:137-140): the mask of every field is folded into its values and the fields are interleaved
gate-major, so each (voxel, gate) pair costs a single gather.  Passes over a large geometry (50 M pairs and up) run
through a compact device copy of the CSR, built the first time the geometry grids something (16-bit positions in a
per-chunk gate dictionary, the chunk's field values staged in LDS, positions and weights packed three pairs to a 16-byte
record where the weights allow: ``rg_csr_compact_apply_packed_f32`` / ``rg_csr_compact_apply_f32``) -- the same values to
float32 rounding, 35-50 % less time.
"""
from __future__ import annotations

import ctypes
import logging
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native
from .gate_filters import GateFilter
from .grid_geometry import GridGeometry

logger = logging.getLogger("radar_grid.interpolate")


def _stride_for(n_fields: int) -> int:
    return 1 if n_fields == 1 else 2 if n_fields == 2 else 4 if n_fields <= 4 else 8


def _coerce_filters(additional_filters) -> List[GateFilter]:
    """``None`` -> ``[]``, a bare filter -> ``[gf]``, a list stays, anything else is a ValueError
    (``interpolate.py:50-56``)."""
    if isinstance(additional_filters, list):
        return additional_filters
    if additional_filters is None:
        return []
    if isinstance(additional_filters, GateFilter):
        return [additional_filters]
    raise ValueError("additional_filters must be a list of GateFilter objects")


def _host_field(field_data, filters: Sequence[GateFilter]):
    """float32 values + merged exclusion mask (``interpolate.py:59-64``).

    The reference indexes ``np.ma.getmask(field)`` and therefore crashes on inputs without a full mask
     unless a filter happens to broadcast it; here a missing mask simply means "nothing masked",
    which is also what the reference computes whenever it does not crash.
    """
    values = np.ascontiguousarray(np.ma.getdata(field_data), dtype=np.float32).ravel()
    mask = np.ma.getmaskarray(field_data).ravel()
    for gf in filters:
        mask = mask | gf.gate_excluded
    return values, np.ascontiguousarray(mask, dtype=np.uint8)


class CsrGridder:
    """Persistent device buffers + launches for gridding field groups of a fixed size through one geometry.

    Holds the packed-field staging buffer so that repeated volumes cost no allocation, and exposes the two
    stages separately (``pack`` = mask fold + interleave, ``apply`` = the CSR pass) so that callers can time
    or graph-capture them.  All launches go to torch's current stream.
    """

    def __init__(self, geometry: GridGeometry, n_gates: int, n_fields: int, device=None, compact: bool = False,
                 tile: int = 0, packed: bool = True):
        """``compact``: run the pass through the compact device copy of the CSR (built and cached on the geometry the
        first time) when the geometry allows one and its LDS window for this field count covers (nearly) all pairs:
        ``rg_csr_compact_apply_packed_f32`` (row-wise kernel over the packed records; agrees with ``rg_csr_apply_f32`` to
        float32 rounding) for 1-8 fields when the weights are codable and ``packed`` is set, ``rg_csr_compact_apply_f32``
        (tile kernel; agrees bit for bit) otherwise.  ``tile``: diagnostic override -- 128...512 the pipeline tile of the
        tile kernels (non-default values change the order of the float32 adds), 384 also selects the tile kernel over
        the packed records, 2000 + h the lane split of the row-wise kernel."""
        torch = _native.torch_mod()
        self.lib = _native.load_library()
        if not 1 <= n_fields <= _native.RG_MAX_FIELDS:
            raise ValueError(f"n_fields must be in 1..{_native.RG_MAX_FIELDS}")
        self.dev = _native.canonical_device(device)
        self.csr = geometry.device_csr(self.dev)
        self.n_gates = int(n_gates)
        self.n_fields = int(n_fields)
        self.stride = _stride_for(n_fields)
        self.tile = int(tile)
        self.grid_shape = tuple(int(s) for s in geometry.grid_shape)
        self.n_vox = int(np.prod(self.grid_shape))
        if self.n_vox != self.csr.n_vox:
            raise ValueError(f"grid_shape {self.grid_shape} does not match the geometry's {self.csr.n_vox} rows")
        if self.csr.max_gate >= self.n_gates:
            # the reference's fancy index (interpolate.py:74) raises the same way
            raise IndexError(f"index {self.csr.max_gate} is out of bounds for axis 0 with size {self.n_gates}")
        self.packed = torch.empty(max(self.n_gates, 1) * self.stride, dtype=torch.float32, device=self.dev)
        compact_only = self.csr.gate_indices is None
        packed_only = self.csr.weights is None           # only the packed pair stream exists (layout="packed")
        # Which kernel (ms per pass on config 2 / the bench grid, MI355X):
        #                              1 field        2 fields       3 fields       4 fields       8 fields
        #   rg_csr_apply_f32           1.9 / 13.1     2.1 / 14.7     2.3 / 16.5     3.0 / 20.5     7.3 / 49.8
        #   compact, tile kernel       1.11 / 8.4     1.69 / 11.6    2.2 / 14.6     3.3 / 19.5     9.6 / 65
        #   compact, row-wise kernel   1.02 / 7.9     1.13 / 8.7     1.29 / 10.1    1.47 / 10.3    3.9 / 14.4
        # The row-wise kernel reads the packed records (weights codable in 26 bits: Barnes, nearest) and has no tile in
        # LDS, so it wins at every window it can hold.  The tile kernel (weights not codable) wins while its
        # window stays <= 24 KiB next to its tiles and loses beyond; a compact-only geometry has no choice.
        want = compact or compact_only
        self.compact = geometry.device_compact(self.dev) if (want and self.csr.n_pairs) else None
        self.window = 0
        self.packed_stream = False
        if self.compact is not None:
            self.window = self.compact.window_for(self.n_fields)
            if not compact_only and self.compact.fallback_fraction(self.window) > _COMPACT_MAX_FALLBACK:
                # too many chunks would gather per pair (dense scans next to many fields): the standard kernel is faster
                self.compact, self.window = None, 0
        if self.compact is not None:
            # passes of 1-8 fields stream the packed records (positions + weights, 5.33 bytes per pair) when the geometry's
            # weights allow the lossless 26-bit code and the memory is there
            self.packed_stream = bool((packed or packed_only) and self.compact.ensure_packed(self.csr))
            if self.packed_stream:
                self.window = self.compact.window_for(self.n_fields, rowwise=True)
            if (not self.packed_stream and not compact_only
                    and self.window * self.compact.entry_bytes(self.n_fields) > _COMPACT_MAX_WINDOW_BYTES):
                self.compact, self.window = None, 0

    def _check_fields(self, fields, masks, shared_mask):
        torch = _native.torch_mod()
        if len(fields) != self.n_fields:
            raise ValueError(f"expected {self.n_fields} fields, got {len(fields)}")
        for i, f in enumerate(fields):
            if not (f.is_cuda and f.device == self.packed.device and f.dtype == torch.float32 and f.is_contiguous()
                    and f.numel() == self.n_gates):
                raise ValueError(f"field {i}: expected a contiguous cuda float32 tensor of {self.n_gates} gates "
                                 f"on {self.packed.device}")
        if len(masks) != self.n_fields:
            raise ValueError("masks must have one entry (tensor or None) per field")
        for i, m in enumerate(list(masks) + [shared_mask]):
            if m is not None and not (m.is_cuda and m.device == self.packed.device and m.dtype == torch.uint8
                                      and m.is_contiguous() and m.numel() == self.n_gates):
                raise ValueError(f"mask {i}: expected a contiguous cuda uint8 tensor of {self.n_gates} gates")

    def pack(self, fields: Sequence, masks: Optional[Sequence] = None, shared_mask=None) -> None:
        """``rg_pack_fields_f32``: fold every field's exclusion mask into its values, interleave gate-major."""
        masks = [None] * self.n_fields if masks is None else list(masks)
        self._check_fields(fields, masks, shared_mask)
        nf = self.n_fields
        fptrs = (ctypes.c_void_p * nf)(*[_native.ptr(f) for f in fields])
        mptrs = (ctypes.c_void_p * nf)(*[_native.ptr(m) for m in masks])
        _native.check(self.lib.rg_pack_fields_f32(nf, fptrs, mptrs, _native.ptr(shared_mask), self.n_gates, self.stride,
                                                  _native.ptr(self.packed), _native.stream_ptr()), "rg_pack_fields_f32")

    def apply(self, out, fill_value: float = np.nan) -> None:
        """One pass over the CSR for all packed fields -> ``out[F, n_vox]`` (``rg_csr_compact_apply_packed_f32`` /
        ``rg_csr_compact_apply_f32`` through the compact copy, ``rg_csr_apply_f32`` otherwise)."""
        csr = self.csr
        nz, ny, nx = self.grid_shape
        if self.compact is not None and self.packed_stream and (self.tile in (0, 384, 576, 768) or self.tile >= 2000
                                                                or csr.weights is None):
            c = self.compact
            _native.check(self.lib.rg_csr_compact_apply_packed_f32(
                _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.rec), _native.ptr(c.rec_ptr), c.rec_order, c.w_base,
                _native.ptr(c.dict_ptr), _native.ptr(c.dict), self.n_vox, csr.n_pairs, nx, ny, _native.ptr(self.packed),
                self.n_fields, self.stride, self.n_gates, float(np.float32(fill_value)), _native.ptr(out), self.window,
                self.tile if (self.tile in (384, 576, 768) or self.tile >= 2000) else 0, _native.stream_ptr()),
                "rg_csr_compact_apply_packed_f32")
            return
        if self.compact is not None:
            c = self.compact
            _native.check(self.lib.rg_csr_compact_apply_f32(
                _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.local_idx), _native.ptr(csr.weights),
                _native.ptr(c.dict_ptr), _native.ptr(c.dict), self.n_vox, csr.n_pairs, nx, ny, _native.ptr(self.packed),
                self.n_fields, self.stride, self.n_gates, float(np.float32(fill_value)), _native.ptr(out), self.window,
                self.tile if self.tile in _PIPELINE_TILES else 0, _native.stream_ptr()), "rg_csr_compact_apply_f32")
            return
        _native.check(self.lib.rg_csr_apply_f32_ex(
            _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(csr.gate_indices), _native.ptr(csr.weights),
            self.n_vox, csr.n_pairs, nx, _native.ptr(self.packed), self.n_fields, self.stride, self.n_gates,
            float(np.float32(fill_value)), _native.ptr(out), self.tile if self.tile in _PIPELINE_TILES else 0,
            _native.stream_ptr()), "rg_csr_apply_f32")

    # ---- column mode of the row-wise kernel (rg_csr_compact_apply_columns_f32) ----------------------------------------
    @property
    def has_columns_kernel(self) -> bool:
        """The column mode of the row-wise kernel reads the packed records: 1-4 fields, codable weights."""
        return self.compact is not None and self.packed_stream and self.n_fields <= 4

    def _column_plan(self, z_pieces: int):
        """``(z_pieces, order tensor)`` for this geometry: enough workgroups to fill the chip (a workgroup is one column
        of chunks, or one of ``z_pieces`` level ranges of it), listed heaviest first -- workgroups are handed out in
        ``blockIdx`` order as slots free up, so starting the long columns first is what keeps the tail short (greedy
        longest-processing-time scheduling).  Cached on the compact copy."""
        torch = _native.torch_mod()
        c = self.compact
        nz, ny, nx = self.grid_shape
        nsx, nyg, _ = c.layout(self.grid_shape)
        n_cols = nsx * nyg
        if z_pieces <= 0:
            z_pieces = max(1, min(nz, -(-_COLUMNS_MIN_WORKGROUPS // max(n_cols, 1))))
        z_pieces = max(1, min(int(z_pieces), nz))
        cache = c._column_orders
        order = cache.get(z_pieces)
        if order is None:
            if c.chunk_pairs is not None and c.chunk_pairs.numel() == nz * nyg * nsx:
                cp = c.chunk_pairs.view(nz, nyg, nsx)
                yg = torch.arange(nyg, device=cp.device)[:, None]
                col = torch.arange(nsx, device=cp.device)[None, :]
                sx = (col + (yg * _native.RG_COMPACT_ROTATION) % nsx) % nsx            # the kernel's column -> segment map
                parts = []
                for p in range(z_pieces):
                    z0, z1 = p * nz // z_pieces, (p + 1) * nz // z_pieces
                    parts.append(torch.gather(cp[z0:z1].sum(dim=0), 1, sx).reshape(-1))
                weight = torch.cat(parts)
                order = torch.argsort(weight, descending=True, stable=True).to(torch.int32)
            else:
                order = False                                        # no statistics: identity order
            cache[z_pieces] = order
        return z_pieces, (None if order is False else order)

    def apply_columns(self, out=None, fill_value: float = np.nan, level_planes=None, keep_lo: int = 0, col_max=None,
                      col_arg=None, col_window=None, z_pieces: int = 0, lanes_hint: int = 0, ordered: bool = True) -> None:
        """One pass of ``rg_csr_compact_apply_columns_f32`` over the packed records (the row-wise kernel with every workgroup
        walking the levels of one column of chunks): ``out`` ``[F, n_vox]`` receives the same bits as :meth:`apply` with the
        row-wise kernel, or is ``None`` when only 2-D products are wanted -- ``level_planes`` ``[F, n_keep, ny, nx]``
        (planes ``keep_lo ..`` of every grid), ``col_max`` / ``col_arg`` ``[F, ny, nx]`` (``column_argmax`` over the level
        window ``col_window = (lo, hi)``, default all levels)."""
        torch = _native.torch_mod()
        if not self.has_columns_kernel:
            raise _native.NativeError("the column mode needs the packed records (1-4 fields, codable weights)")
        csr, c = self.csr, self.compact
        nz, ny, nx = self.grid_shape
        n_keep = 0 if level_planes is None else int(level_planes.shape[1])
        lo, hi = (0, nz - 1) if col_window is None else (int(col_window[0]), int(col_window[1]))
        pieces, order = self._column_plan(z_pieces)
        if not ordered:
            order = None
        ws = None
        if col_max is not None and pieces > 1:
            nbytes = int(self.lib.rg_csr_columns_workspace_bytes(ny, nx, self.n_fields, pieces))
            ws = getattr(self, "_columns_ws", None)
            if ws is None or ws.numel() < nbytes:
                ws = self._columns_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        window = self.window
        _native.check(self.lib.rg_csr_compact_apply_columns_f32(
            _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.rec), _native.ptr(c.rec_ptr), c.rec_order, c.w_base,
            _native.ptr(c.dict_ptr), _native.ptr(c.dict), self.n_vox, csr.n_pairs, nx, ny, _native.ptr(self.packed),
            self.n_fields, self.stride, self.n_gates, float(np.float32(fill_value)), _native.ptr(out),
            _native.ptr(level_planes), int(keep_lo), n_keep, _native.ptr(col_max), _native.ptr(col_arg), lo, hi, window,
            pieces, _native.ptr(order), _native.ptr(ws), 0 if ws is None else int(ws.numel()), int(lanes_hint),
            _native.stream_ptr()), "rg_csr_compact_apply_columns_f32")

    def columns_bytes(self, store_grid: bool, n_keep: int = 0, colmax: bool = False) -> Optional[int]:
        """Bytes one ``apply_columns`` launch must move: the compact kernel's, minus the grids that are not stored, plus the
        kept planes and the (max, arg) planes."""
        base = self.compact_bytes()
        if base is None:
            return None
        nz, ny, nx = self.grid_shape
        return (base - (0 if store_grid else self.n_fields * 4 * self.n_vox)
                + self.n_fields * 4 * ny * nx * (int(n_keep) + (2 if colmax else 0)))

    def algorithmic_bytes(self) -> int:
        """Bytes one ``apply`` launch must move: index + weight per pair, the row pointers,
        each field's values + mask once, each output grid once."""
        csr = self.csr
        ip = 8 if csr.is_i64 else 4
        return 8 * csr.n_pairs + ip * (self.n_vox + 1) + self.n_fields * (5 * self.n_gates + 4 * self.n_vox)

    def settle_records(self, tries: int = 3, probe_launches: int = 3, out=None):
        """Placement settling of the packed record array -- part of the one-off geometry build, like the reference's
        precompute (``radar_grid/compute.py:106-284`` runs once per scan strategy, ``apply_geometry`` thousands of times).

        On this part the 1.4 % of its traffic the gridding kernel WRITES costs 5-15 % of the launch depending on where the
        driver happened to place the 44 GB it READS (EXPERIMENTS.md: thirty + twenty processes in two clusters, the placement
        decided per allocation, nothing on the kernel's side moves it).  So: copy the records into ``tries - 1`` further
        allocations, time ``probe_launches`` launches of this gridder's own kernel through each (the fields are whatever
        ``packed`` holds; values do not matter to the timing), keep the fastest placement and free the others.  Needs
        ``tries`` x the record bytes of free HBM for a moment (skipped otherwise); nothing about the results changes -- the
        same records, the same kernel, the same bits.  Returns ``{"tries", "probe_ms", "kept"}`` or ``None`` when it did
        not run."""
        torch = _native.torch_mod()
        c = self.compact
        if c is None or not self.packed_stream or tries <= 1 or c.rec is None or c.rec.numel() == 0:
            return None
        nbytes = int(c.rec.numel()) * c.rec.element_size()
        free_b, _ = torch.cuda.mem_get_info(self.dev)
        if free_b < (tries - 1) * nbytes + 4 * self.n_fields * self.n_vox + (8 << 30):
            return None
        with torch.cuda.device(self.dev):
            if out is None:
                out = torch.empty((self.n_fields, self.n_vox), dtype=torch.float32, device=self.dev)
            candidates = [c.rec] + [torch.empty_like(c.rec).copy_(c.rec) for _ in range(tries - 1)]      # all alive at once:
            probe_ms = []                                                                                # distinct placements
            for cand in candidates:
                c.rec = cand
                self.apply(out)                                         # warm-up through this placement
                times = []
                for _ in range(probe_launches):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self.apply(out)
                    e1.record()
                    e1.synchronize()
                    times.append(e0.elapsed_time(e1))
                probe_ms.append(round(float(np.median(times)), 4))
            kept = int(np.argmin(probe_ms))
            c.rec = candidates[kept]
            del candidates, cand
            torch.cuda.empty_cache()                                    # hand the losers' memory back to the driver
        logger.info(f"Record placement settled: probe ms {probe_ms}, kept allocation {kept}")
        return {"tries": tries, "probe_ms": probe_ms, "kept": kept}

    def compact_bytes(self) -> Optional[int]:
        """Bytes one launch of the compact kernel must move: 16-bit position + weight per pair, the dictionaries and
        their offsets, the row pointers, every field once, every grid once (``None`` without a compact copy)."""
        if self.compact is None:
            return None
        csr, c = self.csr, self.compact
        ip = 8 if csr.is_i64 else 4
        if self.packed_stream:      # 16-byte records of three pairs + one record offset per segment
            stream = 16 * int(c.rec.shape[0]) + 8 * int(c.rec_ptr.numel())
        else:
            stream = 6 * csr.n_pairs
        return (stream + 4 * c.n_dict + 8 * int(c.dict_ptr.numel()) + ip * (self.n_vox + 1)
                + self.n_fields * (5 * self.n_gates + 4 * self.n_vox))


_COLUMNS_FUSE_MIN_FIELDS = 5        # grid_products_device(fused=None): field-volumes per pass from which the products epilogue pays
                                    # in TIME -- more than a pass can hold: never by default (see grid_products_device)
_COLUMNS_MIN_WORKGROUPS = 8192      # column kernel: cut columns into level pieces until the launch has about this many workgroups
_PIPELINE_TILES = (128, 192, 256, 320, 384, 512)   # pairs per pipeline step the tile kernels accept (0 = their default)
_COMPACT_MIN_PAIRS = 50_000_000     # below this a pass takes well under a millisecond either way
_COMPACT_MAX_WINDOW_BYTES = 24576  # LDS window beyond which the standard kernel is the faster one (CsrGridder.__init__)
_COMPACT_MAX_FALLBACK = 0.02        # share of pairs allowed on the per-pair path before the standard kernel is preferred


def _use_compact(geometry: GridGeometry, dev) -> bool:
    """Policy of ``grid_fields_device`` / ``apply_geometry``: a geometry large enough to matter grids through the compact
    copy of its CSR from its FIRST pass on, provided the memory is there -- the one-time conversion costs about as much
    as building the geometry did (15 ms per 1e9 pairs for the copy, as much again for the packed records), and deciding by
    size alone means that every pass of a geometry runs the same kernel and returns the same bits.  Once built the copy is
    always used.  ``GridGeometry.device_compact()`` builds it explicitly ahead of time."""
    cached = getattr(geometry, "_compact", None)
    if cached is not None and cached[0] is geometry.device_csr(dev):
        return cached[1] is not None
    n_pairs = geometry.device_csr(dev).n_pairs
    if n_pairs < _COMPACT_MIN_PAIRS:
        return False
    free_b, _ = _native.torch_mod().cuda.mem_get_info(dev)
    return free_b > 3.2 * n_pairs + (8 << 30)


def _fields_per_pass(geometry: GridGeometry, dev, use_compact: bool) -> int:
    """Fields one CSR pass should fuse.  Through the packed records (the row-wise kernel, 1-8 fields): 8 where the geometry's
    LDS window holds 40-byte entries (bench grid: one pass of eight 15.5 ms against two of four 2 x 10.5), 4 where it does not
    (config 2's 1792-entry window: one pass of eight 3.4-3.9 ms against 2 x 1.4); 8 (``RG_MAX_FIELDS``) through the other
    kernels."""
    csr = geometry.device_csr(dev)
    compact = None
    if csr.weights is None:                 # packed-only geometry
        compact = geometry.device_compact(dev)
    elif use_compact and csr.n_pairs:
        compact = geometry.device_compact(dev)
        if compact is not None and not compact.ensure_packed(csr):
            compact = None
    if compact is not None:
        return 8 if compact.window_cap * compact.entry_bytes(8, rowwise=True) <= 32768 else 4    # 5 workgroups per CU
    return _native.RG_MAX_FIELDS


def fields_per_pass(geometry: GridGeometry, device=None) -> int:
    """How many fields (or field-volumes of a batch) ``grid_fields_device`` fuses into one pass over this geometry on
    ``device`` (builds the device copy if it does not exist yet)."""
    dev = _native.canonical_device(device)
    return _fields_per_pass(geometry, dev, _use_compact(geometry, dev))


def volumes_per_pass_cap(geometry: GridGeometry, device=None) -> int:
    """Field-volumes of a batch one pass should fuse (``batch.VolumeBatch``): what the packed records take (8 or 4, see
    ``_fields_per_pass``); 4 through ``rg_csr_apply_f32`` (bench grid: 20.7 ms for four, 55.1 for eight)."""
    dev = _native.canonical_device(device)
    use_compact = _use_compact(geometry, dev)
    csr = geometry.device_csr(dev)
    if csr.weights is None or (use_compact and csr.n_pairs and geometry.device_compact(dev) is not None
                               and geometry.device_compact(dev).ensure_packed(csr)):
        return _fields_per_pass(geometry, dev, use_compact)
    return min(4, _native.RG_MAX_FIELDS)


def _cached_gridder(geometry: GridGeometry, n_gates: int, n_fields: int, dev, compact: bool) -> "CsrGridder":
    """One :class:`CsrGridder` (and its packed-field staging buffer) per (device CSR, gate count, field count, kernel)
    of a geometry, kept on the geometry object: repeated ``apply_geometry`` calls allocate nothing.  The cache dies with
    the device copy (``GridGeometry.invalidate_device`` / attribute assignment)."""
    csr = geometry.device_csr(dev)
    cache = geometry.__dict__.setdefault("_gridders", {})
    key = (id(csr), int(n_gates), int(n_fields), bool(compact))
    gridder = cache.get(key)
    if gridder is None or gridder.csr is not csr:
        if len(cache) >= 8:                 # a handful of shapes per geometry is the norm; never grow without bound
            cache.clear()
        gridder = cache[key] = CsrGridder(geometry, n_gates, n_fields, device=dev, compact=compact)
    return gridder


def grid_fields_device(geometry: GridGeometry, fields: Sequence, masks: Optional[Sequence] = None,
                       shared_mask=None, fill_value: float = np.nan, out=None):
    """Grid ``len(fields)`` device-resident fields with one CSR pass per group of up to 8.

    Parameters
    ----------
    fields : sequence of cuda float32 tensors ``[G]``
    masks : optional sequence (same length) of uint8 tensors ``[G]`` or ``None`` (``1`` = gate excluded)
    shared_mask : optional uint8 tensor OR-ed into every field's mask (e.g. one QC GateFilter for all)
    out : optional float32 tensor ``[len(fields), nz, ny, nx]`` to write into

    Returns the ``[F, nz, ny, nx]`` float32 tensor (device).
    """
    torch = _native.torch_mod()
    n_fields = len(fields)
    if n_fields == 0:
        raise ValueError("no fields to grid")
    dev = fields[0].device
    if dev.type != "cuda":
        raise _native.NativeUnavailable("grid_fields_device needs device-resident (cuda) tensors")
    if masks is None:
        masks = [None] * n_fields
    if len(masks) != n_fields:
        raise ValueError("masks must have one entry (tensor or None) per field")
    nz, ny, nx = (int(s) for s in geometry.grid_shape)
    n_vox = nz * ny * nx
    if out is None:
        out = torch.empty((n_fields, nz, ny, nx), dtype=torch.float32, device=dev)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n_fields * n_vox):
        raise ValueError("out must be a contiguous cuda float32 tensor of shape [F, nz, ny, nx]")
    n_gates = int(fields[0].numel())
    with torch.cuda.device(dev):
        use_compact = _use_compact(geometry, dev)
        per_pass = _fields_per_pass(geometry, dev, use_compact)
        for f0 in range(0, n_fields, per_pass):
            f1 = min(n_fields, f0 + per_pass)
            gridder = _cached_gridder(geometry, n_gates, f1 - f0, dev, compact=use_compact)
            gridder.pack(fields[f0:f1], masks[f0:f1], shared_mask)
            gridder.apply(out.view(n_fields, n_vox)[f0:f1], fill_value)
    return out.view(n_fields, nz, ny, nx)


class PlaneProducts:
    """The 2-D products a products-only pass keeps of every gridded field (``grid_products_device``,
    ``batch.VolumeBatch.grid_shard(products=PlaneProducts(...))``): the column maximum over a level window
    (``radar_grid/products.py:420-490``; same window arguments as ``column_max``), optionally the level that attains it
    (``column_argmax``), and CAPPIs at the given altitudes (``products.py:317-415``).  With these and nothing else wanted,
    the 3-D grid never has to exist in HBM: the gridding kernel, walking grid columns, keeps the running maximum in registers
    and stores only the levels the CAPPIs blend."""

    def __init__(self, colmax: bool = True, argmax: bool = True, cappi: Sequence[float] = (), interpolation: str = "linear",
                 z_min_idx: Optional[int] = None, z_max_idx: Optional[int] = None, z_min_alt: Optional[float] = None,
                 z_max_alt: Optional[float] = None, fused: Optional[bool] = None):
        """``fused``: ``True`` = take the planes out of the gridding kernel (no 3-D grid in HBM: the memory-saving way),
        ``False`` = grid, then reduce with the separate kernels, ``None`` = whichever the build measured faster (the same
        planes either way, bit for bit)."""
        if interpolation not in ("linear", "nearest"):
            raise ValueError(f"Unknown interpolation method: {interpolation}")
        self.colmax = bool(colmax or argmax)
        self.argmax = bool(argmax)
        self.cappi = tuple(float(a) for a in cappi)
        self.interpolation = interpolation
        self.window = (z_min_idx, z_max_idx, z_min_alt, z_max_alt)
        self.fused = fused


def grid_products_device(geometry: GridGeometry, fields: Sequence, masks: Optional[Sequence] = None, shared_mask=None,
                         products: Optional[PlaneProducts] = None, fill_value: float = np.nan, fused: Optional[bool] = None):
    """Grid device-resident fields and return ONLY 2-D products: a list with one ``dict`` per field --
    ``{"colmax": [ny, nx] float32, "argmax": [ny, nx] int32, "cappi": {altitude: [ny, nx] float32}}`` (keys present as
    requested by ``products``).  The planes are bit-identical to ``column_argmax`` / ``constant_altitude_ppi`` applied to
    the grid ``grid_fields_device`` returns for the same pass.

    ``fused=True``: on large geometries (packed records present) every group of up to four field-volumes runs as ONE launch
    of the row-wise kernel in column mode with its products epilogue -- no 3-D grid is written or read back (640 MB each
    way per field on the bench grid), so a pass needs no grid memory at all; it takes about as long as gridding + reducing
    separately (measured, see below).  Default (``fused=None``) and ``fused=False``: grid as usual, reduce with the separate
    kernels."""
    from . import grid_products as gp
    torch = _native.torch_mod()
    products = products if products is not None else PlaneProducts()
    if fused is None:
        fused = products.fused
    n_fields = len(fields)
    if n_fields == 0:
        raise ValueError("no fields to grid")
    dev = fields[0].device
    if dev.type != "cuda":
        raise _native.NativeUnavailable("grid_products_device needs device-resident (cuda) tensors")
    if masks is None:
        masks = [None] * n_fields
    nz, ny, nx = (int(s) for s in geometry.grid_shape)
    lo, hi = gp._level_window(nz, *products.window, geometry)
    if products.colmax and lo > hi:
        raise ValueError(f"empty level window [{lo}, {hi}]")
    plans = {alt: gp.cappi_plan(geometry.grid_limits[0], nz, alt, products.interpolation) for alt in products.cappi}
    for alt, plan in plans.items():
        if plan[0] == "outside":
            z_min, z_max = geometry.grid_limits[0]
            gp.logger.warning(f"Altitude {alt}m is outside grid range [{z_min}, {z_max}]m")
    needed = sorted({k for plan in plans.values() if plan[0] != "outside" for k in ((plan[1], plan[1] + 1) if plan[0] == "blend"
                                                                                    else (plan[1],))})
    keep_lo, n_keep = (needed[0], needed[-1] - needed[0] + 1) if needed else (0, 0)
    n_gates = int(fields[0].numel())
    results = []
    with torch.cuda.device(dev):
        use_compact = _use_compact(geometry, dev)
        per_pass = _fields_per_pass(geometry, dev, use_compact)
        if fused:                                # the products epilogue (column mode of the row-wise kernel) takes 1-4 fields
            per_pass = min(per_pass, 4)
        for f0 in range(0, n_fields, per_pass):
            f1 = min(n_fields, f0 + per_pass)
            nf = f1 - f0
            gridder = _cached_gridder(geometry, n_gates, nf, dev, compact=use_compact)
            gridder.pack(fields[f0:f1], masks[f0:f1], shared_mask)
            # measured (profiles/r04_columns_variants_*.json, r04_nf4_lds_rowsums.json): walking columns costs what the store it
            # saves costs -- 8.2 vs 8.1 ms for one field, 10.7 vs 10.5 for three, 11.4 vs 11.2 for four on the bench grid --
            # so the epilogue is the MEMORY-saving choice (no F x 640 MB of grid) and is taken on request, not by default
            run_fused = (gridder.has_columns_kernel and (nf >= _COLUMNS_FUSE_MIN_FIELDS if fused is None else bool(fused)))
            if run_fused:
                cmax = torch.empty((nf, ny, nx), dtype=torch.float32, device=dev) if products.colmax else None
                carg = torch.empty((nf, ny, nx), dtype=torch.int32, device=dev) if products.argmax else None
                planes = torch.empty((nf, n_keep, ny, nx), dtype=torch.float32, device=dev) if n_keep else None
                if cmax is None and planes is None:      # nothing but out-of-range CAPPIs
                    pass
                else:
                    gridder.apply_columns(out=None, fill_value=fill_value, level_planes=planes, keep_lo=keep_lo, col_max=cmax,
                                          col_arg=carg, col_window=(lo, hi))
                level = (lambda k, z: planes[k, z - keep_lo])
            else:
                grids = torch.empty((nf, nz, ny, nx), dtype=torch.float32, device=dev)
                gridder.apply(grids.view(nf, -1), fill_value)
                cmax = carg = None
                level = (lambda k, z: grids[k, z])
            for k in range(nf):
                rec = {}
                if products.colmax:
                    if run_fused:
                        rec["colmax"] = cmax[k]
                        if products.argmax:
                            rec["argmax"] = carg[k]
                    else:
                        got = gp._column("max", grids[k], lo, hi, None, None, None, want_arg=products.argmax)
                        rec["colmax"], rec["argmax"] = got if products.argmax else (got, None)
                        if not products.argmax:
                            del rec["argmax"]
                if products.cappi:
                    rec["cappi"] = {}
                    for alt, plan in plans.items():
                        if plan[0] == "outside":
                            rec["cappi"][alt] = torch.full((ny, nx), float("nan"), dtype=torch.float32, device=dev)
                        elif plan[0] == "level":
                            rec["cappi"][alt] = level(k, plan[1])
                        else:
                            out = torch.empty((ny, nx), dtype=torch.float32, device=dev)
                            lo_plane = level(k, plan[1])      # levels k and k + 1 are adjacent planes of one buffer
                            _native.check(gridder.lib.rg_cappi_lerp_f32(_native.ptr(lo_plane), ny * nx, 0,
                                                                        float(np.float32(plan[2])), float(np.float32(plan[3])),
                                                                        _native.ptr(out), _native.stream_ptr()), "rg_cappi_lerp_f32")
                            rec["cappi"][alt] = out
                results.append(rec)
    return results


def _to_host(t) -> np.ndarray:
    """Device -> NumPy through a page-locked staging tensor (about twice the rate of a pageable copy; the
    640 MB grid download dominates the NumPy-in / NumPy-out path).  The array owns its buffer."""
    torch = _native.torch_mod()
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:        # no pinned memory available: plain copy
        return t.cpu().numpy()
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def apply_geometry(geometry: GridGeometry, field_data: np.ndarray,
                   additional_filters: Optional[List[GateFilter]] = None,
                   fill_value: float = np.nan) -> np.ndarray:
    """Weighted-mean gridding of one field (``radar_grid/interpolate.py:15-104``).

    ``field_data``: flattened (masked) field, shape ``(n_gates,)``; masked gates and gates excluded by any of
    ``additional_filters`` contribute to neither sum; voxels without a positive weight sum get ``fill_value``.
    Returns a float32 array of shape ``geometry.grid_shape``.
    """
    filters = _coerce_filters(additional_filters)
    values, mask = _host_field(field_data, filters)
    torch = _native.torch_mod()
    dev = _native.device()
    f_t = torch.from_numpy(values).to(dev)
    m_t = torch.from_numpy(mask).to(dev) if mask.any() else None
    grid = grid_fields_device(geometry, [f_t], [m_t], fill_value=fill_value)
    return _to_host(grid[0]).reshape(geometry.grid_shape)


def apply_geometry_multi(geometry: GridGeometry, fields: Dict[str, np.ndarray],
                         additional_filters: Optional[Dict[str, List[GateFilter]]] = None,
                         fill_value: float = np.nan) -> Dict[str, np.ndarray]:
    """Grid several fields (``radar_grid/interpolate.py:107-142``) -- one CSR pass for all of them.

    ``additional_filters`` maps a field name to its filter list (a missing name means no filters).
    """
    if additional_filters is None:
        additional_filters = {}
    names = list(fields.keys())
    if not names:
        return {}
    torch = _native.torch_mod()
    dev = _native.device()
    f_ts, m_ts = [], []
    for name in names:
        values, mask = _host_field(fields[name], _coerce_filters(additional_filters.get(name, None)))
        f_ts.append(torch.from_numpy(values).to(dev))
        m_ts.append(torch.from_numpy(mask).to(dev) if mask.any() else None)
    grid = _to_host(grid_fields_device(geometry, f_ts, m_ts, fill_value=fill_value))
    return {name: grid[i].reshape(geometry.grid_shape) for i, name in enumerate(names)}
