"""CSR geometry container and its ``.npz`` persistence -- host-side mirror of
``radar_grid/geometry.py`` (``GridGeometry`` :14-91, ``save_geometry`` :94-118, ``load_geometry`` :121-150).

Same constructor arguments, attributes, helper methods, ``__repr__`` text and on-disk keys as the reference,
so geometry files are interchangeable.  What is new is where the arrays live: the CSR can be *device
resident* (built by the GPU builder, or uploaded once and cached), because at the sizes MI355X is meant for
(tens of GB of pairs, int64 row pointers) a host copy is neither needed nor wanted.  Host views
(``.indptr`` / ``.gate_indices`` / ``.weights``) are materialised lazily on first access.
"""
from __future__ import annotations

import logging
import os
from typing import Optional, Tuple

import numpy as np

from . import _native

logger = logging.getLogger("radar_grid.geometry")  # same logger name as the reference (docs/LOGGING.md:128-141)

_INT32_MAX = np.iinfo(np.int32).max
_NOT_COMPACTABLE = 0x40000000     # chunk_counts marker of rg_csr_compact_count

# Order in which the packed records of a geometry are stored (include/radargrid_hip.h: RG_REC_ORDER_*).  DISPATCH = the
# order in which the apply kernels' workgroups read them, so that a launch sweeps the array as one moving front;
# SEGMENT = line-major segments (round 2's layout, kept for A/B measurements: bench.py --rec-order segment).
DEFAULT_REC_ORDER = _native.RG_REC_ORDER_DISPATCH


class DeviceCSR:
    """CSR arrays resident in HBM (torch tensors used purely as allocations)."""

    __slots__ = ("indptr", "gate_indices", "weights", "n_vox", "n_pairs", "max_gate", "is_i64")

    def __init__(self, indptr, gate_indices, weights, max_gate: int, n_pairs: Optional[int] = None):
        self.indptr = indptr
        self.gate_indices = gate_indices   # None when the geometry was built compact-only (see CompactCSR)
        self.weights = weights             # None when it was built packed-only (weights live in CompactCSR.rec)
        self.n_vox = int(indptr.shape[0]) - 1
        self.n_pairs = int(weights.shape[0]) if weights is not None else int(n_pairs)
        self.max_gate = int(max_gate)   # largest gate index referenced (-1 when there are no pairs)
        self.is_i64 = indptr.dtype == _native.torch_mod().int64

    def nbytes(self) -> int:
        return sum(int(t.numel()) * t.element_size() for t in (self.indptr, self.gate_indices, self.weights)
                   if t is not None)


class CompactCSR:
    """Compact device copy of a :class:`DeviceCSR` for ``rg_csr_compact_apply_f32``.  Rows are grouped in chunks that
    cover a 2-D patch of the grid -- segment ``sx`` (rows ``[64*sx, 64*sx+64)`` of a grid line) of
    ``RG_COMPACT_LINES`` consecutive lines of one plane, chunk number ``(plane*nyg + yg)*nsx + sx`` --, each chunk's
    distinct gates are listed once (``dict`` / ``dict_ptr``) and every pair stores a 16-bit position in that list
    (``local_idx``).  A chunk with more than 65536 distinct gates (the patch around the radar on a dense scan) is
    *split*: one dictionary per wavefront behind a header of ``RG_COMPACT_LINES`` offsets, recognisable by an entry count
    above 65536.  ``indptr`` and ``weights`` are shared with the standard CSR."""

    __slots__ = ("local_idx", "dict_ptr", "dict", "n_dict", "max_dict", "window_cap", "grid_shape", "chunk_pairs",
                 "chunk_counts", "rec", "rec_ptr", "rec_order", "w_base", "_pack_tried", "_column_orders")

    def __init__(self, local_idx, dict_ptr, dict_, max_dict: int, window_cap: int, grid_shape, chunk_pairs=None,
                 chunk_counts=None):
        self.local_idx = local_idx          # int16 storage of the uint16 positions [P]
        self.dict_ptr = dict_ptr            # int64 [chunks + 1]
        self.dict = dict_                   # int32 [D]
        self.n_dict = int(dict_.shape[0])
        self.max_dict = int(max_dict)
        self.window_cap = int(window_cap)   # LDS window (entries) that covers 99.9 % of the pairs (speed only)
        self.grid_shape = tuple(int(v) for v in grid_shape)   # (planes, lines per plane, rows per line)
        self.chunk_pairs = chunk_pairs      # int64 [chunks]: pairs per chunk    } kept to choose the window for a
        self.chunk_counts = chunk_counts    # int64 [chunks]: distinct gates     } given field count (window_for)
        # packed pair stream for passes of 1-8 fields (ensure_packed): 16-byte records of three pairs, or None
        self.rec = None                     # int32 [n_rec, 4]
        self.rec_ptr = None                 # int64 [slots + 1]: records of the segment in slot s (see rec_order)
        self.rec_order = DEFAULT_REC_ORDER  # RG_REC_ORDER_SEGMENT (slot = segment) / RG_REC_ORDER_DISPATCH (slot_of_segments)
        self.w_base = 0                     # weight code = float32 bits - w_base
        self._pack_tried = False
        self._column_orders = {}            # z_pieces -> workgroup order of the column kernel (gridding.CsrGridder)

    def nbytes(self) -> int:
        return sum(int(t.numel()) * t.element_size()
                   for t in (self.local_idx, self.dict_ptr, self.dict, self.rec, self.rec_ptr) if t is not None)

    def ensure_packed(self, csr: "DeviceCSR") -> bool:
        """Build (once) the packed pair stream ``rg_csr_compact_apply_packed_f32`` reads: positions and weights of three
        consecutive pairs of a segment in one 16-byte record -- 5.33 instead of 6 bytes per pair, one 16-byte load per
        lane.  The weight is stored as its float32 bits minus ``w_base`` in 26 bits, which is lossless exactly when all
        weights are positive and span at most 8 binades (Barnes weights: exp(-4)+1e-5 .. 1+1e-5, 7 binades; a uniform
        weight trivially); other geometries (Cressman: weights down to 0) keep the plain arrays.  Returns whether the
        stream exists.  Costs 5.4 bytes per pair of HBM on top of the copy; skipped when that does not fit."""
        if self.rec is not None or self._pack_tried:
            return self.rec is not None
        self._pack_tried = True
        torch = _native.torch_mod()
        lib = _native.load_library()
        dev = csr.indptr.device
        if csr.n_pairs == 0 or self.local_idx is None or csr.weights is None:
            return False
        free_b, _ = torch.cuda.mem_get_info(dev)
        if free_b < 5.6 * csr.n_pairs + (6 << 30):
            return False
        # codable?  exponent range of the weights (float32 bits >> 23; a sign bit would show up as an exponent >= 256)
        lo, hi = None, None
        step = 1 << 28
        for p0 in range(0, csr.n_pairs, step):
            e = csr.weights[p0:p0 + step].view(torch.int32) >> 23
            a, b = int(e.min()), int(e.max())
            lo, hi = (a if lo is None else min(lo, a)), (b if hi is None else max(hi, b))
        base = lo & ~7          # the code's base exponent: a multiple of 8, so that the kernels can OR it back in
        if lo <= 0 or hi - base > 7 or hi >= 255:
            logger.info(f"weights span exponents {lo}..{hi}: not codable in 26 bits above a base exponent that is a multiple "
                        "of 8, the compact copy keeps the plain arrays")
            return False
        nz, ny, nx = self.grid_shape
        rec_ptr = self.record_pointers(csr.indptr, self.grid_shape, self.rec_order)
        if rec_ptr is None:                      # the kernels address a segment's records with 32-bit byte offsets
            logger.info("a segment holds more than 2^27 records: the compact copy keeps the plain arrays")
            return False
        n_rec = int(rec_ptr[-1])
        rec = torch.empty((max(n_rec, 1), 4), dtype=torch.int32, device=dev)[:n_rec]
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        w_base = base << 23
        with torch.cuda.device(dev):
            _native.check(lib.rg_csr_compact_pack(_native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(self.local_idx),
                                                  _native.ptr(csr.weights), csr.n_vox, nx, ny, _native.ptr(rec_ptr),
                                                  self.rec_order, 0, w_base, _native.ptr(rec), _native.ptr(err),
                                                  _native.stream_ptr()),
                          "rg_csr_compact_pack")
        if int(err.item()):
            raise _native.NativeError(f"rg_csr_compact_pack reported flag {int(err.item())}")
        self.rec, self.rec_ptr, self.w_base = rec, rec_ptr, w_base
        logger.info(f"Packed pair stream: {rec.numel() * 4 / 1e6:.1f} MB ({16 * n_rec / csr.n_pairs:.2f} bytes per pair)")
        return True

    @classmethod
    def slot_of_segments(cls, line, sx, grid_shape, rec_order: int):
        """Slot (index into ``rec_ptr``) of segment ``sx`` of grid line ``line`` (int64 tensors, lines counted through all
        planes).  ``RG_REC_ORDER_SEGMENT``: the line-major segment number.  ``RG_REC_ORDER_DISPATCH``: ``block * H + w``
        for the wavefront ``w`` of the workgroup ``block`` that reads the segment -- the inverse of the kernels'
        block -> chunk rotation (``block_chunk`` in csrc/rg_csr_compact.hip, 32-bit unsigned arithmetic)."""
        nz, ny, nx = (int(v) for v in grid_shape)
        nsx, nyg, _ = cls.layout(grid_shape)
        if rec_order == _native.RG_REC_ORDER_SEGMENT:
            return line * nsx + sx
        lines = _native.RG_COMPACT_LINES
        plane = line // ny
        y = line - plane * ny
        grp = plane * nyg + y // lines
        shift = ((grp * _native.RG_COMPACT_ROTATION) & 0xFFFFFFFF) % nsx
        col = (sx - shift) % nsx                     # the block column whose rotated column is sx
        return (grp * nsx + col) * lines + y % lines

    @classmethod
    def record_pointers(cls, indptr, grid_shape, rec_order: int):
        """``rec_ptr`` (int64 ``[slots + 1]``) for the row pointers of a whole grid: every segment gets
        ``ceil(pairs / 3)`` records, laid out in slot order.  ``None`` when a segment would hold 2^27 records or more."""
        torch = _native.torch_mod()
        dev = indptr.device
        nz, ny, nx = (int(v) for v in grid_shape)
        nsx, nyg, n_chunks = cls.layout(grid_shape)
        starts = torch.tensor(cls.segment_starts(nx), device=dev, dtype=torch.int64)
        line0 = torch.arange(nz * ny, device=dev, dtype=torch.int64) * nx
        edges = indptr[(line0[:, None] + starts[None, :]).reshape(-1)].to(torch.int64).view(nz * ny, nsx + 1)
        n_rec_seg = ((edges[:, 1:] - edges[:, :-1]) + 2) // 3
        if n_rec_seg.numel() and int(n_rec_seg.max()) >= 1 << 27:
            return None
        if rec_order == _native.RG_REC_ORDER_SEGMENT:
            per_slot = n_rec_seg.reshape(-1)
        else:
            line = torch.arange(nz * ny, device=dev, dtype=torch.int64)[:, None].expand(nz * ny, nsx)
            sx = torch.arange(nsx, device=dev, dtype=torch.int64)[None, :].expand(nz * ny, nsx)
            per_slot = torch.zeros(n_chunks * _native.RG_COMPACT_LINES, dtype=torch.int64, device=dev)
            per_slot[cls.slot_of_segments(line, sx, grid_shape, rec_order).reshape(-1)] = n_rec_seg.reshape(-1)
        rec_ptr = torch.zeros(per_slot.numel() + 1, dtype=torch.int64, device=dev)
        rec_ptr[1:] = torch.cumsum(per_slot, 0)
        return rec_ptr

    @staticmethod
    def layout(grid_shape):
        """``(nsx, nyg, n_chunks)`` of a grid: segments per line, line groups per plane, chunks."""
        nz, ny, nx = (int(v) for v in grid_shape)
        nsx = (nx + 63) // 64
        nyg = (ny + _native.RG_COMPACT_LINES - 1) // _native.RG_COMPACT_LINES
        return nsx, nyg, nz * nyg * nsx

    @staticmethod
    def segment_starts(nx: int):
        """First row of each of a line's ``ceil(nx / 64)`` segments (+ ``nx`` as the last entry).  The segments are
        balanced -- the first ``nx % nsx`` hold one row more than the others -- exactly as the kernels cut them."""
        nsx = (int(nx) + 63) // 64
        base, extra = divmod(int(nx), nsx)
        return [sx * base + min(sx, extra) for sx in range(nsx + 1)]

    @staticmethod
    def chunk_of_rows(rows, grid_shape):
        """Chunk number of every (flat, int64 tensor) row index."""
        nz, ny, nx = (int(v) for v in grid_shape)
        nsx, nyg, _ = CompactCSR.layout(grid_shape)
        base, extra = divmod(nx, nsx)
        line = rows // nx
        x = rows - line * nx
        plane = line // ny
        y = line - plane * ny
        split = extra * (base + 1)                  # rows covered by the longer segments
        sx = (x // (base + 1)).where(x < split, extra + (x - split) // max(base, 1))
        return (plane * nyg + y // _native.RG_COMPACT_LINES) * nsx + sx

    @staticmethod
    def entry_bytes(n_fields: int, rowwise: bool = False) -> int:
        """Bytes of one LDS window entry of the compact kernels.  Tile kernels: the packed slots of a gate (1, 2, 4 or 8
        floats; a 3-field entry without its padding slot).  Row-wise kernel (``rowwise``): one field the value and a 0/1
        factor, two fields the two values, three and more the values (v' = value or +0) plus one mask byte per field -- 16,
        20 and 40 bytes for 3, 4 and 5-8 fields (``csrc/rg_compact_layout.hpp``: rowwise_entry_words)."""
        if rowwise and n_fields >= 3:
            return 16 if n_fields == 3 else 20 if n_fields == 4 else 40
        if n_fields > 4:
            return 32
        return 4 * (2 if n_fields <= 2 else 3 if n_fields == 3 else 4)

    def window_for(self, n_fields: int, lds_budget_bytes: Optional[int] = None, rowwise: bool = False) -> int:
        """LDS window (entries) for a pass of ``n_fields`` fields: the geometry's 99.9 % window if its entries fit
        ``lds_budget_bytes`` (default 32 KiB; 48 KiB for the row-wise kernel's 3-8 field entries, which carry mask bytes and
        have no tile next to them), else the largest that does."""
        if lds_budget_bytes is None:
            lds_budget_bytes = 49152 if (rowwise and n_fields >= 3) else 32768
        room = max(0, lds_budget_bytes // self.entry_bytes(n_fields, rowwise)) // 64 * 64
        return int(min(self.window_cap, room))

    def fallback_fraction(self, window: int) -> float:
        """Share of the pairs whose chunk holds more distinct gates than ``window`` (they gather per pair)."""
        if self.chunk_pairs is None or self.chunk_pairs.numel() == 0:
            return 0.0
        total = max(int(self.chunk_pairs.sum()), 1)
        return float(int(self.chunk_pairs[self.chunk_counts > window].sum()) / total)

    @classmethod
    def build(cls, csr: "DeviceCSR", grid_shape) -> Optional["CompactCSR"]:
        """Derive the compact copy on the device (``rg_csr_compact_count`` / ``rg_csr_compact_fill``: one workgroup per
        chunk with an LDS hash set of its gates); ``None`` when a chunk references more than 65536 distinct gates
        (positions are 16 bits; the standard kernel then stays in charge)."""
        torch = _native.torch_mod()
        dev = csr.indptr.device
        nz, ny, nx = (int(v) for v in grid_shape)
        if nz * ny * nx != csr.n_vox:
            raise ValueError(f"grid_shape {grid_shape} does not match the CSR's {csr.n_vox} rows")
        local = torch.empty(max(csr.n_pairs, 1), dtype=torch.int16, device=dev)[:csr.n_pairs]
        with torch.cuda.device(dev):
            built = cls._planes(csr.indptr, _native.ptr(csr.gate_indices), nz, ny, nx, _native.ptr(local))
        if built is None:
            return None
        counts, dict_ = built
        return cls._finish(csr.indptr, grid_shape, local, counts, [dict_])

    @staticmethod
    def _planes(indptr, gate_idx_ptr: int, n_planes: int, ny: int, nx: int, local_ptr: int):
        """Dictionaries and positions of the chunks of ``n_planes`` whole planes whose (absolute) row pointers are
        ``indptr`` (a tensor of ``n_planes*ny*nx + 1`` entries, possibly a view into a longer one).  ``gate_idx_ptr`` /
        ``local_ptr`` are device addresses such that element ``p`` belongs to absolute pair ``p``.  Returns
        ``(counts int64 [chunks], dict int32)`` with the dictionaries back to back, or ``None`` when a chunk is too
        rich."""
        torch = _native.torch_mod()
        lib = _native.load_library()
        dev = indptr.device
        n_rows = n_planes * ny * nx
        n_chunks = int(lib.rg_csr_compact_chunks(n_rows, nx, ny)) if n_rows else 0
        if n_chunks < 0:
            raise _native.NativeError("rg_csr_compact_chunks rejected the grid shape")
        if n_chunks == 0:
            return torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
        is_i64 = int(indptr.dtype == torch.int64)
        stream = _native.stream_ptr()
        counts = torch.zeros(n_chunks + 1, dtype=torch.int32, device=dev)
        rounds = torch.empty(n_chunks, dtype=torch.uint8, device=dev)
        _native.check(lib.rg_csr_compact_count(_native.ptr(indptr), is_i64, gate_idx_ptr, n_rows, nx, ny,
                                               _native.ptr(counts), _native.ptr(rounds), stream), "rg_csr_compact_count")
        if int(counts.max()) >= _NOT_COMPACTABLE:      # a single 64-row segment references more than 65536 gates
            return None
        dict_ptr = torch.empty(n_chunks + 1, dtype=torch.int64, device=dev)
        ws_bytes = int(lib.rg_scan_workspace_bytes(n_chunks))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        _native.check(lib.rg_scan_counts_i64(_native.ptr(counts), n_chunks, _native.ptr(dict_ptr), _native.ptr(ws), ws_bytes,
                                             stream), "rg_scan_counts_i64")
        n_dict = int(dict_ptr[-1])
        dict_ = torch.empty(max(n_dict, 1), dtype=torch.int32, device=dev)[:n_dict]
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        _native.check(lib.rg_csr_compact_fill(_native.ptr(indptr), is_i64, gate_idx_ptr, n_rows, nx, ny,
                                              _native.ptr(dict_ptr), _native.ptr(rounds), _native.ptr(dict_), local_ptr,
                                              _native.ptr(err), stream), "rg_csr_compact_fill")
        flag = int(err.item())
        if flag:
            raise _native.NativeError(f"rg_csr_compact_fill: the fill pass saw different inputs than the count pass "
                                      f"(flag {flag}); were the gate indices modified during the build?")
        return counts[:n_chunks].to(torch.int64), dict_

    @classmethod
    def _finish(cls, indptr, grid_shape, local, counts, parts) -> "CompactCSR":
        torch = _native.torch_mod()
        lines = _native.RG_COMPACT_LINES
        dev = counts.device
        nz, ny, nx = (int(v) for v in grid_shape)
        nsx, nyg, n_chunks = cls.layout(grid_shape)
        assert n_chunks == int(counts.shape[0])
        dict_ptr = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
        dict_ptr[1:] = torch.cumsum(counts, 0)
        dict_ = torch.cat(parts) if parts else torch.zeros(1, dtype=torch.int32, device=dev)[:0]
        # pairs per chunk: per (line, segment) from the row pointers, summed over the lines of a group
        if n_chunks:
            seg_x = torch.tensor(cls.segment_starts(nx), device=dev, dtype=torch.int64)
            line0 = torch.arange(nz * ny, device=dev, dtype=torch.int64) * nx
            edges = indptr[(line0[:, None] + seg_x[None, :]).reshape(-1)].to(torch.int64).view(nz * ny, nsx + 1)
            seg_pairs = (edges[:, 1:] - edges[:, :-1]).view(nz, ny, nsx)
            pad = nyg * lines - ny
            if pad:
                seg_pairs = torch.cat([seg_pairs, torch.zeros((nz, pad, nsx), dtype=torch.int64, device=dev)], dim=1)
            chunk_pairs = seg_pairs.view(nz, nyg, lines, nsx).sum(dim=2).reshape(-1)
        else:
            chunk_pairs = torch.zeros(0, dtype=torch.int64, device=dev)
        # LDS window: the smallest size (multiple of 256 entries) that leaves at most 0.1 % of the pairs to the
        # per-pair fallback
        window_cap = _native.RG_COMPACT_MAX_WINDOW
        total = max(int(chunk_pairs.sum()), 1) if n_chunks else 1
        for cap in range(256, _native.RG_COMPACT_MAX_WINDOW + 1, 256):
            if int(chunk_pairs[counts > cap].sum()) <= total // 1000:
                window_cap = cap
                break
        return cls(local, dict_ptr, dict_, int(counts.max()) if n_chunks else 0, window_cap, grid_shape,
                   chunk_pairs, counts)

    def _record_fields(self, csr: "DeviceCSR", r0: int, r1: int):
        """Positions (int64) and weights (float32) of the pairs of rows ``[r0, r1)``, unpacked from the 16-byte records
        (the inverse of ``rg_csr_compact_pack``; used when the plain ``local_idx`` / ``weights`` arrays do not exist)."""
        torch = _native.torch_mod()
        dev = csr.indptr.device
        nz, ny, nx = self.grid_shape
        nsx = (nx + 63) // 64
        base, extra = divmod(nx, nsx)
        ip = csr.indptr[r0:r1 + 1].to(torch.int64)
        rows = torch.arange(r0, r1, device=dev, dtype=torch.int64)
        line = rows // nx
        x = rows - line * nx
        split = extra * (base + 1)
        sx = (x // (base + 1)).where(x < split, extra + (x - split) // max(base, 1))
        seg = self.slot_of_segments(line, sx, self.grid_shape, self.rec_order)
        x0 = (sx * base + torch.minimum(sx, torch.full_like(sx, extra)))
        seg_first_pair = csr.indptr[line * nx + x0].to(torch.int64)            # pair offset where the row's segment starts
        lens = ip[1:] - ip[:-1]
        n = int(ip[-1] - ip[0])
        row_of_pair = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), lens, output_size=n)
        pair = torch.arange(n, device=dev, dtype=torch.int64) + ip[0]
        q = pair - seg_first_pair[row_of_pair]                                 # pair's offset inside its segment
        rec = self.rec[self.rec_ptr[seg[row_of_pair]] + q // 3].to(torch.int64) & 0xFFFFFFFF      # [n, 4] as unsigned
        j = q % 3
        code = torch.where(j == 0, rec[:, 0], torch.where(j == 1, rec[:, 1], rec[:, 2])) & 0x3FFFFFF
        weights = (code + self.w_base).to(torch.int32).view(torch.float32)
        p2 = (rec[:, 0] >> 26) | ((rec[:, 1] >> 26) << 6) | (((rec[:, 2] >> 26) & 0xF) << 12)
        pos = torch.where(j == 0, rec[:, 3] & 0xFFFF, torch.where(j == 1, rec[:, 3] >> 16, p2))
        return pos, weights

    def decode_weights(self, csr: "DeviceCSR", row0: int = 0, row1: Optional[int] = None, rows_per_slab: int = 500_000):
        """float32 weights of rows ``[row0, row1)``: a view of ``csr.weights`` when it exists, else unpacked from the
        records (bit-exact: the 26-bit code is lossless)."""
        torch = _native.torch_mod()
        row1 = csr.n_vox if row1 is None else row1
        q0, q1 = int(csr.indptr[row0]), int(csr.indptr[row1])
        if csr.weights is not None:
            return csr.weights[q0:q1]
        out = torch.empty(max(q1 - q0, 1), dtype=torch.float32, device=csr.indptr.device)[:q1 - q0]
        for r0 in range(row0, row1, rows_per_slab):
            r1 = min(row1, r0 + rows_per_slab)
            p0, p1 = int(csr.indptr[r0]), int(csr.indptr[r1])
            if p1 > p0:
                out[p0 - q0:p1 - q0] = self._record_fields(csr, r0, r1)[1]
        return out

    def decode(self, csr: "DeviceCSR", row0: int = 0, row1: Optional[int] = None, rows_per_slab: int = 500_000):
        """The standard int32 gate indices of rows ``[row0, row1)`` (default: all), rebuilt from positions and
        dictionaries; device tensor of ``indptr[row1] - indptr[row0]`` entries."""
        torch = _native.torch_mod()
        dev = csr.indptr.device
        row1 = csr.n_vox if row1 is None else row1
        q0, q1 = int(csr.indptr[row0]), int(csr.indptr[row1])
        out = torch.empty(max(q1 - q0, 1), dtype=torch.int32, device=dev)[:q1 - q0]
        for r0 in range(row0, row1, rows_per_slab):
            r1 = min(row1, r0 + rows_per_slab)
            ip = csr.indptr[r0:r1 + 1].to(torch.int64)
            p0, p1 = int(ip[0]), int(ip[-1])
            if p1 == p0:
                continue
            rows = torch.arange(r0, r1, device=dev, dtype=torch.int64)
            chunk_of_row = self.chunk_of_rows(rows, self.grid_shape)
            start_of_row = self.dict_ptr[chunk_of_row]
            # split chunks (more than 65536 entries: one dictionary per wavefront behind a header of offsets)
            is_split = (self.dict_ptr[chunk_of_row + 1] - start_of_row) > 65536
            if bool(is_split.any()):
                wave = ((rows // self.grid_shape[2]) % self.grid_shape[1]) % _native.RG_COMPACT_LINES
                header = self.dict[(start_of_row + wave).clamp(max=self.n_dict - 1)].to(torch.int64)
                start_of_row = start_of_row + torch.where(is_split, header, torch.zeros_like(header))
            start_of_pair = torch.repeat_interleave(start_of_row, ip[1:] - ip[:-1], output_size=p1 - p0)
            if self.local_idx is not None:
                pos = self.local_idx[p0:p1].to(torch.int64) & 0xFFFF
            else:
                pos = self._record_fields(csr, r0, r1)[0]
            out[p0 - q0:p1 - q0] = self.dict[start_of_pair + pos]
        return out


class GridGeometry:
    """Precomputed gate -> voxel mapping in CSR form (``radar_grid/geometry.py:14-52``).

    Row ``v = (iz*ny + iy)*nx + ix`` owns ``gate_indices[indptr[v]:indptr[v+1]]`` and the matching
    ``weights``.  ``indptr`` is int32 like the reference's whenever the pair count fits, int64 otherwise
    (the reference overflows there).
    """

    def __init__(self, grid_shape: Tuple[int, int, int], grid_limits, indptr, gate_indices, weights,
                 toa: float, radar_altitude: float = 0.0):
        self.grid_shape = grid_shape
        self.grid_limits = grid_limits
        self.toa = toa
        self.radar_altitude = radar_altitude
        self._indptr = indptr
        self._gate_indices = gate_indices
        self._weights = weights
        self._dev: Optional[DeviceCSR] = None

    # ---- construction from device-resident arrays (GPU builder) -----------------------------------
    @classmethod
    def from_device(cls, grid_shape, grid_limits, csr: DeviceCSR, toa: float, radar_altitude: float = 0.0,
                    compact: Optional["CompactCSR"] = None):
        g = cls(grid_shape, grid_limits, None, None, None, toa, radar_altitude)
        g._dev = csr
        if compact is not None:
            g._compact = (csr, compact)
        elif csr.gate_indices is None:
            raise ValueError("a CSR without gate indices needs its compact copy")
        return g

    # ---- host views (lazy) -----------------------------------------------------------------------
    def _host(self, name: str):
        arr = getattr(self, "_" + name)
        if arr is None:
            if self._dev is None:
                raise AttributeError(f"GridGeometry has no {name}")
            logger.debug("copying %s to the host", name)
            dev_arr = getattr(self._dev, name)
            if dev_arr is None:      # compact-only / packed-only geometry: rebuild the reference's array from the copy
                dev_arr = (self._compact[1].decode(self._dev) if name == "gate_indices"
                           else self._compact[1].decode_weights(self._dev))
            arr = dev_arr.cpu().numpy()
            setattr(self, "_" + name, arr)
        return arr

    def _assign(self, name: str, value) -> None:
        """Plain attribute assignment, as on the reference dataclass: the other two arrays are pulled to the host
        first (a device-built geometry holds its only copy in HBM), then every device-side cache -- the resident
        CSR and its compact copy -- is dropped, so the next gridding call uploads and validates the new arrays."""
        if self._dev is not None:
            for other in ("indptr", "gate_indices", "weights"):
                if other != name:
                    self._host(other)
        setattr(self, "_" + name, value)
        self.invalidate_device()

    def invalidate_device(self) -> None:
        """Forget the device-resident copies (CSR, compact copy, cached gridders).  Call this after modifying the host
        arrays IN PLACE: ``device_csr`` otherwise keeps serving the arrays it uploaded the first time."""
        if self._dev is not None:
            for name in ("indptr", "gate_indices", "weights"):
                self._host(name)
        self._dev = None
        self.__dict__.pop("_compact", None)
        self.__dict__.pop("_gridders", None)

    @property
    def indptr(self) -> np.ndarray:
        return self._host("indptr")

    @indptr.setter
    def indptr(self, value):
        self._assign("indptr", value)

    @property
    def gate_indices(self) -> np.ndarray:
        return self._host("gate_indices")

    @gate_indices.setter
    def gate_indices(self, value):
        self._assign("gate_indices", value)

    @property
    def weights(self) -> np.ndarray:
        return self._host("weights")

    @weights.setter
    def weights(self, value):
        self._assign("weights", value)

    @property
    def is_device_resident(self) -> bool:
        return self._dev is not None

    # ---- reference helper methods (geometry.py:54-91) --------------------------------------------
    def memory_usage_mb(self) -> float:
        if self._indptr is None and self._dev is not None:
            cached = getattr(self, "_compact", None)      # a compact-only / packed-only geometry lives in its copy
            extra = cached[1].nbytes() if (cached is not None and cached[1] is not None
                                           and (self._dev.gate_indices is None or self._dev.weights is None)) else 0
            return (self._dev.nbytes() + extra) / 1e6
        return (self.indptr.nbytes + self.gate_indices.nbytes + self.weights.nbytes) / 1e6

    def n_grid_points(self) -> int:
        return int(np.prod(self.grid_shape))

    def n_pairs(self) -> int:
        if self._gate_indices is None and self._dev is not None:
            return self._dev.n_pairs
        return len(self.gate_indices)

    def avg_neighbors(self) -> float:
        return self.n_pairs() / self.n_grid_points()

    def z_levels(self) -> np.ndarray:
        nz = self.grid_shape[0]
        z_min, z_max = self.grid_limits[0]
        return np.linspace(z_min, z_max, nz)

    def z_levels_absolute(self) -> np.ndarray:
        return self.z_levels() + self.radar_altitude

    def __repr__(self) -> str:
        return (
            "GridGeometry(\n"
            f"  grid_shape={self.grid_shape},\n"
            f"  grid_limits={self.grid_limits},\n"
            f"  toa={self.toa}m,\n"
            f"  radar_altitude={self.radar_altitude}m,\n"
            f"  n_pairs={self.n_pairs():,},\n"
            f"  avg_neighbors={self.avg_neighbors():.1f},\n"
            f"  memory={self.memory_usage_mb():.1f} MB\n"
            ")"
        )

    def __eq__(self, other):  # the reference is a dataclass: field-wise equality
        if not isinstance(other, GridGeometry):
            return NotImplemented
        return (tuple(self.grid_shape) == tuple(other.grid_shape) and self.grid_limits == other.grid_limits
                and self.toa == other.toa and self.radar_altitude == other.radar_altitude
                and np.array_equal(self.indptr, other.indptr)
                and np.array_equal(self.gate_indices, other.gate_indices)
                and np.array_equal(self.weights, other.weights))

    __hash__ = None

    # ---- device residency -----------------------------------------------------------------------
    def device_compact(self, device=None) -> Optional[CompactCSR]:
        """Compact copy of the device CSR (built once, cached); ``None`` when the geometry cannot be compacted."""
        csr = self.device_csr(device)
        cached = getattr(self, "_compact", None)
        if cached is not None and cached[0] is csr:
            return cached[1]
        compact = CompactCSR.build(csr, self.grid_shape)
        self._compact = (csr, compact)
        if compact is not None:
            logger.info(f"Compact CSR copy: {compact.nbytes() / 1e6:.1f} MB, {compact.n_dict:,} dictionary entries, "
                        f"largest chunk {compact.max_dict}")
        return compact

    def device_csr(self, device=None) -> DeviceCSR:
        """CSR in HBM: uploaded (and validated) once, then cached on the object."""
        dev = _native.canonical_device(device)
        if self._dev is not None and self._dev.indptr.device == dev:
            return self._dev
        torch = _native.torch_mod()
        if self._dev is not None:   # resident on another GPU: replicate device-to-device
            src = self._dev
            if src.gate_indices is None or src.weights is None:
                raise _native.NativeError("a compact-only geometry cannot be replicated to another GPU; rebuild it there")
            self._dev = DeviceCSR(src.indptr.to(dev), src.gate_indices.to(dev), src.weights.to(dev), src.max_gate)
            return self._dev
        indptr = np.ascontiguousarray(self._indptr)
        gidx = np.ascontiguousarray(self._gate_indices)
        wts = np.ascontiguousarray(self._weights)
        n_vox = self.n_grid_points()
        if indptr.ndim != 1 or indptr.shape[0] != n_vox + 1:
            raise ValueError(f"indptr has {indptr.shape[0]} entries, expected {n_vox + 1} for grid {self.grid_shape}")
        if not np.issubdtype(indptr.dtype, np.integer):
            raise ValueError("indptr must be an integer array")
        if gidx.shape[0] != wts.shape[0]:
            raise ValueError("gate_indices and weights differ in length")
        if indptr.shape[0] and (indptr[0] != 0 or indptr[-1] != gidx.shape[0] or np.any(np.diff(indptr) < 0)):
            raise ValueError("indptr must start at 0, end at len(gate_indices) and be non-decreasing")
        if gidx.shape[0] and gidx.min() < 0:
            raise ValueError("negative gate indices are not supported")
        ip_dtype = np.int32 if gidx.shape[0] <= _INT32_MAX else np.int64
        max_gate = int(gidx.max()) if gidx.shape[0] else -1
        self._dev = DeviceCSR(
            torch.from_numpy(indptr.astype(ip_dtype, copy=False)).to(dev),
            torch.from_numpy(gidx.astype(np.int32, copy=False)).to(dev),
            torch.from_numpy(wts.astype(np.float32, copy=False)).to(dev),
            max_gate)
        logger.info(f"Geometry resident on {dev}: {self._dev.nbytes() / 1e6:.1f} MB")
        return self._dev


def save_geometry(geometry: GridGeometry, filepath: str) -> None:
    """Write the nine-key ``.npz`` of ``radar_grid/geometry.py:105-116`` (deflate-compressed)."""
    np.savez_compressed(
        filepath,
        grid_shape=np.array(geometry.grid_shape),
        grid_limits_z=np.array(geometry.grid_limits[0]),
        grid_limits_y=np.array(geometry.grid_limits[1]),
        grid_limits_x=np.array(geometry.grid_limits[2]),
        indptr=geometry.indptr,
        gate_indices=geometry.gate_indices,
        weights=geometry.weights,
        toa=np.array([geometry.toa]),
        radar_altitude=np.array([geometry.radar_altitude]),
    )
    file_size_mb = os.path.getsize(filepath) / 1e6
    logger.info(f"Saved geometry to {filepath} ({file_size_mb:.1f} MB on disk)")


def load_geometry(filepath: str) -> GridGeometry:
    """Read a geometry ``.npz``; files without ``toa`` / ``radar_altitude`` default to ``inf`` / ``0.0``
    (``radar_grid/geometry.py:146-147``)."""
    with np.load(filepath) as data:
        geometry = GridGeometry(
            grid_shape=tuple(data["grid_shape"]),
            grid_limits=(tuple(data["grid_limits_z"]), tuple(data["grid_limits_y"]), tuple(data["grid_limits_x"])),
            indptr=data["indptr"],
            gate_indices=data["gate_indices"],
            weights=data["weights"],
            toa=float(data["toa"][0]) if "toa" in data else np.inf,
            radar_altitude=float(data["radar_altitude"][0]) if "radar_altitude" in data else 0.0,
        )
    logger.info(f"Loaded geometry: {geometry.memory_usage_mb():.1f} MB in memory, toa={geometry.toa}m")
    return geometry


# --------------------------------------------------------------------------------------------------------------------
# Sidecar of the device layout.  The reference's expensive precompute is a file (radar_grid/geometry.py:94-150) and that
# file -- the nine-key .npz above -- stays the interchange format.  What this build derives from it for large geometries
# (the compact copy: dictionaries + packed records) can be kept NEXT to it, so that a process that loads the .npz does
# not derive it again: an uncompressed .npz keyed by the sha256 of the reference arrays it was derived from, read with
# numpy.load(allow_pickle=False).  (A geometry built on the GPU from gate coordinates needs none of this: rebuilding the
# bench geometry takes 0.3 s of kernels, less than reading any file of its size.)
# --------------------------------------------------------------------------------------------------------------------
def _reference_arrays_digest(geometry: GridGeometry) -> str:
    import hashlib
    h = hashlib.sha256()
    h.update(repr((tuple(int(v) for v in geometry.grid_shape), _native.RG_COMPACT_LINES, _native.RG_COMPACT_ROTATION)).encode())
    for arr in (geometry.indptr, geometry.gate_indices, geometry.weights):
        a = np.ascontiguousarray(arr)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(memoryview(a).cast("B"))
    return h.hexdigest()


def save_device_layout(geometry: GridGeometry, filepath: str, device=None) -> bool:
    """Write the compact copy of ``geometry`` (dictionaries, packed records or plain positions) as a sidecar ``.npz`` keyed by
    the digest of its reference arrays.  Returns ``False`` (nothing written) when the geometry has no compact copy."""
    compact = geometry.device_compact(device)
    if compact is None:
        return False
    csr = geometry.device_csr(device)
    compact.ensure_packed(csr)
    arrays = dict(key=np.frombuffer(_reference_arrays_digest(geometry).encode(), dtype=np.uint8),
                  grid_shape=np.array(compact.grid_shape, dtype=np.int64), dict_ptr=compact.dict_ptr.cpu().numpy(),
                  dict=compact.dict.cpu().numpy(), max_dict=np.array([compact.max_dict]), window_cap=np.array([compact.window_cap]),
                  chunk_pairs=compact.chunk_pairs.cpu().numpy(), chunk_counts=compact.chunk_counts.cpu().numpy())
    if compact.rec is not None:
        arrays.update(rec=compact.rec.cpu().numpy(), rec_ptr=compact.rec_ptr.cpu().numpy(),
                      rec_order=np.array([compact.rec_order]), w_base=np.array([compact.w_base], dtype=np.uint32))
    if compact.local_idx is not None:
        arrays["local_idx"] = compact.local_idx.cpu().numpy()
    np.savez(filepath, **arrays)
    logger.info(f"Saved device layout to {filepath} ({os.path.getsize(filepath) / 1e6:.1f} MB)")
    return True


def load_device_layout(geometry: GridGeometry, filepath: str, device=None) -> bool:
    """Attach the compact copy stored by :func:`save_device_layout` to ``geometry`` instead of deriving it again.  The
    sidecar is used only when its key matches the geometry's reference arrays (same CSR, same chunk layout); returns
    whether it was."""
    torch = _native.torch_mod()
    if not os.path.exists(filepath):
        return False
    with np.load(filepath, allow_pickle=False) as data:
        if bytes(data["key"]).decode() != _reference_arrays_digest(geometry):
            logger.warning(f"{filepath} was derived from another geometry: ignored")
            return False
        csr = geometry.device_csr(device)
        dev = csr.indptr.device
        up = lambda name: torch.from_numpy(np.ascontiguousarray(data[name])).to(dev)       # noqa: E731
        compact = CompactCSR(up("local_idx") if "local_idx" in data else None, up("dict_ptr"), up("dict"), int(data["max_dict"][0]),
                             int(data["window_cap"][0]), tuple(int(v) for v in data["grid_shape"]), up("chunk_pairs"),
                             up("chunk_counts"))
        if "rec" in data:
            compact.rec, compact.rec_ptr = up("rec"), up("rec_ptr")
            compact.rec_order, compact.w_base = int(data["rec_order"][0]), int(data["w_base"][0])
            compact._pack_tried = True
    geometry._compact = (csr, compact)
    logger.info(f"Device layout attached from {filepath}: {compact.nbytes() / 1e6:.1f} MB")
    return True
