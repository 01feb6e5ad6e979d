"""Fused on-the-fly gridding (no CSR) -- ``rg_roi_grid_f32`` (csrc/rg_roi_grid.hip).

``compute_grid_geometry`` + ``apply_geometry`` in one kernel: for every voxel the neighbour search, the weights
(``radar_grid/compute.py:46-91``) and the masked weighted mean (``radar_grid/interpolate.py:69-104``) are
evaluated together and nothing but the output grid is written.  This is the path for grids whose CSR is too
large to be worth keeping; for repeated volumes on a moderate grid the CSR path (K1) is faster
because it amortises the search.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _native
from .geometry_builder import WEIGHTINGS, RoiSearch
from .gridding import _stride_for


def roi_grid_fields_device(search: RoiSearch, fields: Sequence, masks: Optional[Sequence] = None, shared_mask=None,
                           weighting: str = "barnes2", fill_value: float = np.nan, out=None):
    """Grid device-resident fields straight from the gates.  Arguments as ``grid_fields_device`` with a
    :class:`RoiSearch` (cell-sorted gates) in place of the CSR geometry.  Returns ``[F, nz, ny, nx]`` float32."""
    torch = _native.torch_mod()
    lib = _native.load_library()
    if weighting not in WEIGHTINGS and weighting != "closest":   # 'closest' = single nearest gate (fused path only)
        raise ValueError(f"Unknown weighting function: {weighting}")
    n_fields = len(fields)
    if n_fields == 0:
        raise ValueError("no fields to grid")
    dev = search.dev
    n_gates = search.n_gates
    for i, f in enumerate(fields):
        if not (f.is_cuda and f.device == dev and f.dtype == torch.float32 and f.is_contiguous()
                and f.numel() == n_gates):
            raise ValueError(f"field {i}: expected a contiguous float32 tensor of {n_gates} gates on {dev}")
    if masks is None:
        masks = [None] * n_fields
    if len(masks) != n_fields:
        raise ValueError("masks must have one entry (tensor or None) per field")
    for i, m in enumerate(list(masks) + [shared_mask]):
        if m is not None and not (m.is_cuda and m.device == dev and m.dtype == torch.uint8 and m.is_contiguous()
                                  and m.numel() == n_gates):
            raise ValueError(f"mask {i}: expected a contiguous uint8 tensor of {n_gates} gates on {dev}")
    nz, ny, nx = search.grid_shape
    n_vox = nz * ny * nx
    if out is None:
        out = torch.empty((n_fields, nz, ny, nx), dtype=torch.float32, device=dev)
    elif not (out.is_cuda and out.device == dev and out.dtype == torch.float32 and out.is_contiguous()
              and out.numel() == n_fields * n_vox):
        # the kernel writes n_fields * n_vox floats through a raw pointer: anything else is an out-of-bounds write
        raise ValueError(f"out must be a contiguous float32 tensor of shape [F, nz, ny, nx] on {dev}")
    fill = float(np.float32(fill_value))
    with torch.cuda.device(dev):
        stream = _native.stream_ptr()
        for f0 in range(0, n_fields, _native.RG_MAX_FIELDS):
            group = list(range(f0, min(n_fields, f0 + _native.RG_MAX_FIELDS)))
            nf = len(group)
            stride = _stride_for(nf)
            packed = torch.empty(max(n_gates, 1) * stride, dtype=torch.float32, device=dev)
            fptrs = (ctypes.c_void_p * nf)(*[_native.ptr(fields[i]) for i in group])
            mptrs = (ctypes.c_void_p * nf)(*[_native.ptr(masks[i]) for i in group])
            _native.check(lib.rg_pack_fields_f32(nf, fptrs, mptrs, _native.ptr(shared_mask), n_gates, stride,
                                                 _native.ptr(packed), stream), "rg_pack_fields_f32")
            out_view = out.view(n_fields, n_vox)[f0:f0 + nf]
            _native.check(lib.rg_roi_grid_f32(
                _native.ptr(search.sorted_gates), _native.ptr(search.cell_start), search.cells, _native.ptr(search.xc),
                _native.ptr(search.yc), _native.ptr(search.zc), nz, ny, nx, search.min_radius, search.beam_factor,
                _native.WEIGHTINGS[weighting], _native.ptr(packed), nf, stride, fill, _native.ptr(out_view), stream),
                "rg_roi_grid_f32")
    return out.view(n_fields, nz, ny, nx)
