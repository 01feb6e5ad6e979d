"""ctypes binding of libradargrid_hip.so (declared in include/radargrid_hip.h).

There is deliberately NO CPU fallback: if the shared library is missing, does not load, or no HIP device is
visible, every product entry point raises :class:`NativeUnavailable`.  PyTorch is used by the callers only
as plumbing -- device allocations (``tensor.data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_uint8, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library and nothing else: no environment override.  Measurement scripts that want an
# experiment build (tools/build_experiments.py) assign ``_native.LIB_PATH`` themselves before the first load; the ABI
# check below applies to them as well.
LIB_PATH = os.path.join(HERE, "csrc", "libradargrid_hip.so")
ABI_VERSION = 104          # include/radargrid_hip.h: RG_VERSION -- load_library refuses a library built from another header

RG_MAX_FIELDS = 8
RG_EXCLUDED_BITS = 0x7FD1CE5D

RG_OK, RG_EINVAL, RG_EALIGN, RG_ELAUNCH, RG_EWORKSPACE, RG_EUNSUPPORTED, RG_ENODEVICE = 0, -1, -2, -3, -4, -5, -6
GATE_OPS = {"below": 0, "above": 1, "between": 2, "outside": 3, "equal": 4, "invalid": 5}
COLUMN_OPS = {"max": 0, "min": 1, "mean": 2}
WEIGHTINGS = {"barnes2": 0, "cressman": 1, "nearest": 2, "closest": 3}
RG_MAX_PLANE_TESTS = 12
RG_TEST_LO, RG_TEST_HI, RG_TEST_LO_INCLUSIVE, RG_TEST_NONFINITE = 1, 2, 4, 8
RG_MINMAX_WORKSPACE_BYTES = 32768
RG_MAX_LUT = 4093
RG_COMPACT_LINES = 4          # grid lines (= wavefronts) per chunk of the compact CSR copy (header: RG_COMPACT_LINES)
RG_COMPACT_MAX_WINDOW = 8192
RG_COMPACT_ROTATION = 5       # block -> chunk column rotation per line group (header: RG_COMPACT_ROTATION)
RG_REC_ORDER_SEGMENT, RG_REC_ORDER_DISPATCH = 0, 1


class NativeUnavailable(RuntimeError):
    """The HIP extension (or a HIP device) is missing; the product path refuses to run without it."""


class NativeError(RuntimeError):
    """A C-ABI entry point returned a negative rg_status."""


class CellGrid(Structure):
    _fields_ = [("x0", c_double), ("y0", c_double), ("inv_cx", c_double), ("inv_cy", c_double),
                ("z_lo", c_double), ("z_hi", c_double), ("ncx", c_int32), ("ncy", c_int32), ("levels", c_int32),
                ("level0", c_int32)]


class PlaneTest(Structure):
    _fields_ = [("plane", c_void_p), ("lo", c_float), ("hi", c_float), ("flags", c_int32)]


# name -> (restype, argtypes); mirrors include/radargrid_hip.h one to one
SIGNATURES = {
    "rg_version": (c_int32, []),
    "rg_last_error": (c_char_p, []),
    "rg_device_count": (c_int32, []),
    "rg_stream_read_probe": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "rg_antenna_to_cartesian_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "rg_gate_mask_f32": (c_int32, [c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "rg_pack_fields_f32": (c_int32, [c_int32, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int64, c_int32,
                                     c_void_p, c_void_p]),
    "rg_csr_apply_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32,
                                   c_int32, c_int64, c_float, c_void_p, c_void_p]),
    "rg_csr_apply_f32_ex": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                      c_int32, c_int32, c_int64, c_float, c_void_p, c_int32, c_void_p]),
    "rg_column_reduce_f32": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p]),
    "rg_cappi_lerp_f32": (c_int32, [c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "rg_elevation_ppi_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double, c_double,
                                       c_double, c_double, c_double, c_double, c_double, c_double, c_int32, c_int32,
                                       c_void_p, c_void_p]),
    "rg_geom_bin_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32]),
    "rg_geom_bin_gates_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, POINTER(CellGrid),
                                        c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rg_geom_bin_levels_count": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, POINTER(CellGrid), c_void_p,
                                           c_int32, c_double, c_double, c_void_p, c_void_p]),
    "rg_geom_bin_levels_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "rg_geom_bin_gates_levels_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, POINTER(CellGrid),
                                               c_void_p, c_int32, c_double, c_double, c_int64, c_void_p, c_void_p, c_void_p,
                                               c_int64, c_void_p]),
    "rg_geom_count_f32": (c_int32, [c_void_p, c_void_p, POINTER(CellGrid), c_void_p, c_void_p, c_void_p, c_int32,
                                    c_int32, c_int32, c_double, c_double, c_void_p, c_void_p]),
    "rg_csr_compact_apply_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                           c_int64, c_int64, c_void_p, c_int32, c_int32, c_int64, c_float, c_void_p,
                                           c_int32, c_int32, c_void_p]),
    "rg_csr_compact_pack": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                      c_int32, c_int64, ctypes.c_uint32, c_void_p, c_void_p, c_void_p]),
    "rg_csr_compact_apply_packed_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, ctypes.c_uint32, c_void_p,
                                                  c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int32,
                                                  c_int32, c_int64, c_float, c_void_p, c_int32, c_int32, c_void_p]),
    "rg_csr_columns_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32, c_int32]),
    "rg_csr_compact_apply_columns_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, ctypes.c_uint32, c_void_p,
                                                   c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int32,
                                                   c_int32, c_int64, c_float, c_void_p, c_void_p, c_int32, c_int32,
                                                   c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                                   c_void_p, c_int64, c_int32, c_void_p]),
    "rg_csr_compact_chunks": (c_int64, [c_int64, c_int64, c_int64]),
    "rg_csr_compact_count": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                       c_void_p]),
    "rg_csr_compact_fill": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "rg_scan_workspace_bytes": (c_int64, [c_int64]),
    "rg_scan_counts_i64": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "rg_geom_fill_f32": (c_int32, [c_void_p, c_void_p, POINTER(CellGrid), c_void_p, c_void_p, c_void_p, c_int32,
                                   c_int32, c_int32, c_double, c_double, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "rg_roi_grid_f32": (c_int32, [c_void_p, c_void_p, POINTER(CellGrid), c_void_p, c_void_p, c_void_p, c_int32,
                                  c_int32, c_int32, c_double, c_double, c_int32, c_void_p, c_int32, c_int32,
                                  c_float, c_void_p, c_void_p]),
    "rg_collapse_ppi_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double,
                                      c_double, c_void_p, c_void_p, c_void_p]),
    "rg_plane_filter_f32": (c_int32, [c_void_p, c_void_p, c_int64, POINTER(PlaneTest), c_int32, c_void_p, c_void_p,
                                      c_void_p]),
    "rg_grid_filter": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_double, c_double, c_void_p, c_double, c_void_p,
                                 c_void_p]),
    "rg_nan_minmax": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_double, c_void_p, c_void_p, c_void_p]),
    "rg_colormap_rgba": (c_int32, [c_void_p, c_int32, c_int64, c_double, c_double, c_int32, c_double, c_void_p,
                                   c_int32, c_void_p, c_void_p]),
}

_lock = threading.Lock()
_lib = None


def load_library(require_device: bool = True):
    """dlopen the in-tree library and bind every declared symbol.  Raises NativeUnavailable loudly."""
    global _lib
    with _lock:
        if _lib is None:
            # PyTorch-ROCm ships its own HIP runtime; it must be the one already loaded when our library is
            # dlopen-ed, so that both share ONE runtime (streams and device pointers are only meaningful within it).
            # Loading ours first would pull in the system libamdhip64 and leave torch without a usable device.
            torch_mod()
            if not os.path.exists(LIB_PATH):
                raise NativeUnavailable(
                    f"{LIB_PATH} is missing: build it with `python -m radar_processor_amd.build` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback for the radar_grid hot path.")
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as exc:  # missing ROCm runtime etc.
                raise NativeUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
            try:
                lib.rg_version.restype = c_int32
                built_for = int(lib.rg_version())
            except AttributeError:
                raise NativeUnavailable(f"{LIB_PATH} does not export rg_version") from None
            if built_for != ABI_VERSION:
                raise NativeUnavailable(
                    f"{LIB_PATH} was built from header version {built_for}, this package binds version {ABI_VERSION}: the "
                    "signatures differ, calling it would shift arguments. Rebuild it with `python -m radar_processor_amd.build`.")
            for name, (restype, argtypes) in SIGNATURES.items():
                try:
                    fn = getattr(lib, name)
                except AttributeError:
                    raise NativeUnavailable(f"{LIB_PATH} does not export {name}") from None
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    if require_device and _lib.rg_device_count() <= 0:
        raise NativeUnavailable("libradargrid_hip.so loaded but no HIP device is visible "
                                f"({_lib.rg_last_error().decode()}); the radar_grid hot path has no CPU fallback.")
    return _lib


def check(status: int, what: str) -> None:
    if status < 0:
        msg = load_library(require_device=False).rg_last_error().decode()
        raise NativeError(f"{what} failed with rg_status {status}: {msg}")


def torch_mod():
    try:
        import torch
    except Exception as exc:  # pragma: no cover
        raise NativeUnavailable(f"PyTorch-ROCm is required for device memory and streams: {exc}") from exc
    return torch


def device(index=None):
    """Return the torch device the HIP path runs on; raises when there is none."""
    torch = torch_mod()
    load_library(require_device=True)
    if not torch.cuda.is_available():
        raise NativeUnavailable("torch.cuda.is_available() is False: no MI355X visible to PyTorch-ROCm")
    if index is None:
        index = torch.cuda.current_device()
    return torch.device("cuda", index)


def canonical_device(dev):
    """``torch.device("cuda")`` / ``"cuda:1"`` / an int -> the indexed ``torch.device``: tensors report ``cuda:N``, and an
    index-less device compares unequal to it (a cache keyed on the device would otherwise never hit)."""
    torch = torch_mod()
    if dev is None:
        return device()
    dev = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (0 = the null stream)."""
    torch = torch_mod()
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else int(t.data_ptr())
