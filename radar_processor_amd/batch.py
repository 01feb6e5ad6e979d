"""Multi-GPU batches of independent volumes.

Volumes (and fields) are independent and share one geometry per radar / scan strategy, so the batch shards
embarrassingly: volume ``b`` goes to rank ``b mod world_size``, one process per GPU, the geometry is replicated
(or rebuilt -- 0.4 s on the GPU) on every rank, and the gridding path needs **no collective**.  The reference has
nothing comparable (its only parallelism is a ``multiprocessing.Pool`` over z-levels inside the geometry build,
``radar_grid/compute.py:218-222``).

The one exchange step that makes physical sense is an optional multi-radar composite: the element-wise
NaN-ignoring maximum of the ranks' 2-D product planes, an all-reduce(MAX) -- RCCL over xGMI when the planes are
cuda tensors (backend ``"nccl"`` is RCCL on ROCm), gloo for CPU tensors (tests).  A 2000x2000 float32 plane is
16 MB: about 28 MB per link direction in a ring, ~0.2 ms at ~150 GB/s per xGMI link, negligible next to a
multi-millisecond gridding kernel.
"""
from __future__ import annotations

import os
import contextlib
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def init_distributed(backend: Optional[str] = None) -> bool:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (as set by
    ``torch.distributed.run``).  Returns ``True`` when a group with more than one rank is active."""
    dist = _dist()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    if not dist.is_initialized():
        import torch
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kwargs = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, **kwargs)
    return True


def rank_and_world() -> tuple:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> List[int]:
    """Items owned by ``rank``: ``b`` with ``b mod world_size == rank`` (round-robin keeps the shards within one
    item of each other for any batch size)."""
    if rank is None or world_size is None:
        rank, world_size = rank_and_world()
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_items, world_size))


def composite_max(plane, group=None):
    """NaN-ignoring element-wise maximum of ``plane`` over all ranks (``np.fmax.reduce`` over the per-radar
    planes); a pixel that is NaN on every rank stays NaN.  ``plane``: float32 torch tensor (cuda -> RCCL,
    cpu -> gloo).  Returns a new tensor on every rank; with a single rank it is a copy."""
    import torch
    dist = _dist()
    neg_inf = torch.full_like(plane, float("-inf"))
    work = torch.where(torch.isnan(plane), neg_inf, plane).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(work, op=dist.ReduceOp.MAX, group=group)
    return torch.where(work == float("-inf"), torch.full_like(work, float("nan")), work)


def run_sharded(n_items: int, work_fn: Callable[[int], object], rank: Optional[int] = None,
                world_size: Optional[int] = None) -> Dict[int, object]:
    """Apply ``work_fn(b)`` to every item ``b`` of this rank's shard; returns ``{b: result}``.  No
    communication happens here -- the data path of a sharded batch has no collective."""
    return {b: work_fn(b) for b in shard_indices(n_items, rank, world_size)}


def gather_results(local: Dict[int, object], n_items: int, dst: int = 0, group=None) -> Optional[List[object]]:
    """Collect small per-item results (product planes, checksums) on rank ``dst`` in item order.  Control-plane
    convenience built on ``gather_object``; bulk grids should stay on their GPU."""
    dist = _dist()
    rank, world = rank_and_world()
    if world == 1:
        return [local[b] for b in range(n_items)]
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local, bucket, dst=dst, group=group)
    if rank != dst:
        return None
    merged: Dict[int, object] = {}
    for part in bucket:
        merged.update(part)
    missing = [b for b in range(n_items) if b not in merged]
    if missing:
        raise RuntimeError(f"items {missing} were not produced by any rank")
    return [merged[b] for b in range(n_items)]


class VolumeBatch:
    """Grids this rank's shard of a batch of volumes through one shared geometry, fusing up to 4 (CSR path) or 8
    (CSR-free path) field-volumes into each pass (the CSR -- or, for the CSR-free gridder, the candidate search -- is the
    dominant cost, so it is paid once for the whole group).  Inputs that live in host memory are uploaded one group
    ahead on a copy stream, so that the upload of group g+1 overlaps the gridding of group g (page-locked tensors
    copy asynchronously; NumPy arrays and pageable tensors still work, synchronously).

    ``geometry``: a :class:`GridGeometry` (CSR path, ``rg_csr_apply_f32``) or a :class:`RoiSearch` (fused path,
    ``rg_roi_grid_f32``; measured faster than the CSR path as soon as 3 or more field-volumes share a pass).
    ``volumes``: sequence of ``{field name: (values, mask)}`` per volume, values float32 ``[G]`` and mask
    uint8/bool ``[G]`` or ``None`` -- NumPy arrays or tensors already on the device.
    """

    def __init__(self, geometry, field_names: Sequence[str], device=None, weighting: str = "barnes2"):
        from . import _native
        from .geometry_builder import RoiSearch
        self.geometry = geometry
        self.fused = isinstance(geometry, RoiSearch)
        self.weighting = weighting
        self.field_names = list(field_names)
        self.dev = geometry.dev if self.fused else _native.canonical_device(device)
        if not 1 <= len(self.field_names) <= _native.RG_MAX_FIELDS:
            raise ValueError("1..8 fields per volume")
        # measured on the bench grid (ms per fused pass): row-wise kernel over the packed records 7.9 / 8.7 / 10.1 / 10.5 / 15.5
        # for 1 / 2 / 3 / 4 / 8 field-volumes (1.9 ms each at eight, where the geometry's LDS window admits eight: decided per
        # geometry by gridding.fields_per_pass, on first use), rg_csr_apply_f32 13.1 / 14.9 / 18.8 / 20.7 and 55.1 for eight;
        # the CSR-free gridder keeps gaining up to 8 because it shares the whole neighbour search
        self._cap = _native.RG_MAX_FIELDS if self.fused else None

    @property
    def volumes_per_pass(self) -> int:
        """Volumes one pass fuses (decided on first use: the CSR path asks the geometry's device copy)."""
        if self._cap is None:
            from .gridding import volumes_per_pass_cap
            self._cap = volumes_per_pass_cap(self.geometry, self.dev)
        return max(1, self._cap // len(self.field_names))

    def _to_dev(self, a, dtype):
        import torch
        if a is None:
            return None
        if not type(a).__module__.startswith("torch"):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=self.dev, dtype=dtype).contiguous()

    def _stage(self, volumes, group, slot_set, copy_stream):
        """Device tensors of one group's fields and masks.  Inputs already on the device pass through; anything that has
        to cross PCIe is copied on ``copy_stream`` into one of two persistent sets of staging buffers (no allocation per
        step, nothing for the caching allocator to keep alive across streams).  Returns ``(fields, masks, event or
        None)`` -- the event marks the end of the copies."""
        import torch

        def on_device(a):
            return type(a).__module__.startswith("torch") and a.is_cuda

        host = [a for b in group for name in self.field_names for a in volumes[b][name] if a is not None and not on_device(a)]
        if not host:
            fields = [self._to_dev(volumes[b][name][0], torch.float32) for b in group for name in self.field_names]
            masks = [self._to_dev(volumes[b][name][1], torch.uint8) for b in group for name in self.field_names]
            return fields, masks, None
        n_gates = int(np.prod(host[0].shape))
        slots = self.volumes_per_pass * len(self.field_names)
        if getattr(self, "_staging", None) is None or self._staging[0][0].shape != (slots, n_gates):
            self._staging = [(torch.empty((slots, n_gates), dtype=torch.float32, device=self.dev),
                              torch.empty((slots, n_gates), dtype=torch.uint8, device=self.dev)) for _ in range(2)]
            self._staging_free = [None, None]
        vals, msks = self._staging[slot_set]
        fields, masks = [], []
        with torch.cuda.stream(copy_stream):
            if self._staging_free[slot_set] is not None:      # the pass that last read this set must have finished
                copy_stream.wait_event(self._staging_free[slot_set])
            k = 0
            for b in group:
                for name in self.field_names:
                    for a, buf, out in zip(volumes[b][name], (vals[k], msks[k]), (fields, masks)):
                        if a is None or on_device(a):
                            out.append(a if a is None else self._to_dev(a, buf.dtype))
                            continue
                        src = a if type(a).__module__.startswith("torch") else torch.from_numpy(np.ascontiguousarray(a))
                        buf.copy_(src.reshape(-1), non_blocking=True)
                        out.append(buf)
                    k += 1
            event = copy_stream.record_event()
        return fields, masks, event

    def grid_shard(self, volumes: Sequence[dict], products: Optional[Callable] = None, rank=None,
                   world_size=None, events: Optional[list] = None) -> Dict[int, object]:
        """Returns ``{volume index: grids [F, nz, ny, nx]}`` -- or ``{index: products(grids)}`` when a reducer is
        given, so that only 2-D planes outlive the pass.  ``products`` may also be a :class:`gridding.PlaneProducts`
        (column maximum / argmax / CAPPIs): the result is then ``{index: [one dict of planes per field]}`` and, on the CSR
        path of a large geometry and on request, the pass runs the gridding kernel's column mode with its products epilogue -- the 3-D grids
        are neither written nor read back (``gridding.grid_products_device``).  ``volumes`` is indexed by the GLOBAL volume number; only this
        rank's entries (``shard_indices``) are touched, the others may be ``None``.  ``events``: optional list that
        receives one ``(start, end)`` pair of stream events per gridding pass (mask fold + gridding kernel), for
        callers that time the kernel itself (``bench.py``)."""
        import torch
        from .gridding import PlaneProducts, grid_fields_device, grid_products_device
        from .roi_grid import roi_grid_fields_device
        plane_spec = products if isinstance(products, PlaneProducts) else None
        if plane_spec is not None and self.fused:      # the CSR-free gridder has no epilogue: reduce its grids as usual
            from . import grid_products as gp
            geom_like = self.geometry

            def products(g, _spec=plane_spec):         # noqa: F811 -- the reducer form of the same request
                recs = []
                for k in range(g.shape[0]):
                    rec = {}
                    lo, hi = gp._level_window(int(g.shape[1]), *_spec.window, geom_like)
                    if _spec.colmax:
                        got = gp._column("max", g[k], lo, hi, None, None, None, want_arg=_spec.argmax)
                        if _spec.argmax:
                            rec["colmax"], rec["argmax"] = got
                        else:
                            rec["colmax"] = got
                    if _spec.cappi:
                        rec["cappi"] = {alt: gp.constant_altitude_ppi(g[k], geom_like, alt, _spec.interpolation)
                                        for alt in _spec.cappi}
                    recs.append(rec)
                return recs
            plane_spec = None
        mine = shard_indices(len(volumes), rank, world_size)
        out: Dict[int, object] = {}
        n_f = len(self.field_names)
        groups = [mine[g0:g0 + self.volumes_per_pass] for g0 in range(0, len(mine), self.volumes_per_pass)]
        if not groups:
            return out
        with torch.cuda.device(self.dev):
            compute = torch.cuda.current_stream()
            copy_stream = self._copy_stream = getattr(self, "_copy_stream", None) or torch.cuda.Stream()
            staged = self._stage(volumes, groups[0], 0, copy_stream)
        for gi, group in enumerate(groups):
            fields, masks, ready = staged
            if ready is not None:
                compute.wait_event(ready)
            if gi + 1 < len(groups):                    # the next group's upload runs while this group is gridded
                with torch.cuda.device(self.dev):
                    staged = self._stage(volumes, groups[gi + 1], (gi + 1) & 1, copy_stream)
            if events is not None:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record()
            planes = None
            if self.fused:
                grids = roi_grid_fields_device(self.geometry, fields, masks, weighting=self.weighting)
            elif plane_spec is not None:                # products only: one fused launch per group, no 3-D grid in HBM
                planes = grid_products_device(self.geometry, fields, masks, products=plane_spec)
            else:
                grids = grid_fields_device(self.geometry, fields, masks)
            if events is not None:
                pair[1].record()
                events.append(pair)
            if ready is not None:                       # this group's staging set may be overwritten once the pass is done
                self._staging_free[gi & 1] = compute.record_event()
            for i, b in enumerate(group):
                if planes is not None:
                    out[b] = planes[i * n_f:(i + 1) * n_f]
                    continue
                g = grids[i * n_f:(i + 1) * n_f]
                out[b] = products(g) if products is not None else g
        return out
