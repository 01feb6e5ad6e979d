"""Per-volume pipeline with persistent buffers, replayable as one hipGraph.

The per-volume hot path of SURVEY.md §3.1 -- mask fold (``rg_pack_fields_f32``) -> gridding
(``rg_csr_apply_f32``) -> COLMAX(+argmax) (``rg_column_reduce_f32``) -> CAPPI (``rg_cappi_lerp_f32``) -- is 3 + 2F
kernel launches.  On the big bench grid the launches are noise next to a 13 ms gridding kernel; on small grids
(the reference's own 315x315x9 example grids, single-sweep 500x500 PPIs) they are the whole cost.  Every C-ABI entry
point only enqueues work on the caller's stream and never allocates or synchronises, so the sequence can be captured
once into a hipGraph (``torch.cuda.CUDAGraph``) and replayed per volume with a single launch.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from . import _native
from .grid_geometry import GridGeometry
from .gridding import CsrGridder


class VolumePipeline:
    """Grids ``n_fields`` fields of one volume and collapses each to COLMAX / argmax / CAPPI.

    All inputs are copied into fixed device buffers (``.fields_in`` / ``.masks_in``), all outputs live in fixed
    buffers (``.grid`` ``[F, nz, ny, nx]``, ``.colmax`` / ``.cappi`` ``[F, ny, nx]`` float32, ``.argmax`` int32), so a
    captured graph stays valid from volume to volume.  ``run()`` returns views of those buffers: consume or clone
    them before the next call.
    """

    def __init__(self, geometry: GridGeometry, n_gates: int, n_fields: int = 1, cappi_altitude: float = 4000.0,
                 fill_value: float = np.nan, use_graph: bool = True, device=None, compact: bool = False):
        torch = _native.torch_mod()
        self.lib = _native.load_library()
        self.dev = _native.canonical_device(device)
        self.geometry = geometry
        # compact=True: pipelines over a large geometry grid through its compact CSR copy (the same values to float32
        # rounding: passes of 1-8 fields run the row-wise kernel over the packed records)
        self.gridder = CsrGridder(geometry, n_gates, n_fields, device=self.dev, compact=compact)
        self.n_fields, self.n_gates = int(n_fields), int(n_gates)
        self.fill_value = fill_value
        nz, ny, nx = self.gridder.grid_shape
        self.shape = (nz, ny, nx)
        f32, dev = torch.float32, self.dev
        self.fields_in = [torch.zeros(n_gates, dtype=f32, device=dev) for _ in range(n_fields)]
        self.masks_in = [torch.zeros(n_gates, dtype=torch.uint8, device=dev) for _ in range(n_fields)]
        self.grid = torch.empty((n_fields, nz, ny, nx), dtype=f32, device=dev)
        self.colmax = torch.empty((n_fields, ny, nx), dtype=f32, device=dev)
        self.argmax = torch.empty((n_fields, ny, nx), dtype=torch.int32, device=dev)
        self.cappi = torch.empty((n_fields, ny, nx), dtype=f32, device=dev)
        self._cappi_plan = self._plan_cappi(cappi_altitude)
        self._graph = None
        self._want_graph = bool(use_graph)

    def _plan_cappi(self, altitude: float):
        """Scalar control flow of constant_altitude_ppi (radar_grid/products.py:361-404), resolved once."""
        from .grid_products import cappi_plan
        plan = cappi_plan(self.geometry.grid_limits[0], self.shape[0], altitude)
        if plan[0] == "blend":
            return ("lerp", plan[1], float(np.float32(plan[2])), float(np.float32(plan[3])))
        return ("nan",) if plan[0] == "outside" else plan

    def _enqueue(self) -> None:
        lib, ptr, stream = self.lib, _native.ptr, _native.stream_ptr()
        nz, ny, nx = self.shape
        n_xy = ny * nx
        self.gridder.pack(self.fields_in, self.masks_in)
        self.gridder.apply(self.grid.view(self.n_fields, -1), self.fill_value)
        for f in range(self.n_fields):
            _native.check(lib.rg_column_reduce_f32(ptr(self.grid[f]), nz, n_xy, 0, nz - 1, _native.COLUMN_OPS["max"],
                                                   ptr(self.colmax[f]), ptr(self.argmax[f]), stream), "rg_column_reduce_f32")
            plan = self._cappi_plan
            if plan[0] == "lerp":
                _native.check(lib.rg_cappi_lerp_f32(ptr(self.grid[f]), n_xy, plan[1], plan[2], plan[3], ptr(self.cappi[f]),
                                                    stream), "rg_cappi_lerp_f32")
            elif plan[0] == "level":
                self.cappi[f].copy_(self.grid[f, plan[1]])
            else:
                self.cappi[f].fill_(float("nan"))

    def run(self, fields: Sequence, masks: Optional[Sequence] = None) -> Dict[str, object]:
        """``fields``: ``n_fields`` float32 arrays/tensors ``[G]``; ``masks``: matching uint8/bool or ``None``."""
        torch = _native.torch_mod()
        if len(fields) != self.n_fields:
            raise ValueError(f"expected {self.n_fields} fields")
        masks = [None] * self.n_fields if masks is None else list(masks)
        with torch.cuda.device(self.dev):
            for i in range(self.n_fields):
                src = fields[i]
                if not type(src).__module__.startswith("torch"):
                    src = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))
                self.fields_in[i].copy_(src.view(-1), non_blocking=True)
                m = masks[i]
                if m is None:
                    self.masks_in[i].zero_()
                else:
                    if not type(m).__module__.startswith("torch"):
                        m = torch.from_numpy(np.ascontiguousarray(m).astype(np.uint8))
                    self.masks_in[i].copy_(m.view(-1), non_blocking=True)
            if self._want_graph and self._graph is None:
                self._enqueue()                       # warm-up outside capture (module load, allocator)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._enqueue()
                self._graph = graph
            if self._graph is not None:
                self._graph.replay()
            else:
                self._enqueue()
        return {"grid": self.grid, "colmax": self.colmax, "argmax": self.argmax, "cappi": self.cappi}
