#!/usr/bin/env python3
"""Eager vs hipGraph-replayed per-volume pipeline on a small grid (launch-bound regime)."""
import os, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import radar_processor_amd as rg
from radar_processor_amd import synthetic
from radar_processor_amd.pipeline import VolumePipeline

for label, kw, shape, limits in (
        ("C1 1x500x500", dict(n_elev=1, n_az=360, n_gates=500), (1, 500, 500), ((0.0, 0.0), (-240e3, 240e3), (-240e3, 240e3))),
        ("notebook-like 9x315x315", dict(n_elev=12, n_az=360, n_gates=652), (9, 315, 315), ((0.0, 8000.0), (-157e3, 157e3), (-157e3, 157e3)))):
    vol = synthetic.make_volume(seed=0, fields=("DBZH",), **kw)
    if shape[0] == 1:
        vol.gate_z = np.zeros_like(vol.gate_z)
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, tmp)
    f = [torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).cuda()]
    m = [torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).cuda()]
    for use_graph in (False, True):
        p = VolumePipeline(geom, vol.n_total_gates, 1, cappi_altitude=limits[0][1] * 0.37, use_graph=use_graph)
        for _ in range(5):
            p.run(f, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            p.run(f, m)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{label}: pairs={geom.n_pairs():,} graph={use_graph}: {dt * 1e6:.1f} us/volume "
              f"({np.prod(shape) / dt / 1e6:.1f} Mvoxel/s)", flush=True)
