#!/usr/bin/env python3
"""Experiment: the single-field kernels over the compact copy on one grid (default METRIC), interleaved rounds: the
row-wise kernel (tile 0), the tile kernel over the packed records with 384 / 576 / 768-pair tiles, and -- over the plain
arrays -- the timing-only ablations of the tile kernel (tile codes 902-909: no row phase / no products / no window)."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_experiments            # the timing-only tile codes exist only in the -DRG_EXPERIMENTS build
build_experiments.use()
import torch
import radar_processor_amd as rg
from radar_processor_amd import synthetic
from radar_processor_amd.gridding import CsrGridder
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "METRIC"]
vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as tmp:
    geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
g.pack([f], [m])
print('packed stream:', g.packed_stream, 'bytes/launch', g.compact_bytes())
print('window', g.window, 'dict B/pair', 4 * g.compact.n_dict / g.csr.n_pairs, 'max_dict', g.compact.max_dict)
out = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
variants = [(0, 0), (384, 0), (576, 0), (768, 0), (902, 0), (903, 0), (909, 0)]
times = {v: [] for v in variants}
for rnd in range(6):
    for t, r in variants:
        g.tile = t
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.apply(out); e1.record(); e1.synchronize()
        if rnd:
            times[(t, r)].append(e0.elapsed_time(e1))
for v in variants:
    print(v, round(float(np.median(times[v])), 3), round(float(np.min(times[v])), 3))
