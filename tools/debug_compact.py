#!/usr/bin/env python3
"""Debug aid: K1 vs K1c vs float64 oracle on a small random-cloud geometry, printing where they differ."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import radar_processor_amd as rg
from radar_processor_amd.gridding import CsrGridder
from oracle import radar_grid_oracle as oracle
from test_gpu_edges import _cloud

shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,3,255").split(","))
gx, gy, gz, val, mask = _cloud(7, 6000)
nz, ny, nx = shape
limits = ((500.0, 500.0 if nz == 1 else 6000.0), (-3e3, -3e3 if ny == 1 else 9e3), (-15e3, -15e3 if nx == 1 else 15e3))
with tempfile.TemporaryDirectory() as tmp:
    geom = rg.compute_grid_geometry(gx, gy, gz, shape, limits, tmp, min_radius=1500.0, beam_factor=0.05)
dev = torch.device("cuda")
f = torch.from_numpy(val).to(dev); m = torch.from_numpy(mask.astype(np.uint8)).to(dev)
extra = [torch.from_numpy(np.roll(val, 7 * (i + 1)) * np.float32(1.0 + i)).to(dev) for i in range(7)]
emask = [torch.from_numpy(np.roll(mask, 13 * (i + 1)).astype(np.uint8)).to(dev) if i % 2 else None for i in range(7)]
c = geom.device_compact(dev)
csr = geom.device_csr(dev)
print("pairs", csr.n_pairs, "window_cap", c.window_cap, "max_dict", c.max_dict, "dict_ptr", c.dict_ptr.tolist())
ip = geom.indptr; print("row lengths max", np.diff(ip).max(), "rows", len(ip) - 1)
for nf in (1, 2, 3, 4, 8):
    fl, ml = [f] + extra[:nf - 1], [m] + emask[:nf - 1]
    for window in (c.window_for(nf), 0):
        gc = CsrGridder(geom, f.numel(), nf, device=dev, compact=True); gc.compact, gc.window = c, window
        gs = CsrGridder(geom, f.numel(), nf, device=dev)
        gc.pack(fl, ml); gs.pack(fl, ml)
        want = torch.empty((nf, gs.n_vox), device=dev); got = torch.full_like(want, 3.0)
        gs.apply(want, -1.0); gc.apply(got, -1.0); torch.cuda.synchronize()
        bad = (got.view(torch.int32) != want.view(torch.int32)).nonzero()
        print(f"nf={nf} window={window}: {bad.shape[0]} mismatches", bad[:12].tolist())
        for i in range(nf):
            fm = ml[i].cpu().numpy().astype(bool) if ml[i] is not None else np.zeros(len(val), bool)
            o = oracle.csr_apply_f64(geom.indptr, geom.gate_indices, geom.weights, fl[i].cpu().numpy(), fm, shape, fill_value=-1.0).ravel()
            for name, t in (("k1", want), ("k1c", got)):
                d = np.abs(t[i].cpu().numpy() - o)
                if d.max() > 1e-3:
                    print(f"   field {i} {name} vs oracle: max abs err {d.max():.4g} at {int(d.argmax())}, n_bad {(d > 1e-3).sum()}")
