#!/usr/bin/env python3
"""A/B timing of rg_csr_apply_f32 tuning variants in ONE process, interleaved rounds (cdna_hip_programming.md
§5.4 rule 24).  Diagnostic tool, not part of the product path.

    python tools/tune_k1.py [--config C2] [--rounds 7] [--variants 0 9 18 20 22 28]   (see dispatch() in rg_csr_apply.hip)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", type=int, nargs="*", default=[0, 9, 18, 20, 22, 28])
    args = ap.parse_args()
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder

    lib = rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[args.config]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
        torch.cuda.synchronize()
        print(f"geometry: {geom.n_pairs():,} pairs in {time.perf_counter() - t0:.2f}s", flush=True)
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    g = CsrGridder(geom, vol.n_total_gates, 1, device=dev)
    g.pack([f], [m])
    csr = g.csr
    outs = {v: torch.empty(g.n_vox, dtype=torch.float32, device=dev) for v in args.variants}
    algo = g.algorithmic_bytes()

    def run(v):
        _native.check(lib.rg_csr_apply_f32_ex(_native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(csr.gate_indices),
                                              _native.ptr(csr.weights), g.n_vox, csr.n_pairs, _native.ptr(g.packed), 1, 1,
                                              g.n_gates, float("nan"), _native.ptr(outs[v]), v, _native.stream_ptr()),
                      "rg_csr_apply_f32_ex")

    for v in args.variants:
        run(v)
    torch.cuda.synchronize()
    ref = outs[args.variants[0]]
    for v in args.variants[1:]:
        same = torch.allclose(ref, outs[v], rtol=1e-6, atol=1e-5, equal_nan=True)
        print(f"variant {v} vs {args.variants[0]}: allclose={same}", flush=True)
    times = {v: [] for v in args.variants}
    for _ in range(args.rounds):
        for v in args.variants:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run(v)
            b.record()
            b.synchronize()
            times[v].append(a.elapsed_time(b))
    for v in args.variants:
        t = np.array(times[v])
        print(f"variant {v}: median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  -> {algo / np.median(t) / 1e6:8.1f} GB/s "
              f"({algo / np.median(t) / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
