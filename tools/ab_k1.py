#!/usr/bin/env python3
"""A/B of two builds of libradargrid_hip.so on rg_csr_apply_f32 (interleaved rounds, one process):
    python tools/ab_k1.py tools/libradargrid_old.so [--config METRIC] [--fields 1]
The first library is "old", the in-tree one is "new".  Development tool."""
import argparse
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old_lib")
    ap.add_argument("--config", default="METRIC")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--fields", type=int, nargs="*", default=[1])
    args = ap.parse_args()
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder
    new = _native.load_library()
    old = ctypes.CDLL(os.path.abspath(args.old_lib))
    for lib in (old,):
        fn = lib.rg_csr_apply_f32
        fn.restype, fn.argtypes = _native.SIGNATURES["rg_csr_apply_f32"]
    cfg = synthetic.CONFIGS[args.config]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    csr = search.build_csr("barnes2")
    geom = rg.GridGeometry.from_device(cfg["grid_shape"], cfg["grid_limits"], csr, 17000.0)
    dev = search.dev
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    for nf in args.fields:
        g = CsrGridder(geom, f.numel(), nf, device=dev)
        g.pack([f * (1 + 0.1 * i) for i in range(nf)], [m] * nf)
        outs = {k: torch.empty((nf, g.n_vox), dtype=torch.float32, device=dev) for k in ("old", "new")}

        def run(lib, out):
            _native.check(lib.rg_csr_apply_f32(_native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(csr.gate_indices),
                                               _native.ptr(csr.weights), g.n_vox, csr.n_pairs, _native.ptr(g.packed), nf,
                                               g.stride, g.n_gates, float("nan"), _native.ptr(out),
                                               _native.stream_ptr()), "apply")

        libs = {"old": old, "new": new}
        for k in libs:
            run(libs[k], outs[k])
        torch.cuda.synchronize()
        same = torch.equal(outs["old"].view(torch.int32), outs["new"].view(torch.int32))
        if not same:
            diff = (outs["old"].view(torch.int32) != outs["new"].view(torch.int32))
            idx = diff.nonzero()
            print(f"  {idx.shape[0]} mismatching outputs; per field: {diff.sum(dim=1).tolist()}")
            ip = csr.indptr.to(torch.int64)
            for fi, vi in idx[:6].tolist():
                print(f"  field {fi} voxel {vi} (chunk {vi // 64}, lane {vi % 64}) rowlen {int(ip[vi + 1] - ip[vi])} "
                      f"chunk span {int(ip[min(vi // 64 * 64 + 64, g.n_vox)] - ip[vi // 64 * 64])} "
                      f"old {float(outs['old'][fi, vi])} new {float(outs['new'][fi, vi])}")
        times = {k: [] for k in libs}
        for _ in range(args.rounds):
            for k in libs:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(libs[k], outs[k])
                b.record()
                b.synchronize()
                times[k].append(a.elapsed_time(b))
        print(f"{args.config} fields={nf} bit_identical={same} " +
              " ".join(f"{k}: median {np.median(v):.3f} min {min(v):.3f} ms" for k, v in times.items()), flush=True)
        del g, outs


if __name__ == "__main__":
    main()
