#!/usr/bin/env python3
"""Column-persistent kernel (rg_csr_compact_apply_columns_f32) against the one-chunk-per-workgroup row-wise kernel, one
configuration, same process, same arrays, randomised interleaved rounds:

    python tools/exp_columns.py [--config C2|METRIC|C4] [--fields 1,3,4] [--pieces 0,1,2,4] [--rounds 9]

Variants per field count: ``row`` (rg_csr_compact_apply_packed_f32, tile = 0), ``col/pN`` (columns kernel, 3-D grid
stored, N level pieces; 0 = the package's choice), ``col/pN/plain`` (identity workgroup order instead of heaviest
first), ``prod/pN`` (products only: COLMAX + argmax + the two levels of a 4000 m CAPPI, no 3-D store), and ``row+k3`` = the
row-wise kernel followed by the separate COLMAX/argmax and CAPPI kernels (what ``prod`` replaces).  One JSON object on stdout:
median ms, TB/s in the kernel's own algorithmic bytes, and whether the grids / planes are the same bits."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--fields", default="1,3")
    ap.add_argument("--pieces", default="0,1,2,4")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--layout", default="auto")
    ap.add_argument("--exp-lib", default="", help="tag[:-DFLAG,...]: run through an experiment build (tools/build_experiments.py)")
    args = ap.parse_args()
    if args.exp_lib:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import build_experiments
        tag, _, flags = args.exp_lib.partition(":")
        build_experiments.use(tag, [f for f in flags.split(",") if f])
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import grid_products as gp, synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[args.config]
    names = ("DBZH", "ZDR", "RHOHV")
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=names)
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp,
                                        layout=args.layout)
    base_f = [torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields[n]))).to(dev) for n in names]
    base_m = [torch.from_numpy(np.ma.getmaskarray(vol.fields[n]).astype(np.uint8)).to(dev) for n in names]
    qc = rg.device_gate_mask(base_f[2], "below", 0.8)
    nz, ny, nx = cfg["grid_shape"]
    n_vox = nz * ny * nx
    compact = geom.device_compact(dev)
    rec = {"config": args.config, "pairs": geom.n_pairs(), "window_cap": compact.window_cap, "runs": []}
    plan = gp.cappi_plan(cfg["grid_limits"][0], nz, 4000.0)
    keep_lo, n_keep = (plan[1], 2) if plan[0] == "blend" else (plan[1], 1)

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1)

    for nf in [int(x) for x in args.fields.split(",")]:
        fl = [base_f[i % 3] for i in range(nf)]
        ml = [base_m[i % 3] for i in range(nf)]
        g = CsrGridder(geom, fl[0].numel(), nf, device=dev, compact=True)
        assert g.has_columns_kernel, "needs the packed records"
        g.pack(fl, ml, qc if nf >= 3 else None)
        out = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
        ref = torch.empty_like(out)
        g.apply(ref)
        cmax = torch.empty((nf, ny, nx), dtype=torch.float32, device=dev)
        carg = torch.empty((nf, ny, nx), dtype=torch.int32, device=dev)
        planes = torch.empty((nf, n_keep, ny, nx), dtype=torch.float32, device=dev)
        cap = torch.empty((ny, nx), dtype=torch.float32, device=dev)

        def row_k3():
            g.apply(out)
            for k in range(nf):
                grid = out[k].view(nz, ny, nx)
                rg.column_argmax(grid)
                rg.constant_altitude_ppi(grid, geom, 4000.0)
        variants = [("row", lambda: g.apply(out)), ("row+k3", row_k3)]
        for p in [int(x) for x in args.pieces.split(",")]:
            variants.append((f"col/p{p}", lambda p=p: g.apply_columns(out=out, z_pieces=p)))
            variants.append((f"col/p{p}/plain", lambda p=p: g.apply_columns(out=out, z_pieces=p, ordered=False)))
            variants.append((f"prod/p{p}/plain", lambda p=p: g.apply_columns(out=None, level_planes=planes, keep_lo=keep_lo, col_max=cmax,
                                                                             col_arg=carg, z_pieces=p, ordered=False)))
            variants.append((f"prod/p{p}", lambda p=p: g.apply_columns(out=None, level_planes=planes, keep_lo=keep_lo, col_max=cmax,
                                                                       col_arg=carg, z_pieces=p)))
        times = {v[0]: [] for v in variants}
        rng = np.random.default_rng(7)
        for r in range(args.rounds + 1):
            order = list(range(len(variants))) if r == 0 else list(rng.permutation(len(variants)))
            for vi in order:
                ms = timed(variants[vi][1])
                if r:
                    times[variants[vi][0]].append(ms)
        own = g.compact_bytes()
        k3_max = [rg.column_argmax(ref[k].view(nz, ny, nx)) for k in range(nf)]
        for name, fn in variants:
            ms = float(np.median(times[name]))
            out.fill_(-3.0); cmax.fill_(-3.0); carg.fill_(-3)
            fn()
            same = None
            bytes_ = own
            if name.startswith("prod"):
                same = all(bool(torch.equal(cmax[k].view(torch.int32), k3_max[k][0].view(torch.int32)))
                           and bool(torch.equal(carg[k], k3_max[k][1]))
                           and bool(torch.equal(planes[k].view(torch.int32),
                                                ref[k].view(nz, ny, nx)[keep_lo:keep_lo + n_keep].contiguous().view(torch.int32)))
                           for k in range(nf))
                bytes_ = own - nf * 4 * n_vox + nf * 4 * ny * nx * (n_keep + 2)
            elif name != "row+k3":
                same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            rec["runs"].append({"fields": nf, "kernel": name, "ms": round(ms, 4), "min_ms": round(float(np.min(times[name])), 4),
                                "own_TBps": round(bytes_ / ms / 1e9, 3), "frac_of_8TBps": round(bytes_ / ms / 1e9 / 8.0, 4),
                                "same_bits": same})
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
