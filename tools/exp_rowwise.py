#!/usr/bin/env python3
"""Row-wise kernel (rg_csr_compact_apply_packed_f32) against the tile kernel over the same packed records, one configuration:

    python tools/exp_rowwise.py [--config C2|METRIC] [--fields 1,3] [--codes 0,2004,2008,2074,2076] [--rounds 5]

``--codes`` are ``tile`` arguments of the entry point: 0 = the shipped row-wise kernel, 2000 + h = a diagnostic lane split
(h = 1..64 lanes per row, h = 70 + t: aim for t records per lane and row).  Prints one JSON object: per variant the median
kernel time, TB/s in its own bytes and in SURVEY 8(d)'s, and the largest relative difference to the tile kernel's grids (the
two sum in different orders).  profiles/r02_rowwise_sweep.json was produced with this script at the commit that
introduced the kernel, when records per step, window entry size and the home of the row sums were still template
parameters selectable through the code."""
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
    ap.add_argument("--fields", default="3")
    ap.add_argument("--codes", default="0,2004,2008,2016,2074,2076,2078")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--libs", default="", help="name=path,... : other builds of the library to time next to the in-tree one, "
                                                "same process, same arrays (every code is run through every library)")
    args = ap.parse_args()
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    libs = []
    if args.libs:
        import ctypes
        from radar_processor_amd import _native
        for item in args.libs.split(","):
            name, path = item.split("=")
            lib = ctypes.CDLL(os.path.abspath(path))
            for sym, (restype, argtypes) in _native.SIGNATURES.items():
                fn = getattr(lib, sym)
                fn.restype, fn.argtypes = restype, argtypes
            libs.append((name, lib))
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[args.config]
    names = ("DBZH", "ZDR", "RHOHV")
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=names)
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
    base_f = [torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields[n]))).to(dev) for n in names]
    base_m = [torch.from_numpy(np.ma.getmaskarray(vol.fields[n]).astype(np.uint8)).to(dev) for n in names]
    qc = rg.device_gate_mask(base_f[2], "below", 0.8)
    n_vox = int(np.prod(cfg["grid_shape"]))
    compact = geom.device_compact(dev)
    rec = {"config": args.config, "pairs": geom.n_pairs(), "window_cap": compact.window_cap, "runs": []}

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1)

    for nf in [int(x) for x in args.fields.split(",")]:
        fl = [base_f[i % 3] for i in range(nf)]
        ml = [base_m[i % 3] for i in range(nf)]
        ref = CsrGridder(geom, fl[0].numel(), nf, device=dev, compact=True)
        if ref.compact is None:
            ref.compact, ref.window = compact, compact.window_for(nf)
            ref.packed_stream = compact.ensure_packed(ref.csr)
        out_ref = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
        if nf <= 4:
            ref.tile = 384                                   # the tile kernel over the packed records
            ref.pack(fl, ml, qc if nf >= 3 else None)
            ref.apply(out_ref)
            variants = [("tile", ref, out_ref)]
        else:                # 5-8 fields: the reference grids come from row-wise passes of <= 4 fields (another add order)
            for lo in range(0, nf, 4):
                part = CsrGridder(geom, fl[0].numel(), min(4, nf - lo), device=dev, compact=True)
                part.pack(fl[lo:lo + 4], ml[lo:lo + 4], qc)
                part.apply(out_ref[lo:lo + 4])
                if lo == 0:
                    variants = [("two_passes_first", part, torch.empty((part.n_fields, n_vox), dtype=torch.float32, device=dev))]
                del part
            ref.pack(fl, ml, qc)
            keep = out_ref.clone()
        for code in [int(x) for x in args.codes.split(",")]:
            g = CsrGridder(geom, fl[0].numel(), nf, device=dev, compact=True, tile=code)
            g.compact, g.packed_stream = compact, True
            g.window = compact.window_for(nf)
            g.packed = ref.packed
            # every variant writes the SAME output buffer while it is timed: where a grid lies in memory moves a 1 ms kernel
            # by several per cent (round 3, tools/exp_placement5.py), which would drown the differences looked for here
            variants.append((f"row{code}", g, out_ref))
            for lname, lib in libs:
                g2 = CsrGridder(geom, fl[0].numel(), nf, device=dev, compact=True, tile=code)
                g2.compact, g2.packed_stream, g2.window, g2.packed, g2.lib = compact, True, g.window, ref.packed, lib
                variants.append((f"row{code}@{lname}", g2, out_ref))
        times = {v[0]: [] for v in variants}
        rng = np.random.default_rng(7)
        for r in range(args.rounds + 1):
            order = list(range(len(variants))) if r == 0 else list(rng.permutation(len(variants)))   # position effects
            for vi in order:                                                                            # are a few per cent
                name, g, out = variants[vi]
                ms = timed(lambda: g.apply(out))
                if r:
                    times[name].append(ms)
        if nf <= 4:
            ref.apply(out_ref)
        else:
            out_ref.copy_(keep)
        a = out_ref.double()
        first_row = None
        scratch = torch.empty_like(out_ref)
        for vi, (name, g, _) in enumerate(variants):
            ms = float(np.median(times[name]))
            if g.n_fields != nf:                        # a partial pass timed next to the others: no grids to compare
                rec["runs"].append({"fields": g.n_fields, "kernel": name, "ms": round(ms, 4)})
                continue
            g.apply(scratch)                            # the values, outside the timed region
            out = scratch
            if first_row is None and name.startswith("row"):
                first_row = scratch.clone()
            b = out.double()
            nan_same = bool(torch.equal(torch.isnan(a), torch.isnan(b)))
            ok = ~torch.isnan(a)
            rel = float(((a[ok] - b[ok]).abs() / a[ok].abs().clamp_min(1e-3)).max()) if nan_same else None
            rec["runs"].append({"fields": nf, "kernel": name, "ms": round(ms, 4),
                                "own_TBps": round(g.compact_bytes() / ms / 1e9, 3),
                                "frac_8d": round(g.algorithmic_bytes() / ms / 1e9 / 8.0, 4),
                                "nan_pattern_same": nan_same, "max_rel_diff_to_tile": rel,
                                "bit_identical": bool(torch.equal(out.view(torch.int32), out_ref.view(torch.int32))),
                                "same_bits_as_first_row_variant": bool(torch.equal(
                                    torch.nan_to_num(out, nan=-7e9), torch.nan_to_num(first_row, nan=-7e9)))
                                if first_row is not None else None})
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
