#!/usr/bin/env python3
"""A/B timing of the CSR kernels on one configuration, one process, interleaved rounds:

    python tools/tune_compact.py [--config C2|METRIC|C4] [--fields 1,3] [--tiles 0,128,256] [--rounds 5]

For every field count: rg_csr_apply_f32 (K1, the reference's CSR format) and rg_csr_compact_apply_f32 (K1c TILE kernel over the
compact arrays; the row-wise kernel over the packed records has its own script, exp_rowwise.py) with each tile, median kernel time over the rounds, bytes per launch, TB/s, and whether K1c == K1 bit for bit for the
same tile.  Prints one JSON object."""
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
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--windows", default="", help="extra LDS windows (entries) to try for the compact kernel")
    args = ap.parse_args()
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
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
    rec = {"config": args.config, "pairs": geom.n_pairs(), "window_cap": compact.window_cap if compact else None,
           "n_dict": compact.n_dict if compact else None, "max_dict": compact.max_dict if compact else None,
           "dict_bytes_per_pair": round(4 * compact.n_dict / geom.n_pairs(), 4) if compact else None, "runs": []}
    if compact is not None:
        rec["fallback_fraction"] = {str(w): round(compact.fallback_fraction(w), 5) for w in (256, 512, 768, 1024, 1536, 2048, 4096)}

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1)

    for nf in [int(x) for x in args.fields.split(",")]:
        fl = [base_f[i % 3] for i in range(nf)]
        ml = [base_m[i % 3] for i in range(nf)]
        for tile in [int(x) for x in args.tiles.split(",")]:
            try:
                g_s = CsrGridder(geom, fl[0].numel(), nf, device=dev, tile=tile)
                g_c = CsrGridder(geom, fl[0].numel(), nf, device=dev, compact=True, tile=tile, packed=False)   # the tile kernel
            except Exception as exc:
                rec["runs"].append({"fields": nf, "tile": tile, "error": repr(exc)})
                continue
            if g_c.compact is None and compact is not None:
                g_c.compact, g_c.window = compact, compact.window_for(nf)
            out_s = torch.empty((nf, n_vox), dtype=torch.float32, device=dev)
            out_c = torch.empty_like(out_s)
            variants = [("k1", g_s, out_s, None)]
            if g_c.compact is not None:
                variants.append((f"k1c_w{g_c.window}", g_c, out_c, g_c.window))
                for w in [int(x) for x in args.windows.split(",") if x]:
                    variants.append((f"k1c_w{w}", g_c, out_c, w))
            for _, gr, _, _ in variants[:2]:
                gr.pack(fl, ml, qc if nf >= 3 else None)
            times = {v[0]: [] for v in variants}
            try:
                for r in range(args.rounds + 1):
                    for name, gr, out, w in variants:
                        if w is not None:
                            gr.window = w
                        ms = timed(lambda: gr.apply(out))
                        if r:
                            times[name].append(ms)
                g_c.window = variants[1][3] if len(variants) > 1 else 0
                g_c.apply(out_c)
                same = bool(torch.equal(out_s.view(torch.int32), out_c.view(torch.int32))) if g_c.compact is not None else None
            except Exception as exc:
                rec["runs"].append({"fields": nf, "tile": tile, "error": repr(exc)})
                continue
            for name, gr, _, w in variants:
                ms = float(np.median(times[name]))
                by = gr.compact_bytes() if name != "k1" else gr.algorithmic_bytes()
                rec["runs"].append({"fields": nf, "tile": tile, "kernel": name, "ms": round(ms, 4),
                                    "bytes": int(by), "TBps": round(by / ms / 1e9, 3),
                                    "ref_format_TBps": round(gr.algorithmic_bytes() / ms / 1e9, 3),
                                    "frac_8d": round(gr.algorithmic_bytes() / ms / 1e9 / 8.0, 4),
                                    "bit_identical_to_k1": same if name != "k1" else None})
            del out_s, out_c, g_s, g_c
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
