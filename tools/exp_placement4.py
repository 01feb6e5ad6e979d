#!/usr/bin/env python3
"""Which PART of the row-wise kernel feels the placement of the record array?  Per placement (fresh 44 GB allocation of
the same records): the kernel as shipped and timing-only ablations of it (rg_csr_compact_apply_packed_f32, tile = 2100 +
DIAG bits: 1 = window not gathered, 2 = no output store, 4 / 8 = sc0 / nt on the record loads, 16 = no record loads) and
a few lane splits (tile = 2000 + lanes per row), and the number of consecutive chunks a workgroup takes (tile = 2200 + n;
n = 1 is the round-2 kernel: the store's acknowledgement at the end of every workgroup's life)."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_experiments            # the timing-only tile codes exist only in the -DRG_EXPERIMENTS build
build_experiments.use()

VARIANTS = [("shipped", 0), ("no_window_gather", 2101), ("no_store", 2102), ("no_gather_no_store", 2103),
            ("rec_sc0", 2104), ("rec_nt", 2108), ("no_rec_loads", 2116), ("no_rec_no_gather", 2117),
            ("no_rec_no_gather_no_store", 2119), ("lanes4", 2004), ("lanes8", 2008), ("lanes16", 2016),
            ("chunks_per_wg_1", 2201), ("chunks_per_wg_2", 2202), ("chunks_per_wg_4", 2204), ("chunks_per_wg_8", 2208),
            ("chunks_per_wg_16", 2216), ("chunks_per_wg_32", 2232)]
if len(sys.argv) > 1 and sys.argv[1] == "store":
    VARIANTS = [("shipped", 0), ("chunks_per_wg_1", 2201), ("no_store", 2102), ("store_dense_dispatch_order", 2132),
                ("store_whole_lines_only", 2164), ("store_nt", 2170), ("store_sc0_sc1", 2171), ("store_sc1", 2172)]
elif len(sys.argv) > 1 and sys.argv[1] == "cpw":
    VARIANTS = [v for v in VARIANTS if v[0] in ("shipped", "no_store") or v[0].startswith("chunks")]


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS["METRIC"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp,
                                        layout="packed")
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    g.pack([f], [m])
    out = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
    compact = g.compact

    def timed(reps=4):
        g.apply(out)
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 3)

    rows, keep = [], []
    for trial in range(6):
        row = {"placement": trial, "address": hex(compact.rec.data_ptr())}
        for name, tile in VARIANTS:
            g.tile = tile
            row[name] = timed()
        g.tile = 0
        rows.append(row)
        print(json.dumps(row), flush=True)
        if trial < 5:
            keep.append(compact.rec)
            compact.rec = compact.rec.clone()
            if len(keep) > 3:
                keep.pop(0)
    json.dump(rows, open("gpurun_out/exp_placement4.json", "w"), indent=1)


if __name__ == "__main__":
    main()
