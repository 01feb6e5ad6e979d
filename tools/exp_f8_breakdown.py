#!/usr/bin/env python3
"""Where does the eight-field row-wise pass spend its 14.4 ms on the bench grid?  Timing-only variants of the kernel
(rg_csr_compact_apply_packed_f32, tile = 2100 + code, experiment build only; results wrong by construction): no window
gather (1), no output store (2), no record loads (16), no window reads in the pair loop (40), and combinations."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_experiments            # the timing-only tile codes exist only in the -DRG_EXPERIMENTS build
build_experiments.use("abl")


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "METRIC"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    label = {0: "shipped", 2101: "no_window_gather", 2102: "no_store", 2116: "no_record_loads", 2140: "no_window_reads",
             2142: "no_window_reads_no_store", 2143: "no_window_reads_no_store_no_gather", 2159: "control_and_arithmetic_only",
             2119: "no_record_loads_no_store_no_gather"}
    g = CsrGridder(geom, f.numel(), 8, device=dev, compact=True)
    g.pack([f] * 8, [m] * 8)
    out = torch.empty((8, g.n_vox), dtype=torch.float32, device=dev)
    times = {c: [] for c in label}
    for r in range(8):
        for c in label:
            g.tile = c
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            if r:
                times[c].append(e0.elapsed_time(e1))
    res = {label[c]: round(float(np.median(t)), 4) for c, t in times.items()}
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/exp_f8_breakdown.json", "w"), indent=1)


if __name__ == "__main__":
    main()
