#!/usr/bin/env python3
"""What would a denser record buy the one-field kernel?  Timing-only variant (experiment build, tile = 2173): every segment reads
its records 1/7 closer to the start of the array, so neighbouring streams overlap and 14 % fewer distinct bytes come from HBM with
the same loads, the same arithmetic (results wrong by construction) -- the HBM side of a 7-pairs-per-32-bytes record."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_experiments
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
    g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    g.pack([f], [m])
    out = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
    label = {0: "shipped", 2173: "streams_overlap_by_a_seventh", 2102: "no_store", 2116: "no_record_loads"}
    times = {c: [] for c in label}
    for r in range(10):
        for c in label:
            g.tile = c
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            if r:
                times[c].append(e0.elapsed_time(e1))
    res = {label[c]: round(float(np.median(t)), 4) for c, t in times.items()}
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/exp_stream_bytes.json", "w"), indent=1)


if __name__ == "__main__":
    main()
