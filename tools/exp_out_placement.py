#!/usr/bin/env python3
"""Does the placement of the OUTPUT grid matter once the record placement is settled?  Bench geometry, one field: settle the
records (best of four), then time the kernel into six different allocations of the grid (all alive at once)."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    g = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    g.pack([f], [m])
    outs = [torch.empty((1, g.n_vox), dtype=torch.float32, device=dev) for _ in range(6)]
    rep = g.settle_records(tries=4, out=outs[0])

    def probe(out, n=5):
        ts = []
        g.apply(out)
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 4)
    res = {"records_settled": rep, "out_probe_ms": [probe(o) for o in outs], "again": [probe(o) for o in outs]}
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/exp_out_placement.json", "w"), indent=1)


if __name__ == "__main__":
    main()
