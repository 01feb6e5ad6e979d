#!/usr/bin/env python3
"""Does the row-wise kernel's time depend on WHERE the records live?  One process, the bench geometry: time the kernel,
then move the 44 GB record array to a fresh allocation (same contents, another address) and time it again, several times.
Prints the median kernel time per placement and the device address of each copy."""
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
    cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "METRIC"]
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
    keep = []          # earlier copies stay allocated so that every new copy really gets new memory
    rows = []
    for trial in range(6):
        times = []
        for r in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            if r >= 2:
                times.append(e0.elapsed_time(e1))
        rows.append({"placement": trial, "rec_address": hex(compact.rec.data_ptr()), "ms_median": round(float(np.median(times)), 4),
                     "ms_min": round(float(np.min(times)), 4)})
        if trial < 5:
            keep.append(compact.rec)
            compact.rec = compact.rec.clone()
            if len(keep) > 3:
                keep.pop(0)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
