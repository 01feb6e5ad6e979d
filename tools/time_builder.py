#!/usr/bin/env python3
"""Times the phases of the GPU geometry build (bin, count, scan, fill) with events; used for DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    for name in sys.argv[1:] or ["METRIC"]:
        cfg = synthetic.CONFIGS[name]
        vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
        rec = {}
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            csr = search.build_csr("barnes2")
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rec[f"rep{rep}"] = {"search_s": round(t1 - t0, 4), "build_csr_s": round(t2 - t1, 4),
                                "pairs": int(csr.gate_indices.numel())}
            del csr, search
        print(name, json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
