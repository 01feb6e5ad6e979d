#!/usr/bin/env python3
"""K2 (rg_roi_grid_f32) and the builder's count pass on the bench grid against the search structure: one gate list for all levels
or one per level (RoiSearch(per_level=...)), and the cell size as a multiple of the automatic one."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.roi_grid import roi_grid_fields_device
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "METRIC"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    out = torch.empty((1, *cfg["grid_shape"]), dtype=torch.float32, device=dev)
    base = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], device=dev, per_level=False)
    auto = base.cell_size
    ref = None
    res = {"auto_cell_m": auto, "runs": []}
    for per_level in (False, True):
        for k in ((1.0,) if not per_level else (1.0, 1.5, 2.0, 3.0, 4.0)):
            s = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], device=dev,
                             per_level=per_level, cell_size=auto * k)
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); roi_grid_fields_device(s, [f], [m], out=out); e1.record(); e1.synchronize()
                if r:
                    ts.append(e0.elapsed_time(e1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); n_pairs = s.count_pairs(); e1.record(); e1.synchronize()
            same = None
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(torch.nan_to_num(out, nan=-7e9), torch.nan_to_num(ref, nan=-7e9)))
            res["runs"].append({"per_level": per_level, "cell_factor": k, "entries": s.n_binned, "k2_ms": round(float(np.median(ts)), 3),
                                "count_ms": round(e0.elapsed_time(e1), 3), "pairs": int(n_pairs), "same_grid_bits_as_first": same})
            print(res["runs"][-1], flush=True)
            del s
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/exp_k2_cells.json", "w"), indent=1)


if __name__ == "__main__":
    main()
