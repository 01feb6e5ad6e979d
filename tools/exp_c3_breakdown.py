#!/usr/bin/env python3
"""Where does the three-field row-wise kernel spend config 3's 1.35 ms?  Timing-only variants of the kernel on the config-2
grid (rg_csr_compact_apply_packed_f32, tile = 2100 + code): no window gather (1), no output store (2), no record loads (16),
records fetched but not evaluated (80), neither gathered nor evaluated (81); one field for comparison."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_experiments            # the timing-only tile codes exist only in the -DRG_EXPERIMENTS build
build_experiments.use()


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
    names = ("DBZH", "ZDR", "RHOHV")
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=names)
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp)
    f = [torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields[n]))).to(dev) for n in names]
    m = [torch.from_numpy(np.ma.getmaskarray(vol.fields[n]).astype(np.uint8)).to(dev) for n in names]
    qc = rg.device_gate_mask(f[2], "below", 0.8)
    res = {}
    for nf, codes in ((3, (0, 2101, 2102, 2116, 2180, 2181)), (1, (0, 2101, 2102, 2116, 2180))):
        g = CsrGridder(geom, f[0].numel(), nf, device=dev, compact=True)
        g.pack(f[:nf], m[:nf], qc if nf == 3 else None)
        out = torch.empty((nf, g.n_vox), dtype=torch.float32, device=dev)
        times = {c: [] for c in codes}
        for r in range(12):
            for c in codes:
                g.tile = c
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.apply(out); e1.record(); e1.synchronize()
                if r:
                    times[c].append(e0.elapsed_time(e1))
        label = {0: "shipped", 2101: "no_window_gather", 2102: "no_store", 2116: "no_record_loads", 2180: "records_not_evaluated",
                 2181: "not_evaluated_not_gathered"}
        res[f"F{nf}"] = {label[c]: round(float(np.median(t)), 4) for c, t in times.items()}
    print(json.dumps(res, indent=1))
    json.dump(res, open("gpurun_out/exp_c3_breakdown.json", "w"), indent=1)


if __name__ == "__main__":
    main()
