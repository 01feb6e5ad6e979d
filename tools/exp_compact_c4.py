#!/usr/bin/env python3
"""Dictionary-size distribution and compact-kernel timing vs window size on config 4 (compact-only layout)."""
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    cfg = synthetic.CONFIGS[name]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp,
                                        layout="compact")
    dev = _native.device()
    csr, c = geom.device_csr(dev), geom.device_compact(dev)
    d = (c.dict_ptr[1:] - c.dict_ptr[:-1]).cpu().numpy()
    rec = {"pairs": csr.n_pairs, "dict_entries": c.n_dict, "window_cap": c.window_cap,
           "p50": int(np.percentile(d, 50)), "p90": int(np.percentile(d, 90)), "p99": int(np.percentile(d, 99)),
           "p999": int(np.percentile(d, 99.9)), "max": int(d.max())}
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    g = CsrGridder(geom, f.numel(), 1, device=dev)
    g.pack([f], [m])
    out = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
    lib = _native.load_library()
    for cap in (2048, 3072, 4096, 5120, 6144, 8192):
        for tile in (0, 256):
            def run():
                _native.check(lib.rg_csr_compact_apply_f32(
                    _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.local_idx), _native.ptr(csr.weights),
                    _native.ptr(c.dict_ptr), _native.ptr(c.dict), g.n_vox, csr.n_pairs, _native.ptr(g.packed), g.n_gates,
                    float("nan"), _native.ptr(out), cap, tile, _native.stream_ptr()), "compact")
            run(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); run(); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
            rec[f"cap{cap}_tile{tile}_ms"] = round(min(ts), 2)
    print(name, json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
