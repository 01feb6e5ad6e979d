#!/usr/bin/env python3
"""K2 (rg_roi_grid_f32) on the bench grid: the shipped kernel against other builds of the library loaded next to it -- the
timing-only build whose dense stage does nothing (tools/build_experiments.py --tag k2a1 -DRG_K2_ABL=1: what is left is the
candidate side) and other block shapes (-DRG_K2_BX=8 / 16).  usage: exp_k2_breakdown.py name=lib.so ..."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.roi_grid import roi_grid_fields_device
    rg.load_library()
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS["METRIC"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], device=dev)   # (built by the in-tree library)
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    out = torch.empty((1, *cfg["grid_shape"]), dtype=torch.float32, device=dev)
    libs = {"shipped": _native.load_library()}
    for item in sys.argv[1:]:
        name, path = item.split("=")
        lib = ctypes.CDLL(os.path.abspath(path))
        for sym, (restype, argtypes) in _native.SIGNATURES.items():
            fn = getattr(lib, sym)
            fn.restype, fn.argtypes = restype, argtypes
        libs[name] = lib
    res = {}
    for name, lib in libs.items():
        _native._lib = lib                      # the module-level handle every wrapper goes through
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); roi_grid_fields_device(search, [f], [m], out=out); e1.record(); e1.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        res[name] = round(float(np.median(ts)), 3)
    print(json.dumps(res, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/exp_k2_breakdown.json", "w"), indent=1)


if __name__ == "__main__":
    main()
