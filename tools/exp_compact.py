#!/usr/bin/env python3
"""Compact-CSR kernel (rg_csr_compact_apply_f32) against the standard one: build time of the compact copy, bit
equality, interleaved timing.   python tools/exp_compact.py [METRIC|C2|C4]"""
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
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder
    name = sys.argv[1] if len(sys.argv) > 1 else "METRIC"
    cfg = synthetic.CONFIGS[name]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    csr = search.build_csr("barnes2")
    geom = rg.GridGeometry.from_device(cfg["grid_shape"], cfg["grid_limits"], csr, 17000.0)
    dev = search.dev
    del search
    f = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(dev)
    m = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g_c = CsrGridder(geom, f.numel(), 1, device=dev, compact=True)
    torch.cuda.synchronize()
    rec = {"pairs": csr.n_pairs, "compact_build_s": round(time.perf_counter() - t0, 2)}
    if g_c.compact is None:
        print(name, json.dumps(dict(rec, compactable=False)))
        return
    c = g_c.compact
    rec.update(dict_entries=c.n_dict, max_dict=c.max_dict, window_cap=c.window_cap, compact_GB=round(c.nbytes() / 1e9, 2),
               bytes_std_GB=round(g_c.algorithmic_bytes() / 1e9, 2), bytes_compact_GB=round(g_c.compact_bytes() / 1e9, 2))
    g_s = CsrGridder(geom, f.numel(), 1, device=dev)
    g_c.pack([f], [m]); g_s.pack([f], [m])
    out_c = torch.empty((1, g_c.n_vox), dtype=torch.float32, device=dev)
    out_s = torch.empty_like(out_c)
    g_c.apply(out_c); g_s.apply(out_s)
    torch.cuda.synchronize()
    rec["bit_identical"] = bool(torch.equal(out_c.view(torch.int32), out_s.view(torch.int32)))
    lib = _native.load_library()
    times = {"std": [], "compact": []}
    for tile in (0, 256, 384):
        times[f"compact_t{tile}"] = []

    def run_tile(tile):
        _native.check(lib.rg_csr_compact_apply_f32(
            _native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(c.local_idx), _native.ptr(csr.weights),
            _native.ptr(c.dict_ptr), _native.ptr(c.dict), g_c.n_vox, csr.n_pairs, _native.ptr(g_c.packed), g_c.n_gates,
            float("nan"), _native.ptr(out_c), c.window_cap, tile, _native.stream_ptr()), "compact")

    for _ in range(7):
        for k, fn in (("std", lambda: g_s.apply(out_s)), ("compact_t0", lambda: run_tile(0)),
                      ("compact_t256", lambda: run_tile(256)), ("compact_t384", lambda: run_tile(384))):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            times[k].append(a.elapsed_time(b))
    for k, v in times.items():
        if v:
            rec[k + "_ms"] = round(float(np.median(v)), 3)
    print(name, json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
