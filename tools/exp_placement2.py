#!/usr/bin/env python3
"""Round 3 follow-up to exp_placement.py: WHAT about the placement of the record array moves the row-wise kernel's time?
One process, the bench geometry in the packed-only layout.  Per placement (a fresh 44 GB allocation, earlier ones kept
so that the memory really is new): device address, kernel time (median / min of 6), and what rg_stream_read_probe reads
from that very array.  Then, inside ONE allocation, the same records shifted by a few byte offsets (same physical pages,
other address bits per record): if the time follows the offset, the cause is the address pattern (channel / bank
hashing of the concurrent streams), if it follows the allocation only, it is the pages themselves (fragment size / TLB
reach, or how they spread over the stacks)."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder
    lib = rg.load_library()
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
    sink = torch.zeros(4, dtype=torch.float32, device=dev)

    def kernel_ms():
        times = []
        for r in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            if r >= 2:
                times.append(e0.elapsed_time(e1))
        return round(float(np.median(times)), 4), round(float(np.min(times)), 4)

    def probe_gbs(t):
        nbytes = t.numel() * t.element_size() // 16 * 16
        best = None
        for i in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _native.check(lib.rg_stream_read_probe(_native.ptr(t), nbytes, _native.ptr(sink), _native.stream_ptr()), "probe")
            e1.record(); e1.synchronize()
            if i:
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
        return round(nbytes / (best * 1e-3) / 1e9, 1)

    def describe(t):
        a = t.data_ptr()
        return {"address": hex(a), "mod_2MiB": a % (2 << 20), "mod_1GiB_MiB": round((a % (1 << 30)) / (1 << 20), 3)}

    rows = []
    keep = []
    for trial in range(5):
        med, mn = kernel_ms()
        rows.append({"placement": trial, **describe(compact.rec), "kernel_ms_median": med, "kernel_ms_min": mn,
                     "probe_GBps": probe_gbs(compact.rec)})
        print(json.dumps(rows[-1]), flush=True)
        if trial < 4:
            keep.append(compact.rec)
            compact.rec = compact.rec.clone()
            if len(keep) > 3:
                keep.pop(0)
    # ---- byte offsets inside one allocation -------------------------------------------------------------------
    src = compact.rec
    n = src.shape[0]
    keep.clear()
    torch.cuda.empty_cache()
    pad_rec = (1 << 30) // 16 + (4 << 20) // 16              # room for shifts up to 1 GiB + 4 MiB
    big = torch.empty((n + pad_rec, 4), dtype=torch.int32, device=dev)
    shifts = []
    for off_b in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 1 << 30, 0):
        k = off_b // 16
        view = big[k:k + n]
        view.copy_(src)
        compact.rec = view
        med, mn = kernel_ms()
        shifts.append({"offset_bytes": off_b, **describe(view), "kernel_ms_median": med, "kernel_ms_min": mn,
                       "probe_GBps": probe_gbs(view)})
        print(json.dumps(shifts[-1]), flush=True)
    compact.rec = src
    med, mn = kernel_ms()
    print(json.dumps({"back_on_source": describe(src), "kernel_ms_median": med, "kernel_ms_min": mn}))
    json.dump({"placements": rows, "shifts_in_one_allocation": shifts, "back_on_source": med},
              open("gpurun_out/exp_placement2.json", "w"), indent=1)


if __name__ == "__main__":
    main()
