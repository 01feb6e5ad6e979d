#!/usr/bin/env python3
"""Which access pattern feels the placement of a 44 GB array?  Per placement of the bench geometry's record array: the
row-wise kernel's time, the grid-stride read probe, and a BLOCKED read probe (tools/probe/bw_probe.hip: every workgroup
reads its own contiguous block front to back -- the row-wise kernel's access front without its arithmetic) for block
sizes from 4 KiB to 4 MiB.  Build the probe library first (see tools/probe/bw_probe.py)."""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    probe = ctypes.CDLL(os.path.join(HERE, "probe", "libbw_probe.so"))
    probe.bw_read_blocked.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    probe.bw_read_linear.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS["METRIC"]
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
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=4):
        fn()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    rows, keep = [], []
    for trial in range(6):
        rec = compact.rec
        nbytes = rec.numel() * 4
        row = {"placement": trial, "address": hex(rec.data_ptr()),
               "rowwise_kernel_ms": round(timed(lambda: g.apply(out), 5), 4),
               "grid_stride_GBps": round(nbytes / timed(lambda: probe.bw_read_linear(rec.data_ptr(), nbytes, 8192, 4,
                                                                                     sink.data_ptr(), stream)) / 1e6, 1)}
        for kib in (4, 16, 68, 256, 1024, 4096):
            ms = timed(lambda: probe.bw_read_blocked(rec.data_ptr(), nbytes, kib * 1024, sink.data_ptr(), stream))
            row[f"blocked_{kib}KiB_GBps"] = round(nbytes / ms / 1e6, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
        if trial < 5:
            keep.append(compact.rec)
            compact.rec = compact.rec.clone()
            if len(keep) > 3:
                keep.pop(0)
    json.dump(rows, open("gpurun_out/exp_placement3.json", "w"), indent=1)


if __name__ == "__main__":
    main()
