#!/usr/bin/env python3
"""Is a PHYSICALLY CONTIGUOUS record array (hipExtMallocWithFlags(hipDeviceMallocContiguous)) always a good placement for
the row-wise kernel?  The bench geometry's records copied into several default allocations and several contiguous ones
(earlier ones kept, so that every copy gets new memory); kernel time on each, the output grid first in a default, then in a
contiguous allocation."""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class RawBuf:
    """Just enough of a tensor for CsrGridder.apply: a device pointer and a shape."""

    def __init__(self, ptr, shape):
        self._ptr, self.shape = ptr, shape

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return int(np.prod(self.shape))

    def element_size(self):
        return 4


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder
    rg.load_library()
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
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
    out_default = torch.empty((1, g.n_vox), dtype=torch.float32, device=dev)
    compact = g.compact
    src = compact.rec
    nbytes = src.numel() * 4

    def alloc(size, flag):
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), size, flag)
        return p.value if rc == 0 and p.value else None

    out_contig_ptr = alloc(g.n_vox * 4, 0x4)
    outs = [("out_default", out_default)]
    if out_contig_ptr:
        outs.append(("out_contiguous", RawBuf(out_contig_ptr, (1, g.n_vox))))

    def timed(out, reps=5):
        g.apply(out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.apply(out); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 3)

    rows = []
    for trial in range(8):
        kind = "contiguous" if trial % 2 else "default"
        p = alloc(nbytes, 0x4 if trial % 2 else 0x0)
        if p is None:
            rows.append({"placement": trial, "records": kind, "allocation": "refused"})
            print(json.dumps(rows[-1]), flush=True)
            continue
        hip.hipMemcpy(p, src.data_ptr(), nbytes, 3)
        torch.cuda.synchronize()
        compact.rec = RawBuf(p, tuple(src.shape))
        row = {"placement": trial, "records": kind, "address": hex(p)}
        for name, out in outs:
            row[name + "_ms"] = timed(out)
        rows.append(row)
        print(json.dumps(row), flush=True)
    compact.rec = src
    json.dump(rows, open("gpurun_out/exp_placement8.json", "w"), indent=1)


if __name__ == "__main__":
    main()
