#!/usr/bin/env python3
"""Does the KIND of allocation of the written array change what a trickle of writes costs a streaming read?  The blocked
read probe of exp_placement5.py (44 GB, 68 KiB per workgroup) with 1 KiB written per workgroup into an array obtained from
hipExtMallocWithFlags: default (coarse-grained), fine-grained, uncached, physically contiguous -- and the array that is
READ physically contiguous too, if the driver grants 44 GB of it."""
import ctypes
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = {"default": 0x0, "finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}


def main():
    import torch
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    probe = ctypes.CDLL(os.path.join(HERE, "probe", "libbw_probe.so"))
    probe.bw_read_blocked.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    probe.bw_read_blocked_write.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    block = 69632
    nbytes = 44_340_000_000 // block * block
    n_blocks = nbytes // block
    wbytes = (n_blocks * 256 + (4 << 20)) * 4
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def alloc(size, kind):
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), size, FLAGS[kind])
        if rc != 0 or not p.value:
            return None
        hip.hipMemset(p, 0, size)
        torch.cuda.synchronize()
        return p

    def timed(fn, reps=4):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 3)

    rows = []
    reads = [("default", alloc(nbytes, "default")), ("default_2", alloc(nbytes, "default")), ("contiguous", alloc(nbytes, "contiguous"))]
    for rname, a in reads:
        if a is None:
            rows.append({"read_array": rname, "allocation": "refused"})
            print(json.dumps(rows[-1]), flush=True)
            continue
        ro = timed(lambda: probe.bw_read_blocked(a, nbytes, block, sink.data_ptr(), stream))
        for wname in ("default", "finegrained", "uncached", "contiguous"):
            w = alloc(wbytes, wname)
            row = {"read_array": rname, "read_address": hex(a.value), "write_array": wname, "read_only_ms": ro}
            if w is None:
                row["allocation"] = "refused"
            else:
                row["write_address"] = hex(w.value)
                for name, mode in (("dense", 0), ("grid_layout", 1)):
                    row[name + "_ms"] = timed(lambda: probe.bw_read_blocked_write(a, nbytes, block, w, 1024, 1, mode, stream))
                hip.hipFree(w)
            rows.append(row)
            print(json.dumps(row), flush=True)
    json.dump(rows, open("gpurun_out/exp_placement7.json", "w"), indent=1)


if __name__ == "__main__":
    main()
