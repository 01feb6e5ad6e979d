#!/usr/bin/env python3
"""Follow-up to exp_placement5.py: with the 44 GB array that is read kept where it is, does the placement of the small
array that is WRITTEN (650 MB) move the time as well?  Two read placements x six write placements, dense 1 KiB per
workgroup and the bench grid's 4-line layout."""
import ctypes
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import torch
    probe = ctypes.CDLL(os.path.join(HERE, "probe", "libbw_probe.so"))
    probe.bw_read_blocked.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    probe.bw_read_blocked_write.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    block = 69632
    nbytes = 44_340_000_000 // block * block
    n_blocks = nbytes // block
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=4):
        fn()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 3)

    rows = []
    reads = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]
    writes = [torch.zeros(n_blocks * 264 + (4 << 20), dtype=torch.float32, device=dev) for _ in range(4)]
    for ia, a in enumerate(reads):
        ro = timed(lambda: probe.bw_read_blocked(a.data_ptr(), nbytes, block, sink.data_ptr(), stream))
        for iw, wr in enumerate(writes):
            row = {"read_array": ia, "read_address": hex(a.data_ptr()), "write_array": iw, "write_address": hex(wr.data_ptr()),
                   "read_only_ms": ro}
            for name, mode in (("dense", 0), ("grid_layout", 1), ("grid_layout_aligned_pieces", -1)):
                row[name + "_ms"] = timed(lambda: probe.bw_read_blocked_write(a.data_ptr(), nbytes, block, wr.data_ptr(), 1024,
                                                                              1, mode, stream))
            rows.append(row)
            print(json.dumps(row), flush=True)
    json.dump(rows, open("gpurun_out/exp_placement6.json", "w"), indent=1)


if __name__ == "__main__":
    main()
