#!/usr/bin/env python3
"""What does a byte WRITTEN cost next to a streaming read on this part?  tools/probe/bw_probe.hip's blocked read (44 GB,
68 KiB per workgroup) with 0, 256 B, 1 KiB, 4 KiB written per workgroup (0.16 / 0.65 / 2.6 GB in total, dense), and the
writes alone."""
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
    a = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    wr = torch.zeros(n_blocks * 1024 + (4 << 20), dtype=torch.float32, device=dev)       # up to 4 KiB per block
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=5):
        fn()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 3)

    res = {"read_GB": round(nbytes / 1e9, 2),
           "read_only_ms": timed(lambda: probe.bw_read_blocked(a.data_ptr(), nbytes, block, sink.data_ptr(), stream))}
    for wb in (256, 1024, 4096):
        res[f"read_plus_{wb}B_per_workgroup_ms"] = timed(lambda: probe.bw_read_blocked_write(
            a.data_ptr(), nbytes, block, wr.data_ptr(), wb, 1, 0, stream))
        res[f"written_GB_{wb}B"] = round(n_blocks * wb / 1e9, 3)
        # the writes alone: the same launch over a 256-byte "read" block per workgroup
        res[f"write_only_{wb}B_ms"] = timed(lambda: probe.bw_read_blocked_write(
            a.data_ptr(), n_blocks * 256, 256, wr.data_ptr(), wb, 1, 0, stream))
    fill = torch.empty(160_000_000, dtype=torch.float32, device=dev)
    res["torch_fill_640MB_ms"] = timed(lambda: fill.fill_(1.0))
    print(json.dumps(res, indent=1))
    json.dump(res, open("gpurun_out/exp_write_cost.json", "w"), indent=1)


if __name__ == "__main__":
    main()
