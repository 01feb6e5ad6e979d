#!/usr/bin/env python3
"""Measured read-bandwidth ceilings on the GPU box (diagnostic; see bw_probe.hip).  Build first:
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probe/bw_probe.hip -o tools/probe/libbw_probe.so"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import torch
    lib = ctypes.CDLL(os.path.join(HERE, "libbw_probe.so"))
    lib.bw_read_linear.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.bw_read_two_streams.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p]
    dev = torch.device("cuda")
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_313_732_728
    a = torch.zeros(n, dtype=torch.int32, device=dev)
    b = torch.ones(n, dtype=torch.float32, device=dev)
    out = torch.zeros(16, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    rec = {}
    for blocks in (2048, 8192, 32768):
        for unroll in (2, 4, 8):
            ms = timed(lambda: lib.bw_read_linear(a.data_ptr(), n * 4, blocks, unroll, out.data_ptr(), stream))
            rec[f"linear_b{blocks}_u{unroll}_TBps"] = round(n * 4 / ms / 1e9, 3)
    for per_wave in (3328, 8192, 65536, 1 << 20):
        ms = timed(lambda: lib.bw_read_two_streams(a.data_ptr(), b.data_ptr(), n, per_wave, out.data_ptr(), stream))
        rec[f"two_streams_{per_wave}_TBps"] = round(n * 8 / ms / 1e9, 3)
    ms = timed(lambda: a.sum())
    rec["torch_sum_i32_TBps"] = round(n * 4 / ms / 1e9, 3)
    c = torch.empty_like(b)
    ms = timed(lambda: c.copy_(b))
    rec["torch_copy_f32_read_plus_write_TBps"] = round(n * 8 / ms / 1e9, 3)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
