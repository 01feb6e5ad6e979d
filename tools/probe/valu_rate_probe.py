#!/usr/bin/env python3
"""Cycles a wave64 VALU instruction of each kind occupies a SIMD on this chip (tools/probe/valu_rate_probe.hip): 256 CUs x 8
wavefronts per SIMD, 8 independent instructions per iteration.  cycles = time x clock / (instructions per SIMD); the clock is taken
from the plain v_fma_f32 row (4 cycles by construction of the SIMD16: its row reads 4.00)."""
import ctypes
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libvalu_rate_probe.so"))
lib.valu_rate_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
names = ["v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32", "v_mul_legacy_f32", "v_cndmask_b32",
         "v_cvt_pk_f32_fp8", "v_cvt_f32_ubyte0", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_exp_f32", "v_and_or_b32", "v_mov_b32_dpp"]
out = torch.zeros(256, dtype=torch.float32, device="cuda")
blocks, iters = 256 * 8, 20000          # 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
res = {}
for kind, name in enumerate(names):
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.valu_rate_probe(kind, blocks, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record(); e1.synchronize()
        assert rc == 0, rc
        if r:
            ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    per_simd = 8 * iters * 8            # wavefronts per SIMD x iterations x instructions
    res[name] = {"ms": round(ms, 3), "ns_per_instruction_per_simd": round(ms * 1e6 / per_simd, 4)}
base = res["v_fma_f32"]["ns_per_instruction_per_simd"]
for name, r in res.items():
    r["cycles_if_v_fma_f32_is_4"] = round(4.0 * r["ns_per_instruction_per_simd"] / base, 2)
    print(f"{name:34s} {r['ms']:9.3f} ms   {r['cycles_if_v_fma_f32_is_4']:.2f} cycles")
res["implied_clock_GHz"] = round(4.0 / base, 3)
print("implied clock (v_fma_f32 = 4 cycles):", res["implied_clock_GHz"], "GHz")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/valu_rate_probe.json", "w"), indent=1)
