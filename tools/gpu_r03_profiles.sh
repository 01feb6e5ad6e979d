#!/bin/bash
# Round 3 evidence, part B: rocprofv3 passes of the shipped kernels (kernel trace + stats, then PMC passes, never combined)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r03_*
timeout -k 10 260 bash tools/profile_bench.sh r03_metric > gpurun_out/r03_p1.log 2>&1; tail -1 gpurun_out/r03_p1.log
timeout -k 10 220 bash tools/profile_bench.sh r03_c3 --config C2 --fields 3 > gpurun_out/r03_p2.log 2>&1; tail -1 gpurun_out/r03_p2.log
timeout -k 10 220 bash tools/profile_bench.sh r03_c2 --config C2 > gpurun_out/r03_p3.log 2>&1; tail -1 gpurun_out/r03_p3.log
timeout -k 10 260 bash tools/profile_bench.sh r03_metric_f3 --fields 3 > gpurun_out/r03_p4.log 2>&1; tail -1 gpurun_out/r03_p4.log
timeout -k 10 320 bash tools/profile_bench.sh r03_c4 --config C4 > gpurun_out/r03_p5.log 2>&1; tail -1 gpurun_out/r03_p5.log
timeout -k 10 260 bash tools/profile_bench.sh r03_fused --mode fused > gpurun_out/r03_p6.log 2>&1; tail -1 gpurun_out/r03_p6.log
