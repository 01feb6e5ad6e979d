#!/bin/bash
# Round 4 evidence, part B: rocprofv3 passes of the shipped kernels (kernel trace + stats, then PMC passes, never combined)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r04_*
timeout -k 10 260 bash tools/profile_bench.sh r04_metric > gpurun_out/r04_p1.log 2>&1; tail -1 gpurun_out/r04_p1.log
timeout -k 10 220 bash tools/profile_bench.sh r04_c3 --config C2 --fields 3 > gpurun_out/r04_p2.log 2>&1; tail -1 gpurun_out/r04_p2.log
timeout -k 10 260 bash tools/profile_bench.sh r04_c5 --config C5 > gpurun_out/r04_p3.log 2>&1; tail -1 gpurun_out/r04_p3.log
timeout -k 10 260 bash tools/profile_bench.sh r04_c5fused --config C5 --products fused > gpurun_out/r04_p4.log 2>&1; tail -1 gpurun_out/r04_p4.log
timeout -k 10 220 bash tools/profile_bench.sh r04_c2 --config C2 > gpurun_out/r04_p5.log 2>&1; tail -1 gpurun_out/r04_p5.log
