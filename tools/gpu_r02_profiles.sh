#!/bin/bash
# final pass of round 2, part B: rocprofv3 passes of the shipped kernels (tags r02b_*: the row-wise kernel)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r02b_*
timeout -k 10 240 bash tools/profile_bench.sh r02b_metric > gpurun_out/r02z_p1.log 2>&1; tail -1 gpurun_out/r02z_p1.log
timeout -k 10 200 bash tools/profile_bench.sh r02b_c3 --config C2 --fields 3 > gpurun_out/r02z_p4.log 2>&1; tail -1 gpurun_out/r02z_p4.log
timeout -k 10 240 bash tools/profile_bench.sh r02b_metric_f3 --fields 3 > gpurun_out/r02z_p6.log 2>&1; tail -1 gpurun_out/r02z_p6.log
timeout -k 10 200 bash tools/profile_bench.sh r02b_c2 --config C2 > gpurun_out/r02z_p3.log 2>&1; tail -1 gpurun_out/r02z_p3.log
timeout -k 10 300 bash tools/profile_bench.sh r02b_c4 --config C4 > gpurun_out/r02z_p5.log 2>&1; tail -1 gpurun_out/r02z_p5.log
