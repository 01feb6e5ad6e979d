#!/bin/bash
# Round 4, final code: rocprofv3 passes of the default workload and of config 5 (kernel trace + stats, then PMC passes).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r04c_*
timeout -k 10 300 bash tools/profile_bench.sh r04c_metric > gpurun_out/r04c_p1.log 2>&1; tail -1 gpurun_out/r04c_p1.log
timeout -k 10 300 bash tools/profile_bench.sh r04c_c5 --config C5 > gpurun_out/r04c_p2.log 2>&1; tail -1 gpurun_out/r04c_p2.log
