#!/bin/bash
# Round 4, second half: rocprofv3 passes of the kernels that changed (eight-field pass of config 5, byte-mask three-field pass
# of config 3): kernel trace + stats, then PMC passes (never combined).  Summaries: tools/summarize_profile.py r04b_c5 / r04b_c3
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r04b_*
timeout -k 10 280 bash tools/profile_bench.sh r04b_c5 --config C5 > gpurun_out/r04b_p1.log 2>&1; tail -1 gpurun_out/r04b_p1.log
timeout -k 10 240 bash tools/profile_bench.sh r04b_c3 --config C2 --fields 3 > gpurun_out/r04b_p2.log 2>&1; tail -1 gpurun_out/r04b_p2.log
