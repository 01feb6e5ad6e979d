#!/bin/bash
# profiles of the round-2 kernels: default (METRIC, K1c), reference format (K1), config 2, config 3, config 4
set -u
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k metric > gpurun_out/r02g_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r02g_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 240 bash tools/profile_bench.sh r02_metric > gpurun_out/r02g_p1.log 2>&1; tail -1 gpurun_out/r02g_p1.log
timeout -k 10 240 bash tools/profile_bench.sh r02_metric_k1 --no-compact > gpurun_out/r02g_p2.log 2>&1; tail -1 gpurun_out/r02g_p2.log
timeout -k 10 200 bash tools/profile_bench.sh r02_c2 --config C2 > gpurun_out/r02g_p3.log 2>&1; tail -1 gpurun_out/r02g_p3.log
timeout -k 10 200 bash tools/profile_bench.sh r02_c3 --config C2 --fields 3 > gpurun_out/r02g_p4.log 2>&1; tail -1 gpurun_out/r02g_p4.log
timeout -k 10 400 bash tools/profile_bench.sh r02_c4 --config C4 > gpurun_out/r02g_p5.log 2>&1; tail -1 gpurun_out/r02g_p5.log
