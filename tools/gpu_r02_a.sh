#!/bin/bash
# Round-2 GPU call A: full GPU test-suite, default bench line, config-2/3 lines, config-3 profile (before the multi-field compact kernel)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -m radar_processor_amd.build > gpurun_out/r02a_build.log 2>&1 || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a_gpu_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r02a_gpu_tests.log
tail -25 gpurun_out/r02a_gpu_tests.log
timeout -k 10 400 python3 bench.py > gpurun_out/r02a_bench_default.json 2> gpurun_out/r02a_bench_default.log || { tail -20 gpurun_out/r02a_bench_default.log; exit 1; }
cat gpurun_out/r02a_bench_default.json
timeout -k 10 200 python3 bench.py --config C2 --no-cpu-baseline > gpurun_out/r02a_bench_c2.json 2> gpurun_out/r02a_bench_c2.log
timeout -k 10 200 python3 bench.py --config C2 --fields 3 --no-cpu-baseline > gpurun_out/r02a_bench_c3.json 2> gpurun_out/r02a_bench_c3.log
cat gpurun_out/r02a_bench_c2.json gpurun_out/r02a_bench_c3.json
timeout -k 10 300 bash tools/profile_bench.sh r02_c3_before --config C2 --fields 3 > gpurun_out/r02a_prof_c3.log 2>&1
tail -3 gpurun_out/r02a_prof_c3.log
