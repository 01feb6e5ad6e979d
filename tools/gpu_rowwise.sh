#!/bin/bash
# row-wise kernel: tests of the touched files, smoke, bench lines
mkdir -p gpurun_out
python -m radar_processor_amd.build > gpurun_out/build.log 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/rowwise_tests.log 2>&1
rc=$?
tail -25 gpurun_out/rowwise_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/rowwise_smoke.log 2>&1 || { tail -20 gpurun_out/rowwise_smoke.log; exit 1; }
tail -2 gpurun_out/rowwise_smoke.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/rowwise_bench.json 2> gpurun_out/rowwise_bench.err || { tail -20 gpurun_out/rowwise_bench.err; exit 1; }
cat gpurun_out/rowwise_bench.json
