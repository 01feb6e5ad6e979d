#!/bin/bash
# K2 (rg_roi_grid_f32) and the builder after a change of the block kernel: their tests, then the K2 bench lines.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04k2}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --mode fused --no-cpu-baseline --no-c5-extra --steps 5 > gpurun_out/${T}_k2.json 2> gpurun_out/${T}_k2.log || { tail -5 gpurun_out/${T}_k2.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_k2.json')); r=d['roofline']
print('K2 bench grid', 'ms/step', d['ms_per_step'], r['kernel'], 'kernel ms', r.get('kernel_ms_median'), d['extras'].get('geometry_build_s'))"
grep -i "geometry\|build" gpurun_out/${T}_k2.log | tail -3
if [ "${FULL:-0}" = "1" ]; then
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c4.py -m gpu -x -q -k "fused or c4 or structure or oracle_rows" > gpurun_out/${T}_tests2.log 2>&1
rc=$?; tail -3 gpurun_out/${T}_tests2.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --config C4 --mode fused --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${T}_c4k2.json 2> gpurun_out/${T}_c4k2.log || { tail -5 gpurun_out/${T}_c4k2.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_c4k2.json')); r=d['roofline']
print('K2 config 4', 'ms/step', d['ms_per_step'], 'kernel ms', r.get('kernel_ms_median'))"
timeout -k 10 300 python3 tools/time_builder.py > gpurun_out/${T}_builder.log 2>&1; tail -5 gpurun_out/${T}_builder.log
fi
