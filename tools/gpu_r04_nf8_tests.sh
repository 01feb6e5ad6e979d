#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04nf8t}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_columns.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/${T}_tests1.log 2>&1
rc=$?; tail -5 gpurun_out/${T}_tests1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "eight or passes_use or column_mode or c3_fused" > gpurun_out/${T}_tests2.log 2>&1
rc=$?; tail -5 gpurun_out/${T}_tests2.log; [ $rc -eq 0 ] || exit $rc
for P in 4 0; do
timeout -k 10 300 python3 bench.py --config C5 --c5-per-pass $P --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5_p$P.json 2> gpurun_out/${T}_c5_p$P.log || { tail -5 gpurun_out/${T}_c5_p$P.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_c5_p$P.json')); r=d['roofline']
print('C5 per-pass $P', 'ms/step', d['ms_per_step'], 'value', d['value'], r['kernel'], 'kernel ms', r.get('kernel_ms_median'), 'frac', r['frac'], d['config'].get('fields_per_pass'))"
done
