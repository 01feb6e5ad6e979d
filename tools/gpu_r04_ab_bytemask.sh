#!/bin/bash
# Byte-mask window entries (3-4 fields) against the select + legacy-multiply form: same process, same arrays.
# needs: python tools/build_experiments.py --tag nobm -DRG_ROWWISE_BYTEMASK=5   (run in the build container first)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04bm}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_columns.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/exp_rowwise.py --config C2 --fields 2,3,4 --codes 0 --rounds 9 --libs nobm=tools/_exp/libradargrid_hip_exp_nobm.so > gpurun_out/${T}_c2.json 2> gpurun_out/${T}_c2.log || { tail -20 gpurun_out/${T}_c2.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_c2.json'))
for r in d['runs']: print(r['fields'], r['kernel'], r['ms'], r['bit_identical'], r['same_bits_as_first_row_variant'], r['max_rel_diff_to_tile'])"
timeout -k 10 400 python3 tools/exp_rowwise.py --config METRIC --fields 3,4 --codes 0 --rounds 7 --libs nobm=tools/_exp/libradargrid_hip_exp_nobm.so > gpurun_out/${T}_metric.json 2> gpurun_out/${T}_metric.log || { tail -20 gpurun_out/${T}_metric.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_metric.json'))
for r in d['runs']: print(r['fields'], r['kernel'], r['ms'], r['bit_identical'], r['same_bits_as_first_row_variant'], r['max_rel_diff_to_tile'])"
