#!/bin/bash
# Quick check after a kernel change: the bit-for-bit tests of the row-wise kernel + its times (3 / 4 / 8 fields).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04q}; LIBS=${2:-}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_columns.py -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/gpu_r04_pf.sh $T $LIBS
