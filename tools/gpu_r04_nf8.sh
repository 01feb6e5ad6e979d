#!/bin/bash
# Eight fields in one row-wise pass against two passes of four: same process, same arrays.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04nf8}; shift || true
LIBS=${1:-}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
for CFG in ${CFGS:-METRIC}; do
timeout -k 10 400 python3 tools/exp_rowwise.py --config $CFG --fields 4,8 --codes 0 --rounds 7 ${LIBS:+--libs $LIBS} > gpurun_out/${T}_$CFG.json 2> gpurun_out/${T}_$CFG.log || { tail -20 gpurun_out/${T}_$CFG.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${T}_$CFG.json'))
for r in d['runs']: print('$CFG', r['fields'], r['kernel'], r['ms'], r.get('nan_pattern_same'), r.get('same_bits_as_first_row_variant'), r.get('max_rel_diff_to_tile'))"
done
