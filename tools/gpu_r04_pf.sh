#!/bin/bash
# Record prefetch (touch loads a round ahead) against the same kernel without it: same process, same arrays.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04pf}; LIBS=${2:-}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
run() {  # config fields
  timeout -k 10 400 python3 tools/exp_rowwise.py --config $1 --fields $2 --codes 0 --rounds 7 ${LIBS:+--libs $LIBS} > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.log || { tail -20 gpurun_out/${T}_$1.log; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_$1.json'))
for r in d['runs']: print('$1', r['fields'], r['kernel'], r['ms'], r.get('nan_pattern_same'), r.get('same_bits_as_first_row_variant'), r.get('max_rel_diff_to_tile'))"
}
run C2 ${C2F:-1,2,3,4}
run METRIC ${MF:-1,2,3,4,8}
