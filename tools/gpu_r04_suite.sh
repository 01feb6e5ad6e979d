#!/bin/bash
# Full GPU suite + the bench lines that changed this session (config 3, config 5 both pass sizes, default).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04s}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
line() {  # tag, args...
  local tag=$1; shift
  timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${T}_$tag.json 2> gpurun_out/${T}_$tag.log || { tail -5 gpurun_out/${T}_$tag.log; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_$tag.json')); r=d['roofline']; e=d.get('extras',{})
print('$tag', d['config']['key'], 'ms/step', d['ms_per_step'], 'value', d['value'], 'kernel ms med', r.get('kernel_ms_median'), 'frac', r['frac'], 'per_pass', d['config'].get('fields_per_pass'), json.dumps(e.get('c5'))[:300] if 'c5' in e else '')"
}
line default
line c3 --config C2 --fields 3 --no-c5-extra
line m3 --fields 3 --no-c5-extra --steps 10
line c5 --config C5 --steps 5 --warmup 1
line c5p4 --config C5 --c5-per-pass 4 --steps 5 --warmup 1
line c5fused --config C5 --products fused --steps 5 --warmup 1
