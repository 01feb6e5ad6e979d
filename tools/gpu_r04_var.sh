#!/bin/bash
# N consecutive default bench.py processes (the headline's distribution): median kernel ms, frac, the settle probes
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04v}; N=${2:-6}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
for i in $(seq 1 $N); do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_var_$i.json 2> gpurun_out/${T}_var_$i.log || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_var_$i.json')); r=d['roofline']
print($i, r['kernel_ms_median'], r['frac'], d['config'].get('records_settled'))"
done
