#!/bin/bash
# Round 4: correctness of the column mode (tests) and its timing against the row-wise kernel; in the experiment build also
# the two prefetching variants (walk, loader).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04d}
timeout -k 10 600 python3 -m pytest tests/test_gpu_columns.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -5 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 tools/exp_columns.py --config C2 --fields 1,3,4 --pieces 0,1,4 --exp-lib cols > gpurun_out/${T}_cols_c2.json 2> gpurun_out/${T}_cols_c2.log || { tail -5 gpurun_out/${T}_cols_c2.log; exit 1; }
timeout -k 10 500 python3 tools/exp_columns.py --config METRIC --fields 1,3,4 --pieces 0 --rounds 7 --exp-lib cols > gpurun_out/${T}_cols_metric.json 2> gpurun_out/${T}_cols_metric.log || { tail -5 gpurun_out/${T}_cols_metric.log; exit 1; }
python3 - "$T" <<'PY'
import json, sys
for c in ("c2", "metric"):
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_cols_{c}.json"))
    for r in d["runs"]:
        print(c, r["fields"], r["kernel"], r["ms"], r["min_ms"], r["frac_of_8TBps"], r["same_bits"])
PY
