#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02c_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r02c_tests.log; [ $rc -eq 0 ] || exit $rc
python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/r02c_default.json 2> gpurun_out/r02c_default.log
python3 bench.py --no-cpu-baseline --no-compact --steps 10 > gpurun_out/r02c_nocompact.json 2> gpurun_out/r02c_nocompact.log
timeout -k 10 300 python3 tools/tune_compact.py --config C2 --fields 1,2,3,4,8 --tiles 0 > gpurun_out/r02c_tune_c2.json 2> gpurun_out/r02c_tune_c2.log || tail -5 gpurun_out/r02c_tune_c2.log
timeout -k 10 400 python3 tools/tune_compact.py --config METRIC --fields 1,3,8 --tiles 0 > gpurun_out/r02c_tune_metric.json 2> gpurun_out/r02c_tune_metric.log || tail -5 gpurun_out/r02c_tune_metric.log
python3 - <<'PY'
import json
for f in ("r02c_default","r02c_nocompact"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"].get("ceiling_measured"))
for f in ("r02c_tune_c2", "r02c_tune_metric"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k: d[k] for k in d if k != "runs"})
    for r in d["runs"]:
        print("   ", {k: r[k] for k in r if k not in ("bytes",)})
PY
