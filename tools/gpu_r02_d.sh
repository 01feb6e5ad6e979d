#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_edges.py -m gpu -x -q -k compact > gpurun_out/r02d_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02d_tests.log; [ $rc -eq 0 ] || exit $rc
python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/r02c_default.json 2> gpurun_out/r02c_default.log
python3 bench.py --no-cpu-baseline --no-compact --steps 10 > gpurun_out/r02c_nocompact.json 2> gpurun_out/r02c_nocompact.log
timeout -k 10 300 python3 tools/tune_compact.py --config C2 --fields 1,2,3,4 --tiles 0,256 > gpurun_out/r02c_tune_c2.json 2> gpurun_out/r02c_tune_c2.log || tail -5 gpurun_out/r02c_tune_c2.log
python3 - <<'PY'
import json
for f in ("r02c_default","r02c_nocompact"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"].get("ceiling_measured"))
for f in ("r02c_tune_c2",):
    d = json.load(open(f"gpurun_out/{f}.json"))
    for r in d["runs"]:
        print("   ", {k: r[k] for k in r if k not in ("bytes",)})
PY
