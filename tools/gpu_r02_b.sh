#!/bin/bash
# Round-2 GPU call B: compact v2 (patch chunks, multi-field): correctness tests first, then tuning runs
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -m radar_processor_amd.build > gpurun_out/r02b_build.log 2>&1 || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02b_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r02b_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/tune_compact.py --config C2 --fields 1,2,3,4,8 --tiles 0 > gpurun_out/r02b_tune_c2.json 2> gpurun_out/r02b_tune_c2.log || tail -5 gpurun_out/r02b_tune_c2.log
timeout -k 10 300 python3 tools/tune_compact.py --config C2 --fields 3 --tiles 128,384 --windows 1024,2048 > gpurun_out/r02b_tune_c2_t.json 2> gpurun_out/r02b_tune_c2_t.log || tail -5 gpurun_out/r02b_tune_c2_t.log
timeout -k 10 400 python3 tools/tune_compact.py --config METRIC --fields 1,3 --tiles 0 --windows 512,1024 > gpurun_out/r02b_tune_metric.json 2> gpurun_out/r02b_tune_metric.log || tail -5 gpurun_out/r02b_tune_metric.log
python3 - <<'PY'
import json
for f in ("r02b_tune_c2", "r02b_tune_c2_t", "r02b_tune_metric"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k: d[k] for k in d if k != "runs"})
    for r in d["runs"]:
        print("   ", r)
PY
