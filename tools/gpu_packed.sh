#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02p_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r02p_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/tune_compact.py --config C2 --fields 1,2,3,4 --tiles 0 --rounds 7 > gpurun_out/r02p_c2.json 2> gpurun_out/r02p_c2.log
timeout -k 10 400 python3 tools/tune_compact.py --config METRIC --fields 1,2,3,4 --tiles 0 > gpurun_out/r02p_m.json 2> gpurun_out/r02p_m.log
python3 - <<'PY'
import json
for f in ("r02p_c2","r02p_m"):
    try: d=json.load(open("gpurun_out/%s.json"%f))
    except Exception as e: print(f,"fail",e); continue
    print(f)
    for r in d["runs"]: print("   ", {k:r[k] for k in r if k not in ("ref_format_TBps",)})
PY
