#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02f_tests.log 2>&1
rc=$?; tail -6 gpurun_out/r02f_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 bench.py --config C4 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02f_c4.json 2> gpurun_out/r02f_c4.log || tail -5 gpurun_out/r02f_c4.log
timeout -k 10 300 python3 bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02f_c5.json 2> gpurun_out/r02f_c5.log || tail -5 gpurun_out/r02f_c5.log
timeout -k 10 300 python3 bench.py --config C5 --mode fused --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02f_c5_fused.json 2> gpurun_out/r02f_c5_fused.log || tail -5 gpurun_out/r02f_c5_fused.log
timeout -k 10 300 python3 bench.py --config C2 --fields 3 --no-cpu-baseline > gpurun_out/r02f_c3.json 2> gpurun_out/r02f_c3.log
timeout -k 10 300 python3 bench.py --config C2 --no-cpu-baseline > gpurun_out/r02f_c2.json 2> gpurun_out/r02f_c2.log
python3 - <<'PY'
import json
for f in ("r02f_c4","r02f_c5","r02f_c5_fused","r02f_c3","r02f_c2"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("ceiling_measured"), d["config"]["pairs"], d.get("end_to_end"))
    except Exception as e: print(f, "fail", e)
PY
