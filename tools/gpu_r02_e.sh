#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/tune_compact.py --config METRIC --fields 1,2,3,4,8 --tiles 0,128 > gpurun_out/r02e_tune_metric.json 2> gpurun_out/r02e_tune_metric.log || tail -5 gpurun_out/r02e_tune_metric.log
timeout -k 10 500 python3 bench.py --config C4 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02e_c4.json 2> gpurun_out/r02e_c4.log || tail -5 gpurun_out/r02e_c4.log
timeout -k 10 500 python3 bench.py --config C4 --no-cpu-baseline --no-compact --steps 5 --warmup 1 > gpurun_out/r02e_c4_k1.json 2> gpurun_out/r02e_c4_k1.log || tail -5 gpurun_out/r02e_c4_k1.log
python3 - <<'PY'
import json
for f in ("r02e_c4","r02e_c4_k1"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"], d["roofline"].get("ceiling_measured"), d["config"]["pairs"])
    except Exception as e: print(f, "fail", e)
d = json.load(open("gpurun_out/r02e_tune_metric.json"))
print({k: d[k] for k in d if k != "runs"})
for r in d["runs"]:
    print("   ", {k: r[k] for k in r if k not in ("bytes",)})
PY
tail -3 gpurun_out/r02e_c4.log
