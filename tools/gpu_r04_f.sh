#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04f}
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -5 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 bench.py > gpurun_out/${T}_default.json 2> gpurun_out/${T}_default.log || { tail -20 gpurun_out/${T}_default.log; exit 1; }
timeout -k 10 300 python3 bench.py --layout csr --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_csr.json 2> gpurun_out/${T}_csr.log || exit 1
timeout -k 10 300 python3 bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5.json 2> gpurun_out/${T}_c5.log || { tail -20 gpurun_out/${T}_c5.log; exit 1; }
timeout -k 10 300 python3 bench.py --config C2 --fields 3 --no-cpu-baseline > gpurun_out/${T}_c3.json 2> gpurun_out/${T}_c3.log || exit 1
python3 - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for f in ("default", "csr", "c5", "c3"):
    d = json.load(open(f"gpurun_out/{t}_{f}.json")); r = d["roofline"]; e = d.get("extras", {})
    print(f, d["config"]["key"], "value", d["value"], "ms/step", d["ms_per_step"], r["kernel"], r["kernel_ms_median"], "frac", r["frac"], e.get("geometry_build_s"), e.get("geometry_resident_gb"), e.get("geometry_layout"))
    print("   checked:", d["config"]["checked"])
PY
