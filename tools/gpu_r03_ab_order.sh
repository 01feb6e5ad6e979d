#!/bin/bash
# Round 3, A/B of the record order on one box: full GPU suite, then bench.py processes alternating between the
# dispatch-order records and round 2's line-major segments (same kernel, same contents, another order in memory).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r03a}
N=${2:-4}
python3 -m radar_processor_amd.build > gpurun_out/${TAG}_build.log 2>&1 || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log; [ $rc -eq 0 ] || exit $rc
for i in $(seq 1 $N); do
  for order in dispatch segment; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --rec-order $order > gpurun_out/${TAG}_${order}_$i.json 2> gpurun_out/${TAG}_${order}_$i.log || exit 1
  done
done
python3 - "$TAG" "$N" <<'PY'
import json, sys
tag, n = sys.argv[1], int(sys.argv[2])
for order in ("dispatch", "segment"):
    for i in range(1, n + 1):
        d = json.load(open(f"gpurun_out/{tag}_{order}_{i}.json")); r = d["roofline"]
        print(order, i, "ms/step", d["ms_per_step"], "kernel min/med/mean/max", r["kernel_ms_min"], r["kernel_ms_median"],
              r["kernel_ms"], r["kernel_ms_max"], "frac", r["frac"], "ceil", r.get("ceiling_measured"))
PY
