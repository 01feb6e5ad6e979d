#!/bin/bash
# Round 4 evidence, part A: full GPU suite, every bench configuration, and alternating default bench.py processes (the
# run-to-run spread of the headline kernel).  Part B (rocprofv3): tools/gpu_r04_profiles.sh
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04z}
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${T}_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 bench.py > gpurun_out/${T}_default.json 2> gpurun_out/${T}_default.log || { tail -20 gpurun_out/${T}_default.log; exit 1; }
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_var_$i.json 2> gpurun_out/${T}_var_$i.log || exit 1
done
timeout -k 10 300 python3 bench.py --layout csr --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_csrlayout.json 2> gpurun_out/${T}_csrlayout.log
timeout -k 10 300 python3 bench.py --tile-kernel --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_tile.json 2> gpurun_out/${T}_tile.log
timeout -k 10 300 python3 bench.py --layout csr --no-compact --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_k1.json 2> gpurun_out/${T}_k1.log
timeout -k 10 300 python3 bench.py --config C2 --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_c2.json 2> gpurun_out/${T}_c2.log
timeout -k 10 300 python3 bench.py --config C2 --fields 3 --no-cpu-baseline --no-c5-extra > gpurun_out/${T}_c3.json 2> gpurun_out/${T}_c3.log
timeout -k 10 300 python3 bench.py --fields 3 --no-cpu-baseline --no-c5-extra --steps 10 > gpurun_out/${T}_m3.json 2> gpurun_out/${T}_m3.log
timeout -k 10 400 python3 bench.py --config C4 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c4.json 2> gpurun_out/${T}_c4.log
timeout -k 10 300 python3 bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5.json 2> gpurun_out/${T}_c5.log
timeout -k 10 300 python3 bench.py --config C5 --c5-per-pass 4 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5p4.json 2> gpurun_out/${T}_c5p4.log
timeout -k 10 300 python3 bench.py --config C5 --products fused --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5fused.json 2> gpurun_out/${T}_c5fused.log
timeout -k 10 300 python3 bench.py --config C5 --mode fused --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${T}_c5k2.json 2> gpurun_out/${T}_c5k2.log
timeout -k 10 300 python3 bench.py --mode fused --no-cpu-baseline --no-c5-extra --steps 5 > gpurun_out/${T}_k2.json 2> gpurun_out/${T}_k2.log
timeout -k 10 300 python3 bench.py --config C4 --mode fused --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${T}_c4k2.json 2> gpurun_out/${T}_c4k2.log
python3 - "$T" <<'PY'
import json, sys
t = sys.argv[1]
rows = {}
for f in ["default"] + [f"var_{i}" for i in range(1, 4)] + ["csrlayout", "tile", "k1", "c2", "c3", "m3", "c4", "c5", "c5p4", "c5fused", "c5k2", "k2", "c4k2"]:
    try:
        d = json.load(open(f"gpurun_out/{t}_{f}.json")); r = d["roofline"]; e = d.get("extras", {})
        rows[f] = d
        print(f, d["config"]["key"], "value", d["value"], "ms/step", d["ms_per_step"], r["kernel"], "kernel_ms min/med/mean/max",
              r.get("kernel_ms_min"), r.get("kernel_ms_median"), r["kernel_ms"], r.get("kernel_ms_max"), "frac", r["frac"],
              "ceil", r.get("ceiling_measured"), "frac_ceil", r.get("frac_of_ceiling"), "ref_frac", r.get("reference_format_frac"),
              d.get("end_to_end", {}).get("ms_per_step"), e.get("geometry_build_s"), e.get("geometry_resident_gb"), e.get("geometry_layout"))
        if "c5" in e: print("    extras.c5", json.dumps(e["c5"])[:400])
    except Exception as ex:
        print(f, "fail", ex)
json.dump(rows, open(f"gpurun_out/{t}_bench_lines.json", "w"), indent=1)
PY
