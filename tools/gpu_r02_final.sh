#!/bin/bash
# final pass of round 2, part A: full GPU suite and every bench configuration (part B: tools/gpu_r02_profiles.sh)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -m radar_processor_amd.build > gpurun_out/r02z_build.log 2>&1 || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02z_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r02z_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py > gpurun_out/r02z_default.json 2> gpurun_out/r02z_default.log
timeout -k 10 300 python3 bench.py --tile-kernel --no-cpu-baseline > gpurun_out/r02z_tile.json 2> gpurun_out/r02z_tile.log
timeout -k 10 300 python3 bench.py --no-compact --no-cpu-baseline > gpurun_out/r02z_k1.json 2> gpurun_out/r02z_k1.log
timeout -k 10 300 python3 bench.py --config C2 --no-cpu-baseline > gpurun_out/r02z_c2.json 2> gpurun_out/r02z_c2.log
timeout -k 10 300 python3 bench.py --config C2 --fields 3 --no-cpu-baseline > gpurun_out/r02z_c3.json 2> gpurun_out/r02z_c3.log
timeout -k 10 300 python3 bench.py --config C2 --fields 3 --tile-kernel --no-cpu-baseline > gpurun_out/r02z_c3tile.json 2> gpurun_out/r02z_c3tile.log
timeout -k 10 300 python3 bench.py --fields 3 --no-cpu-baseline --steps 10 > gpurun_out/r02z_m3.json 2> gpurun_out/r02z_m3.log
timeout -k 10 400 python3 bench.py --config C4 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02z_c4.json 2> gpurun_out/r02z_c4.log
timeout -k 10 300 python3 bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02z_c5.json 2> gpurun_out/r02z_c5.log
timeout -k 10 300 python3 bench.py --config C5 --mode fused --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r02z_c5f.json 2> gpurun_out/r02z_c5f.log
timeout -k 10 300 python3 bench.py --mode fused --no-cpu-baseline --steps 5 > gpurun_out/r02z_k2.json 2> gpurun_out/r02z_k2.log
python3 - <<'PY'
import json
for f in ("default","tile","k1","c2","c3","c3tile","m3","c4","c5","c5f","k2"):
    try:
        d=json.load(open("gpurun_out/r02z_%s.json"%f)); r=d["roofline"]
        print(f, d["config"]["key"], "value", d["value"], "ms/step", d["ms_per_step"], r["kernel"], "kernel_ms", r["kernel_ms"], "GB/s", r["achieved"], "frac", r["frac"], "ceil", r.get("ceiling_measured"), "frac_ceil", r.get("frac_of_ceiling"), "refGBps", r["reference_format_GBps"], "ref_frac", r.get("reference_format_frac"), d.get("end_to_end",{}).get("ms_per_step"))
    except Exception as e: print(f, "fail", e)
PY
