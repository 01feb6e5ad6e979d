#!/bin/bash
for r in 0 1 2 3 5 8 13; do
  RG_ROT=$r python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/rot_c_$r.json 2>/dev/null
  RG_ROT=$r python3 bench.py --no-cpu-baseline --no-compact --steps 10 > gpurun_out/rot_s_$r.json 2>/dev/null
done
python3 - <<PY
import json
for r in (0,1,2,3,5,8,13):
    for k in "cs":
        try:
            d=json.load(open("gpurun_out/rot_%s_%d.json"%(k,r))); print(r, k, d["roofline"]["kernel"], d["roofline"]["kernel_ms"])
        except Exception as e: print(r,k,"fail",e)
PY
