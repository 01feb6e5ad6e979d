#!/bin/bash
# same-process A/B of the in-tree library against a build whose rg_csr_compact.hip is the copy under tools/ab_old/
# (a previous revision, put there by hand for the comparison and not committed)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r03n}
python3 -m radar_processor_amd.build > gpurun_out/${TAG}_build.log 2>&1 || exit 1
C=radar_processor_amd/csrc
hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -Iinclude -I$C -ffp-contract=off -c tools/ab_old/rg_csr_compact.hip -o /tmp/${TAG}_old.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC $C/rg_core.o $C/rg_csr_apply.o /tmp/${TAG}_old.o $C/rg_products.o $C/rg_geometry.o $C/rg_roi_grid.o $C/rg_raster.o -o /tmp/${TAG}_libold.so || exit 1
for cfg in C2 METRIC; do
  timeout -k 10 400 python3 tools/exp_rowwise.py --config $cfg --fields 1,2,3,4 --codes 0 --rounds 15 --libs old=/tmp/${TAG}_libold.so > gpurun_out/${TAG}_ab_${cfg}.json 2> gpurun_out/${TAG}_ab_${cfg}.log || exit 1
done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for cfg in ("C2", "METRIC"):
    d = json.load(open(f"gpurun_out/{tag}_ab_{cfg}.json"))
    for nf in (1, 2, 3, 4):
        print(cfg, f"F{nf}", {r["kernel"]: r["ms"] for r in d["runs"] if r["fields"] == nf},
              "same bits:", [r["same_bits_as_first_row_variant"] for r in d["runs"] if r["fields"] == nf and "@" in r["kernel"]])
PY
