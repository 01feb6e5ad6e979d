#!/bin/bash
# Round 3 A/B, one process per configuration, same arrays: the row-wise kernel as shipped against builds of the same
# sources with another number of running sums per lane (-DRG_ROWWISE_SLOTS=1: round 2's single chain; =2) and with a
# register cap (-DRG_ROWWISE_MIN_BLOCKS), loaded next to the in-tree library through tools/exp_rowwise.py --libs.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r03b}
python3 -m radar_processor_amd.build > gpurun_out/${TAG}_build.log 2>&1 || exit 1
C=radar_processor_amd/csrc
build_variant() {   # name, flags...
  local name=$1; shift
  hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -Iinclude -I$C -ffp-contract=off -DRG_EXPERIMENTS "$@" -c $C/rg_csr_compact.hip -o /tmp/${TAG}_$name.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC $C/rg_core.o $C/rg_csr_apply.o /tmp/${TAG}_$name.o $C/rg_products.o $C/rg_geometry.o $C/rg_roi_grid.o $C/rg_raster.o -o /tmp/${TAG}_lib$name.so
}
FIELDS=1,2,3,4
if [ "${2:-slots}" = "cfg3" ]; then         # three fields: records per lane and step / per lane and row / home of the row sums
  FIELDS=3
  build_variant k2t4 -DRG_ROWWISE_KPRE3=2 -DRG_ROWWISE_TARGET3=4 & build_variant k3t6 -DRG_ROWWISE_KPRE3=3 -DRG_ROWWISE_TARGET3=6 &
  build_variant k3t4 -DRG_ROWWISE_KPRE3=3 -DRG_ROWWISE_TARGET3=4 & build_variant k2t8 -DRG_ROWWISE_KPRE3=2 -DRG_ROWWISE_TARGET3=8 &
  build_variant k2t6lds -DRG_ROWWISE_REGS3=false & build_variant k3t8 -DRG_ROWWISE_KPRE3=3 -DRG_ROWWISE_TARGET3=8 &
  wait
  LIBS="k2t4=/tmp/${TAG}_libk2t4.so,k3t6=/tmp/${TAG}_libk3t6.so,k3t4=/tmp/${TAG}_libk3t4.so,k2t8=/tmp/${TAG}_libk2t8.so,k2t6lds=/tmp/${TAG}_libk2t6lds.so,k3t8=/tmp/${TAG}_libk3t8.so"
elif [ "${2:-slots}" = "occ3" ]; then      # three fields: one chain and two records per step buy a wavefront of occupancy
  FIELDS=3
  build_variant k2c1 -DRG_ROWWISE_KPRE3=2 -DRG_ROWWISE_SLOTS=1 & build_variant k3c1 -DRG_ROWWISE_KPRE3=3 -DRG_ROWWISE_SLOTS=1 &
  build_variant k2c1lds -DRG_ROWWISE_KPRE3=2 -DRG_ROWWISE_SLOTS=1 -DRG_ROWWISE_REGS3=false & build_variant k2c2 -DRG_ROWWISE_KPRE3=2 &
  wait
  LIBS="k2c1=/tmp/${TAG}_libk2c1.so,k3c1=/tmp/${TAG}_libk3c1.so,k2c1lds=/tmp/${TAG}_libk2c1lds.so,k2c2=/tmp/${TAG}_libk2c2.so"
elif [ "${2:-slots}" = "fill" ]; then       # entries per thread and batch of the window fill
  build_variant fill2 -DRG_FILL_BATCH=2 & build_variant fill8 -DRG_FILL_BATCH=8 & build_variant fill1 -DRG_FILL_BATCH=1 &
  wait
  LIBS="fill1=/tmp/${TAG}_libfill1.so,fill2=/tmp/${TAG}_libfill2.so,fill8=/tmp/${TAG}_libfill8.so"
elif [ "${2:-slots}" = "occ" ]; then        # register caps: 7 wavefronts per SIMD for one field, 6 for three
  FIELDS=1,3
  build_variant w7 -DRG_ROWWISE_WAVES1=7 -DRG_ROWWISE_WAVES3=6 &
  wait
  LIBS="w7w6=/tmp/${TAG}_libw7.so"
elif [ "${2:-slots}" = "selects" ]; then      # the one-select + v_mul_legacy_f32 form of the masked product against round 2's two selects
  build_variant twoselects -DRG_ROWWISE_TWO_SELECTS &
  wait
  LIBS="twoselects=/tmp/${TAG}_libtwoselects.so"
else
build_variant slots1 -DRG_ROWWISE_SLOTS=1 & build_variant slots2 -DRG_ROWWISE_SLOTS=2 & build_variant slots3cap -DRG_ROWWISE_SLOTS=3 -DRG_ROWWISE_WAVES1=6 -DRG_ROWWISE_WAVES3=5 & build_variant slots3 -DRG_ROWWISE_SLOTS=3 &
wait
LIBS="slots1=/tmp/${TAG}_libslots1.so,slots2=/tmp/${TAG}_libslots2.so,slots3=/tmp/${TAG}_libslots3.so,slots3cap=/tmp/${TAG}_libslots3cap.so"
fi
for cfg in C2 METRIC; do
  timeout -k 10 400 python3 tools/exp_rowwise.py --config $cfg --fields $FIELDS --codes 0 --rounds 15 --libs $LIBS > gpurun_out/${TAG}_slots_${cfg}.json 2> gpurun_out/${TAG}_slots_${cfg}.log || exit 1
done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for cfg in ("C2", "METRIC"):
    d = json.load(open(f"gpurun_out/{tag}_slots_{cfg}.json"))
    for nf in sorted({r["fields"] for r in d["runs"]}):
        print(cfg, f"F{nf}", {r["kernel"]: r["ms"] for r in d["runs"] if r["fields"] == nf},
              "same bits as in-tree:", [r["same_bits_as_first_row_variant"] for r in d["runs"] if r["fields"] == nf and "@" in r["kernel"]])
PY
if [ "${3:-}" = "tests" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
  rc=$?; tail -4 gpurun_out/${TAG}_tests.log; exit $rc
fi
