#!/bin/bash
# What the row-wise kernel's clock does under 1 / 4 / 8 fields: GRBM_GUI_ACTIVE (gfx clock cycles while busy) against the
# kernel's duration from a separate trace pass, plus the SQ busy / VALU / wait counters.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r04clk}
OUT=gpurun_out/prof_$T
rm -rf $OUT; mkdir -p $OUT
python3 -c "from radar_processor_amd import build; assert not build.is_stale(), 'stale library'" || exit 1
CMD="tools/exp_rowwise.py --config METRIC --fields ${F:-1,4,8} --codes 0 --rounds 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $CMD > $OUT/trace.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
echo trace done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --kernel-include-regex "rowwise" --output-format csv -d $OUT/pmc1 -- python3 $CMD > $OUT/pmc1.json 2> $OUT/pmc1.log || { tail -5 $OUT/pmc1.log; exit 1; }
echo pmc1 done
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-include-regex "rowwise" --output-format csv -d $OUT/pmc2 -- python3 $CMD > $OUT/pmc2.json 2> $OUT/pmc2.log || { tail -5 $OUT/pmc2.log; echo "pmc2 failed"; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(list)
for p in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "rowwise" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("rowwise_kernel")[1][:24]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
agg = collections.defaultdict(list)
for p in glob.glob(out + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        agg[(r["Kernel_Name"].split("rowwise_kernel")[1][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(dur):
    d = sorted(dur[k])[len(dur[k]) // 2]
    print(k, "median ms", round(d, 3), "launches", len(dur[k]))
    for (kk, c), v in sorted(agg.items()):
        if kk == k:
            m = sum(v) / len(v)
            print("   ", c, f"{m:.4g}", ("-> GHz %.3f" % (m / d / 1e6)) if c == "GRBM_GUI_ACTIVE" else "")
PY
