#!/bin/bash
# rocprofv3 passes over the default bench workload (run on the GPU box from the repo root).
# Kernel trace + stats first, then PMC counters in separate passes (never combined with tracing).
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"
mkdir -p "$OUT"
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-c5-extra $*"
python3 -m radar_processor_amd.build > "$OUT/build.log" 2>&1 || exit 1   # nothing may compile (= exec hipcc) under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace_bench.json" 2> "$OUT/trace.log" || exit 1
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "csr_apply|csr_compact|roi_block" --output-format csv -d "$OUT/pmc_$C" -- python3 bench.py $ARGS > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.log" || exit 1
  echo "$C done"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex "csr_apply|csr_compact|roi_block" --output-format csv -d "$OUT/pmc_sq1" -- python3 bench.py $ARGS > "$OUT/pmc_sq1_bench.json" 2> "$OUT/pmc_sq1.log" || exit 1
echo "sq1 done"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-include-regex "csr_apply|csr_compact|roi_block" --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py $ARGS > "$OUT/pmc_sq2_bench.json" 2> "$OUT/pmc_sq2.log" || exit 1
echo "sq2 done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --kernel-include-regex "csr_apply|csr_compact|roi_block" --output-format csv -d "$OUT/pmc_tc" -- python3 bench.py $ARGS > "$OUT/pmc_tc_bench.json" 2> "$OUT/pmc_tc.log" || echo "tc pass failed (counter names?)"
echo "all done"
find "$OUT" -name "*.csv" | head -50
