#!/usr/bin/env python3
"""Condense a tools/gpu_r04_clock.sh output directory (gpurun_out/prof_<tag>) into profiles/<name>.txt: per row-wise kernel the
median duration (kernel-trace pass), the mean counter values of the PMC passes and GRBM_GUI_ACTIVE / duration / 8 XCDs = the
clock the kernel ran at."""
import collections
import csv
import glob
import sys


def main():
    out, dst = sys.argv[1], sys.argv[2]
    dur = collections.defaultdict(list)
    for p in glob.glob(out + "/trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(p)):
            if "rowwise" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("rowwise_kernel")[1][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    agg = collections.defaultdict(list)
    for p in glob.glob(out + "/pmc*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            agg[(r["Kernel_Name"].split("rowwise_kernel")[1][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
    lines = []
    for k in sorted(dur):
        d = sorted(dur[k])[len(dur[k]) // 2]
        lines.append(f"csr_compact_rowwise_kernel{k}  median {d:.3f} ms over {len(dur[k])} launches")
        for (kk, c), v in sorted(agg.items()):
            if kk == k:
                m = sum(v) / len(v)
                extra = f"   -> {m / d / 1e6 / 8:.3f} GHz per XCD" if c == "GRBM_GUI_ACTIVE" else ""
                lines.append(f"    {c:24s} {m:.4g}{extra}")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
