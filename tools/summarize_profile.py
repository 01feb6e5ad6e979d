#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>) into small, committed files:

    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels + top others)
    profiles/<tag>_pmc.csv            per kernel / counter: launches, mean value per launch
    profiles/pmc_traffic.json         HBM bytes per launch of the dominant kernel, keyed by bench workload

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB and come
from separate passes; on gfx950 FETCH_SIZE counts half of the bytes of a coalesced streaming read (128-B
requests tallied as 64 B), so the read side is doubled; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:90]


def main():
    tag = sys.argv[1]
    src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if len(stats) > 1:      # gpurun MERGES into gpurun_out/: a second run of the same tag leaves the first run's files behind
        sys.exit(f"{src} holds the output of {len(stats)} profiler runs: delete it locally and profile again")
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:14]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    agg = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(path)):
                agg.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Launches", "MeanPerLaunch"])
        for (k, c), v in agg.items():
            w.writerow([k, c, len(v), f"{sum(v) / len(v):.6g}"])
    bench = os.path.join(src, "trace_bench.json")
    key = None
    if os.path.exists(bench):
        line = open(bench).read().strip().splitlines()[-1]
        info = json.loads(line)
        key = info["config"]["key"]
        kern = info["roofline"]["kernel"].replace("_dyn_kernel", "").replace("_kernel", "")
        fetch = [v for (k, c), v in agg.items() if c == "FETCH_SIZE" and kern in k]
        write = [v for (k, c), v in agg.items() if c == "WRITE_SIZE" and kern in k]
        if fetch and write:
            f_kib = sum(fetch[0]) / len(fetch[0])
            w_kib = sum(write[0]) / len(write[0])
            table_path = os.path.join(dst, "pmc_traffic.json")
            table = json.load(open(table_path)) if os.path.exists(table_path) else {}
            table[key] = {
                "source": f"profiles/{tag}_pmc.csv", "tag": tag,
                "kernel": info["roofline"]["kernel"],
                "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-B requests as 64 B), write = WRITE_SIZE x 1024",
                "hbm_bytes_per_launch": int(2 * f_kib * 1024 + w_kib * 1024),
                "algorithmic_bytes_per_launch": info["roofline"]["algorithmic_bytes_per_launch"],
            }
            json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
            print(json.dumps(table[key], indent=1))
    for (k, c), v in agg.items():
        print(f"{c:24s} {sum(v) / len(v):14.6g}  x{len(v):<3d} {k}")


if __name__ == "__main__":
    main()
