"""Aggregate rocprofv3 CSV output per kernel: kernel-trace stats and PMC counters (average per launch).
usage: pmc_summary.py <dir> [<dir> ...]  -> prints one table per *_kernel_stats.csv / *_counter_collection.csv found."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    return name if len(name) <= 90 else name[:87] + "..."


for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"== {os.path.relpath(f, d)}")
        with open(f) as fh:
            for row in csv.DictReader(fh):
                print(f'{short(row["Name"]):90s} calls {row["Calls"]:>6s} avg_ns {float(row["AverageNs"]):12.0f} '
                      f'total_ms {float(row["TotalDurationNs"]) / 1e6:10.3f} pct {row["Percentage"]:>6s}')
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        print(f"== {os.path.relpath(f, d)}")
        acc = defaultdict(lambda: [0.0, set()])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (short(row["Kernel_Name"]), row["Counter_Name"])
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1].add(row["Dispatch_Id"])
        for (k, c), (tot, disp) in sorted(acc.items()):
            n = max(1, len(disp))
            print(f"{k:90s} {c:26s} launches {n:6d} avg_per_launch {tot / n:16.1f}")
        # MFMA pipe occupancy where the pass holds both counters (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        kernels = sorted({k for (k, c) in acc if c == "SQ_VALU_MFMA_BUSY_CYCLES"})
        for k in kernels:
            if (k, "GRBM_GUI_ACTIVE") in acc and acc[(k, "GRBM_GUI_ACTIVE")][0] > 0:
                util = 8.0 * acc[(k, "SQ_VALU_MFMA_BUSY_CYCLES")][0] / (acc[(k, "GRBM_GUI_ACTIVE")][0] * 1024.0)
                print(f"{k:90s} mfma_pipe_busy_fraction {util:8.4f}")
