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
            print(f"{k:90s} {c:12s} launches {n:6d} avg_per_launch {tot / n:16.1f}")
