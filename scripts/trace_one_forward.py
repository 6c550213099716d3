"""Developer aid: per-launch durations of the last forward in a rocprofv3 --kernel-trace CSV (argv[1]), from the
first kernel whose name contains argv[2] to its next occurrence."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
agg = {}
for r in rows[s:e]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[n[:40]] = agg.get(n[:40], 0) + d
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {d:7.1f} us  grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']:>4s} {n}")
print("---- totals")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{v:8.1f} us  {k}")
print(f"{(int(rows[e]['Start_Timestamp'])-t0)/1e3:8.1f} us  wall, first launch to the next forward's first launch")
