#!/bin/bash
# per-dispatch durations of the conv kernels inside one bench step (rocprofv3 kernel trace)
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(pwd)"
OUT="$R/gpurun_out/trace_conv"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$R/bench.py" --no-cpu-baseline --no-train --no-extra-legs --steps 4 --warmup 2 > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last step: find the last stem entry kernel
last = max(i for i, n in enumerate(names) if "conv1_pool" in n)
prev_end = None
for r in rows[last:last + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r['Grid_Size']:>8}  {n}")
PY
