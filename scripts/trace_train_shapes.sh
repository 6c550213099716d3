#!/bin/bash
# Per-dispatch kernel trace of a few hybrid training steps, aggregated by (kernel, grid): which launch shapes cost what.
# gpurun -- 'bash scripts/trace_train_shapes.sh [train-batch]'  -> gpurun_out/train_shapes.txt
set -e -o pipefail
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(pwd)"
TB="${1:-256}"
OUT="$R/gpurun_out/trace_train"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tr -- \
  python3 "$R/bench.py" --no-cpu-baseline --no-extra-legs --steps 1 --warmup 1 --train-steps 6 --train-batch "$TB" > "$OUT/log.txt" 2>&1
python3 - "$OUT" > "$R/gpurun_out/train_shapes.txt" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    for pre in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        if name.startswith(pre): name = name[len(pre):]
    name = name.split("(")[0][:70]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"total kernel time {tot / 1e3:.2f} ms over the run (9 training steps incl. warm-up + 2 inference steps + profiler pass)")
for (n, g, w), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{n:72s} wgs {g:7d} x{w:4d}  calls {c:5d}  avg {us / c:8.1f} us  total {us / 1e3:8.2f} ms")
PY
echo done
