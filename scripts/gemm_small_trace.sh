#!/bin/bash
# GPU-side duration of the encoder's Linear products at a small token count, per tile configuration (kernel trace)
set -e
cd "$(dirname "$0")/.."
export PYTHONPATH=. TMPDIR=/tmp T=${1:-1600}
rm -rf gpurun_out/gs
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gs -o gs -- python3 scripts/bench_gemm_small.py > gpurun_out/gs.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gs/**/gs_kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{float(r["AverageNs"])/1e3:7.2f} us avg  {int(r["Calls"]):5d} calls  min {float(r["MinNs"])/1e3:6.2f}  {r["Name"][:100]}')
PY
