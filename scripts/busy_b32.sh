#!/bin/bash
# GPU-busy time per B=32 training step: rocprofv3 kernel trace of scripts/profile_host_train.py (45 steps in all)
set -e
cd "$(dirname "$0")/.."
export PYTHONPATH=. TMPDIR=/tmp
rm -rf gpurun_out/busy32
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/busy32 -o b32 -- python3 scripts/profile_host_train.py 32 20 > gpurun_out/busy32.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/busy32/**/b32_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"45 steps: {calls} launches = {calls/45:.0f} per step, busy {tot/45e6:.3f} ms per step, mean kernel {tot/calls/1e3:.2f} us")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{float(r["TotalDurationNs"])/45e3:8.1f} us/step {int(r["Calls"])/45:6.1f} calls/step  {r["Name"][:90]}')
PY
grep "enqueue" gpurun_out/busy32.log
