#!/bin/bash
# LDS / MFMA counters of the direct 3x3 weight-gradient kernel: gpurun -- 'bash scripts/pmc_wgrad3x3.sh' -> gpurun_out/pmc_w3.txt
set -e -o pipefail
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(pwd)"
OUT="$R/gpurun_out/pmc_w3"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH="$R"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT" -o w3 -- python3 "$R/scripts/bench_wgrad3x3.py" 256 > "$OUT/log.txt" 2>&1
python3 - "$OUT" > "$R/gpurun_out/pmc_w3.txt" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:50] + " grid " + r["Grid_Size"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_BUSY_CYCLES": cnt[k] += 1
for k, v in agg.items():
    n = max(1, cnt[k])
    print(k, "launches", n, {c: round(x / n) for c, x in v.items()})
PY
echo done
