#!/bin/bash
# Regenerates the rocprofv3 material behind profiles/r03_*: run on the GPU box from the repository root
# (gpurun -- 'bash scripts/profile_r03.sh'); results land under gpurun_out/prof/, then
# `python scripts/pmc_summary.py gpurun_out/prof > profiles/r03_rocprof_summary.txt` and copy the kernel_stats CSVs.
set -e -o pipefail
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(pwd)"
OUT="$R/gpurun_out/prof"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra rocprof flags..., -- bench args
  local name="$1"; shift
  local flags=(); while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats "${flags[@]}" --output-format csv -d "$OUT/$name" -o "$name" -- \
    python3 "$R/bench.py" "$@" > "$OUT/$name.log" 2>&1
  echo "$name done"
}
run stats_infer -- --no-cpu-baseline --no-train --no-extra-legs --steps 10 --warmup 3
run stats_train -- --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 --train-steps 10
run stats_vitl  -- --workload vitl-infer-fp8 --steps 5 --warmup 2
run stats_vitb  -- --workload vitb-train --steps 5 --warmup 2
run pmc_fetch --pmc FETCH_SIZE -- --no-cpu-baseline --no-train --no-extra-legs --steps 3 --warmup 1
run pmc_write --pmc WRITE_SIZE -- --no-cpu-baseline --no-train --no-extra-legs --steps 3 --warmup 1
# MFMA pipe occupancy per kernel: 8 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs)  (GUI_ACTIVE sums the 8 XCDs)
run pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --no-cpu-baseline --no-train --no-extra-legs --steps 3 --warmup 1
run pmc_mfma_vitb --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --workload vitb-train --steps 2 --warmup 1
run pmc_mfma_vitl --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --workload vitl-infer-fp8 --steps 2 --warmup 1
