#!/bin/bash
# PMC passes over one direct-conv configuration.  usage: c3_pmc.sh <layer> <tile>
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(pwd)"
OUT="$R/gpurun_out/c3pmc_$1_$2"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$OUT/a" -o a -- python3 "$R/scripts/c3_one.py" $1 $2 > "$OUT/a.log" 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- python3 "$R/scripts/c3_one.py" $1 $2 > "$OUT/b.log" 2>&1 || exit 1
cd "$R" && python3 scripts/pmc_summary.py "$OUT" | grep -v "at::\|elementwise\|distribution" 
