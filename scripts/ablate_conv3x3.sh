#!/bin/bash
# timing attribution of the direct 3x3 conv: one run of bench_conv3x3.py per ablation mask (csrc/conv3x3.hip c3_ablate)
cd "$(dirname "$0")/.."
for m in 0 1 2 3 4 8 12 16 32 15 31; do
  echo "== RVIT_C3_ABLATE=$m"
  RVIT_C3_ABLATE=$m timeout -k 10 120 python scripts/bench_conv3x3.py 512 2>&1 | grep -E "layer[234] B=512 tile=(11|15) " || exit 1
done
