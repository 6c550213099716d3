export ONLY=vitB
for us in 0 12 18 24; do for gr in 4 8 16; do echo "== spread $us us groups $gr"; RVIT_G8_SPREAD_US=$us RVIT_G8_GROUPS=$gr timeout -k 10 100 python scripts/bench_gemm_tiles.py 5 2>&1 | grep tile= || exit 1; [ $us = 0 ] && break; done; done
