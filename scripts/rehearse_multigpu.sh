#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: bench.py starts its own launcher (WORLD_SIZE unset), two ranks
# share cuda:0 and exchange over gloo (staged through pinned host memory by vit_amd/dp.py).  Exercises the spawn path,
# rendezvous, per-rank sharding, barriers, the MAX-over-ranks timing and the bucketed gradient exchange.
# Numbers printed by this run are meaningless (one card, gloo).
set -e
cd "$(dirname "$0")/.."
unset WORLD_SIZE RANK LOCAL_RANK
RVIT_BENCH_VERBOSE=1 RVIT_DP_VERBOSE=1 GLOO_SOCKET_IFNAME=lo RVIT_BENCH_BACKEND=gloo timeout -k 10 200 python bench.py --gpus 2 \
  --steps 3 --warmup 1 --batch 64 --train-batch 32 --train-steps 4 --train-timeout 120 --no-cpu-baseline --no-extra-legs
