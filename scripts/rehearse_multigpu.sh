#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: two ranks share cuda:0 and exchange over gloo.
# Exercises rendezvous, per-rank sharding, barriers, the MAX-over-ranks timing and the bucketed gradient
# exchange on its side stream.  Numbers printed by this run are meaningless (one card, gloo).
set -e
cd "$(dirname "$0")/.."
RVIT_BENCH_VERBOSE=1 RVIT_DP_VERBOSE=1 GLOO_SOCKET_IFNAME=lo RVIT_BENCH_BACKEND=gloo timeout -k 10 80 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --train-batch 32 --train-steps 2
