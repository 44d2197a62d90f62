#!/bin/bash
# Two ranks of the headline configuration (batch 256 per rank) on ONE GPU over gloo (host-staged
# collectives): exercises the data-parallel step at the real sizes; timings are not meaningful
# (the ranks time-share the device).
set -e
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 WORLD_SIZE=2 LOCAL_RANK=0 TONAL_DIST_BACKEND=gloo
ARGS="--gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers"
RANK=1 python bench.py $ARGS > /tmp/rank1_full.log 2>&1 &
P1=$!
RANK=0 python bench.py $ARGS
wait $P1
