#!/bin/bash
# Rehearse bench.py's multi-rank path on ONE GPU: two ranks, both on cuda:0, gloo process group with
# host-staged collectives (RCCL refuses two ranks per device).  Small shapes; checks the JSON line.
set -e
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 LOCAL_RANK=0 TONAL_DIST_BACKEND=gloo
ARGS="--gpus 2 --steps 3 --warmup 1 --batch 8 --channels 16 --timepoints 200 --no-cpu-baseline"
RANK=1 python bench.py $ARGS > /tmp/rank1.log 2>&1 &
P1=$!
RANK=0 python bench.py $ARGS
wait $P1
