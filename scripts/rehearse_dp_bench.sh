#!/bin/bash
# Rehearse bench.py's multi-rank path on ONE GPU: bench.py starts the two ranks itself; TONAL_BENCH_SHARE_GPU=1 puts
# both on cuda:0 over a gloo process group with host-staged collectives (RCCL refuses two ranks per device).
# Small shapes; prints the JSON line (parallelism dp2, backend gloo, exchange_ms_per_step).
set -e
cd "$(dirname "$0")/.."
TONAL_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 --channels 16 --timepoints 200 --no-cpu-baseline
