#!/bin/bash
# Two ranks of the headline configuration (global batch 256, strong scaling: 128 windows per rank, label LSTM sharded
# by gate rows) on ONE GPU over gloo (host-staged collectives): exercises the data-parallel step at the real sizes;
# timings are not meaningful (the ranks time-share the device).
set -e
cd "$(dirname "$0")/.."
TONAL_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers
