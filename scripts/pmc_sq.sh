#!/bin/bash
# SQ counter pass over one stage kernel (rocprofv3 --pmc, its own run: no trace domains beside it).
#   scripts/pmc_sq.sh <out-dir> <stage> <pass> [env assignments...]
# Reduce with scripts/pmc_sq_reduce.py <out-dir>.
set -e
out=$1; stage=$2; pass=$3; shift 3
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d "$out" -o sq -- python3 "$GRAFT_REPO_ROOT/scripts/bench_conv.py" --stages "$stage" --passes "$pass" --iters 2
