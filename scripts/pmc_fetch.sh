#!/bin/bash
# HBM-side read traffic (FETCH_SIZE) of one stage kernel: scripts/pmc_fetch.sh <out-dir> <stage> <pass> [env...]
set -e
out=$1; stage=$2; pass=$3; shift 3
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out" -o f -- python3 "$GRAFT_REPO_ROOT/scripts/bench_conv.py" --stages "$stage" --passes "$pass" --iters 2 > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
tot, n = defaultdict(float), defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and float(r["Counter_Value"]) > 1e5:
            tot[r["Kernel_Name"][:50]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:50]] += 1
for k in tot:
    print(f"{k:52s} launches {n[k]}  read {tot[k] / n[k] * 1024 * 2 / 1e9:7.2f} GB per launch (FETCH_SIZE KiB x 1024 x 2)")
PY
