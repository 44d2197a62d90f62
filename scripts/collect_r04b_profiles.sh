#!/bin/bash
# Round-4 evidence runs of the F(6,3) default on the GPU box (one gpurun call per section; outputs under gpurun_out/r04b/, summaries
# copied into profiles/ afterwards).  Counter passes run on their own (one counter, no trace domain beside --kernel-trace).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${R04TAG:-r04b}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras"
case "$1" in
  c3stats) rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o r3 -- $BENCH > $O/c3_stats_run.log 2>&1 ;;
  c3fetch) rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH > $O/c3_fetch_run.log 2>&1 ;;
  c3write) rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH > $O/c3_write_run.log 2>&1 ;;
  c3clock) rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c -o c -- $BENCH > $O/c3_clock_run.log 2>&1 ;;
  *) echo "unknown section $1"; exit 2 ;;
esac
echo "section $1 done"
