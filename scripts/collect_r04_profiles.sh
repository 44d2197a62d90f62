#!/bin/bash
# Round-4 evidence runs on the GPU box (one gpurun call each section; outputs under gpurun_out/r04/, summaries are then
# copied into profiles/ by hand / scripts/make_profile_summary.py).  Counter passes run on their own (no trace domains
# beside --kernel-trace), as the pool requires.   usage: scripts/collect_r04_profiles.sh <section>
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras"
case "$1" in
  c3stats) rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o r3 -- $BENCH > $O/c3_stats_run.log 2>&1 ;;
  c3fetch) rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH > $O/c3_fetch_run.log 2>&1 ;;
  c3write) rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH > $O/c3_write_run.log 2>&1 ;;
  sq)      for p in fwd dgrad wgrad; do
             rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
               --kernel-trace --output-format csv -d $O/sq_$p -o sq -- python3 $R/scripts/bench_conv.py --stages 2 --passes $p --iters 2 > $O/sq_$p.log 2>&1
           done ;;
  sq_old)  export TONAL_WINO_V=0
           for p in fwd dgrad wgrad; do
             rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
               --kernel-trace --output-format csv -d $O/sqold_$p -o sq -- python3 $R/scripts/bench_conv.py --stages 2 --passes $p --iters 2 > $O/sqold_$p.log 2>&1
           done ;;
  c2)      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o l -- python3 $R/bench.py --model lite --channels 32 --timepoints 200 --batch 64 --steps 200 --warmup 20 --no-extras > $O/c2_run.log 2>&1 ;;
  c5)      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 $R/scripts/bench_c5.py --train-classifiers --no-stock-compare > $O/c5_run.log 2>&1 ;;
  signal)  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sig -o s -- python3 $R/scripts/bench_signal.py > $O/sig_run.log 2>&1 ;;
  *) echo "unknown section $1"; exit 2 ;;
esac
echo "section $1 done"
