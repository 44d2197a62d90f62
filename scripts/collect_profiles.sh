#!/bin/bash
# Evidence runs on the GPU box (outputs under gpurun_out/$RTAG/, default r06; the summaries are then built into profiles/ by
# scripts/pmc_traffic.py, scripts/pmc_clock.py and scripts/make_profile_summary.py <tag>).  Counter passes run on their own (no trace domains beside --kernel-trace), as the pool
# requires; the program itself follows `--` (no env / bash -c hop).   usage: [RTAG=r06] scripts/collect_profiles.sh <section> [...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${RTAG:-r06}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras"
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"
for sec in "$@"; do
  case "$sec" in
    c3stats) rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o r3 -- $BENCH > $O/c3_stats_run.log 2>&1 ;;
    c3fetch) rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $BENCH > $O/c3_fetch_run.log 2>&1 ;;
    c3write) rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $BENCH > $O/c3_write_run.log 2>&1 ;;
    c3clock) rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c -o c -- $BENCH > $O/c3_clock_run.log 2>&1 ;;
    sq)      # the default F(6,3) kernels of both stages, all three passes (POOLV, C1WGRAD, tn4y; POOL, MASKY, tn4y + the Y / Vd producer)
             rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq -o sq -- python3 $R/scripts/bench_conv63.py --iters 2 --stages 2,3,4 > $O/sq.log 2>&1 ;;
    c2)      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o l -- python3 $R/bench.py --model lite --channels 32 --timepoints 200 --batch 64 --steps 200 --warmup 20 --no-extras > $O/c2_run.log 2>&1 ;;
    c5)      rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 $R/scripts/bench_c5.py --train-classifiers --no-stock-compare > $O/c5_run.log 2>&1 ;;
    signal)  rocprofv3 --kernel-trace --stats --output-format csv -d $O/sig -o s -- python3 $R/scripts/bench_signal.py > $O/sig_run.log 2>&1 ;;
    bench)   python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line.log 2> $O/bench_line.err ;;
    *) echo "unknown section $sec"; exit 2 ;;
  esac
  echo "section $sec done ($?)"
done
