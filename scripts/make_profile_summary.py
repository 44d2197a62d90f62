"""profiles/<tag>_c3_step_summary.md from the committed rocprofv3 kernel stats (profiles/<tag>_c3_step_kernel_stats.csv), the PMC
traffic / clock reductions (profiles/pmc_traffic.json, effective_clock.json) and the bench line (profiles/<tag>_c3_bench_line.log).

    python scripts/make_profile_summary.py r06 [raw-dir]
"""
import csv, json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
RAW = sys.argv[2] if len(sys.argv) > 2 else TAG          # directory under gpurun_out/ the passes were collected into
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)
rows = list(csv.DictReader(open(P(f"{TAG}_c3_step_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tr = json.load(open(P("pmc_traffic.json")))
ck = json.load(open(P("effective_clock.json")))["effective_clock_ghz"]
line = json.loads([l for l in open(P(f"{TAG}_c3_bench_line.log")) if l.startswith("{")][-1])
roof = line["roofline"]
pl = roof["per_launch"]
out = [f"# Round {int(TAG[1:3])} - C3 train step, 1x MI355X, rocprofv3 --kernel-trace --stats (final build of the round, `gpurun_out/{RAW}`)\n",
       f"Command (GPU box, `RTAG={RAW} scripts/collect_profiles.sh bench c3stats c3fetch c3write c3clock`, one call, one box): "
       "`rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o r3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
       "--no-kernel-timers --no-extras`\n",
       f"4 train steps (1 warm-up incl. the one-time zero-fills + 3 timed) of SynthesisModelCNN 128ch x 400t, batch 256, fp32.  Total kernel time "
       f"{tot/1e6:.1f} ms = {tot/1e6/4:.1f} ms/step under the profiler; the un-profiled default run of the same call (`python bench.py --steps 20 --warmup 5`, "
       f"`{TAG}_c3_bench_line.log`) **{line['ms_per_step']:.1f} ms/step = {line['value']:.0f} mel-frames/s**, `roofline.frac` {roof['frac']:.4f} "
       f"({roof['kernel'][:40]}..., {roof['avg_launch_ms']} ms per launch by HIP events), `step_mfma_issued_frac` {roof['step_mfma_issued_frac']:.4f}; "
       f"`cpu_baseline` {line['cpu_baseline']['value']} mel-frames/s measured live at micro-batch 16 over 5 timed steps on {line['cpu_baseline']['cores']} threads"
       + (f"; at the metric's batch of 256 (static, `profiles/cpu_baseline_b256.json`): {line['cpu_baseline']['measured_at_batch_256']['value']} mel-frames/s.\n"
          if isinstance(line['cpu_baseline'].get('measured_at_batch_256'), dict) else ".\n"),
       "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|"]
for r in rows[:28]:
    out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.1f} |")
out.append("")
out.append("Kernel names: `wino63v_nt_kernel<5>` = <POOLV> conv2 forward (writes V2), `<4>` = <C1WGRAD> conv2 input gradient + fused conv1 weight "
           "gradient, `<2>` = <POOL> conv3 forward, `<6>` = <MASKY> conv3 input gradient writing Y2 / Vd2; `wino63v_tn4y_kernel`: both weight "
           "gradients; `<7>` = <GY> conv4's input gradient on the same kernel (six batches, writes Y3 / Vd3) behind `wino63_unpool_rows6_kernel`; `conv1_fwd_vh_kernel`: conv1 writing V1; "
           "`nadam_lowrank_kernel<true>` (round 6): NAdam on W_hh from its gradient factors AND, in the same pass, the last BPTT product dh_1 = dgates_2 . W_hh - "
           "three `tn_skinny_kernel` launches per step are left (were four), four `nt_window_kernel<32, 0, 0>` (the forward passes over W_hh).\n")
out.append("HIP-event timers of the bench line (second, untimed pass) against the rocprof averages above: "
           + ", ".join(f"{k} {v['ms']:.2f}" for k, v in sorted(pl.items()) if k.startswith(("conv2", "conv3", "conv4"))) + " ms.\n")
out.append("| family | ms (HIP events) | issued TFLOP/s | of 157.3 nominal | held clock GHz (GRBM pass) | at the held clock | HBM-side read GB | write GB |\n|---|---|---|---|---|---|---|---|")
for fam, v in roof["families"].items():
    t = tr.get(fam, {})
    g = ck.get(fam)
    out.append(f"| {fam[:70]} | {v['ms_per_step']:.2f} | {v['issued_tflops']:.1f} | {v['frac']:.3f} | {g if g else '-'} | "
               f"{(v['frac'] / (g / 2.4)):.3f} |" .replace("| - | nan |", "| - | - |") if g else
               f"| {fam[:70]} | {v['ms_per_step']:.2f} | {v['issued_tflops']:.1f} | {v['frac']:.3f} | - | - |"
               )
    out[-1] += f" {t.get('read_bytes', 0)/1e9:.1f} | {t.get('write_bytes', 0)/1e9:.1f} |"
out.append("")
def gb(fam_key, what):
    for k, v in tr.items():
        if fam_key in k:
            return v[what] / 1e9
    return float("nan")


out.append("Traffic: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same bench command, FETCH_SIZE x 2 (gfx950 tallies the 128-B "
           "requests of 16-B/lane streaming reads at 64 B), per launch.  Against the algorithmic bytes: conv2 forward writes "
           f"{gb('<POOLV>', 'write_bytes'):.1f} GB for 9.1 GB of V2 + 0.4 GB of bit words (round 4: 14.2 GB - the difference was spill traffic), reads "
           f"{gb('<POOLV>', 'read_bytes'):.1f} GB for 18.3 GB of V1 (V once + the 8.4 MB tap set per round of tiles against a 4 MB L2: Infinity-Cache traffic, "
           f"`r04_kernel_notes.md` 4); `<MASKY>` writes {gb('<MASKY>', 'write_bytes'):.1f} GB (Y2 + Vd2 = 36.6) and reads {gb('<MASKY>', 'read_bytes'):.1f} GB for "
           "9.1 GB of Vd3 + 0.9 GB of bit words: the same tap re-fetch as conv3 forward plus what 36 GB of write-allocated lines push out of the "
           "L2s on their way through (section 7 of `r05_kernel_notes.md` has the store cache-policy experiment); `<GY>` (conv4's input gradient) writes "
           f"{gb('<GY>', 'write_bytes'):.1f} GB (Y3 + Vd3 = 18.3) and reads {gb('<GY>', 'read_bytes'):.1f} GB for 1.7 GB of operand + 0.4 GB of bit words "
           f"(its 0.5 MB tap set stays in the L2), behind `wino63_unpool_rows6_kernel` ({gb('rows6', 'read_bytes'):.1f} GB read, {gb('rows6', 'write_bytes'):.1f} GB "
           "written); the stand-alone producer they replace moved 21.7 GB and the one-tap GEMM in front of it another 4.4 GB.\n")
open(P(f"{TAG}_c3_step_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:3000])
