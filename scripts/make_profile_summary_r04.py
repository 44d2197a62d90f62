"""profiles/r04_c3_step_summary.md from the committed rocprofv3 kernel stats, PMC traffic and bench line of round 4.

    python scripts/make_profile_summary_r03.py
"""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, "profiles", *a)
rows = list(csv.DictReader(open(P("r04_c3_step_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tr = json.load(open(P("pmc_traffic.json")))
line = json.loads([l for l in open(P("r04_c3_bench_line.log")) if l.startswith("{")][-1])
pl = line["roofline"]["per_launch"]
out = ["# Round 4 - C3 train step, 1x MI355X, rocprofv3 --kernel-trace --stats (final build of the round)\n",
       "Command (GPU box, `scripts/collect_r04_profiles.sh c3stats`): `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o r3 -- "
       "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras`\n",
       f"4 train steps (1 warm-up incl. the one-time zero-fills of the activation / V buffers + 3 timed) of SynthesisModelCNN 128ch x 400t, batch 256, "
       f"fp32.  Total kernel time {tot/1e6:.1f} ms = {tot/1e6/4:.1f} ms/step (the one-time fills are ~14 ms of it); un-profiled default run "
       f"(`python bench.py --steps 20 --warmup 5`) {line['ms_per_step']:.1f} ms/step = {line['value']:.0f} mel-frames/s (`r04_c3_bench_line.log`; "
       f"the boxes of the pool differ by up to 3 %): the stream is never idle.\n",
       "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|"]
for r in rows[:26]:
    out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.1f} |")
out.append("")
out.append("Kernel names: `wino43v_nt_kernel<5>` = <POOLV>: conv2 forward, Winograd F(4,3) on the pre-transformed input V (both operands by "
           "LDS-DMA) + bias + LeakyReLU + max-pool + arg-max and sign bits, and - new this round - V = the F(4,3) input transform of its pooled "
           "output for conv3 (the raw rows of stage 2 are not stored; `wino43_v_fixup_kernel` finishes the last quad of every tile); "
           "`wino43v_nt_kernel<2>` = <POOL>: conv3 forward; `<3>` = <MASK>: conv3 input gradient on Vd (the transformed un-pooled dZ); `<4>` = "
           "<C1WGRAD>: conv2 input gradient on Vd whose epilogue contracts the result with the raw signal into the conv1 weight gradient (G1 is "
           "never stored); `wino43v_tn8_kernel<true>`: conv2 / conv3 weight gradient, 128 x 64 tile, V, the pooled gradient rows and their "
           "arg-max words by LDS-DMA, one launch whose workgroups take turns at writing Vd for the input gradient; `conv1_fwd_vq_kernel`: conv1 + "
           "LeakyReLU + pool writing V1 (P1 is not stored); there is no `wino43_xform_kernel` row any more; "
           "`nt_window_kernel<128,...>` / `tn_window_kernel<.>`: direct-form MFMA kernels for conv4, conv5, the 1x1 stack and the Linear layer; "
           "`nt_window_kernel<32, 0, 0>` / `tn_skinny_kernel`: the h.W_hh^T / dgates.W_hh passes over the 5.4 GB LSTM weight; "
           "`nadam_lowrank_kernel`: NAdam on that weight from its gradient factors.\n")
fw = pl['conv2_fwd']['ms']
out.append("Agreement with bench.py's HIP-event timers (roofline.per_launch of the bench line): the rocprof average of a kernel name is the mean "
           f"over its launches, e.g. conv2 forward {fw:.1f} ms vs the `wino43v_nt_kernel<5>` row, weight gradient "
           f"({pl['conv2_wgrad']['ms']:.1f} + {pl['conv3_wgrad']['ms']:.1f})/2 vs the `wino43v_tn8_kernel<true>` row (the two runs are separate "
           "gpurun calls: boxes of the pool differ by a few per cent; the timers of the bench line come from a second, untimed pass).\n")
out.append("HBM-side traffic per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same bench command "
           "(`collect_r04_profiles.sh c3fetch / c3write`), reduced by `scripts/pmc_traffic.py` into `profiles/pmc_traffic.json`; FETCH_SIZE doubled per "
           "the gfx950 note in MI355X_MICROARCH.md):\n")
out.append("| kernel | read GB | write GB | algorithmic GB read / written |\n|---|---|---|---|")
alg = {"<POOLV>": "V1 20.1 + taps / V2 9.9 + bits 0.4 + halo",
       "<POOL>": "V2 9.9 + taps / pooled output 3.3 + bits 0.2",
       "C1WGRAD": "Vd2 19.7 + bit words 1.0 + x 0.05 = 20.8 / partial sums 0.1",
       "UNPOOL,MASK": "Vd3 9.9 + bits 0.2 = 10.1 / G2 6.6",
       "tn8_kernel<true>": "V 15.0 + G (6.6 + 3.3) / 2 + bits = 20.4 / Vd (19.7 + 9.9) / 2 = 14.8 + split-K slabs 0.4",
       "tn_kernel<false": "(64-wide tile, not the default) 7/8 of V (15.0) + G (6.6 + 3.3) / 2 + bits = 18.3 / split-K slabs 0.7",
       "tn_kernel<true": "(64-wide tile, not the default) 1/8 of V 1.9 + G 5.0 + bits = 7.0 / Vd (19.7 + 9.9) / 2 = 14.8 + slabs 0.1"}
for k, v in tr.items():
    key = [a for a in alg if a in k]
    out.append(f"| {k} | {v['read_bytes']/1e9:.1f} | {v['write_bytes']/1e9:.1f} | {alg[key[0]] if key else ''} |")
out.append("\nThe weight-gradient kernel reads its algorithmic minimum (the four C_in tiles of a (split, C_out tile) share the gradient rows through the "
           "L2).  The NT kernels fetch about 2 x their algorithmic bytes (the six taps of a stage, 6.3 MB, thrash the 4 MB L2 of an XCD beside the streaming "
           "operand: it is V once plus the tap set once per round of tiles, the floor of this tile order - `r04_kernel_notes.md` section 4); writes are at the "
           "algorithmic minimum.  Ablations, microbenchmarks, SQ counters (`r04_sq_counters.txt`) and what was tried this round: `r04_kernel_notes.md`.  "
           "(The PMC passes were taken one build before the final one; the kernels' memory streams did not change in between.)")
open(P("r04_c3_step_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
