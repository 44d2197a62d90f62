"""profiles/<round>_c3_step_summary.md from the committed rocprofv3 kernel stats, PMC traffic and bench line.

    python scripts/make_profile_summary.py r02
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = lambda *a: os.path.join(ROOT, "profiles", *a)
rows = list(csv.DictReader(open(P(f"{tag}_c3_step_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tr = json.load(open(P("pmc_traffic.json")))
line = json.loads([l for l in open(P(f"{tag}_c3_bench_line.log")) if l.startswith("{")][-1])
pl = line["roofline"]["per_launch"]
out = [f"# Round {int(tag[1:])} - C3 train step, 1x MI355X, rocprofv3 --kernel-trace --stats (final build of the round)\n",
       "Command (GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o r2 -- python3 bench.py "
       "--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras`\n",
       f"4 train steps (1 warm-up incl. one-time buffer zero-fills + 3 timed) of SynthesisModelCNN 128ch x 400t, batch 256, fp32. "
       f"Total kernel time {tot/1e6:.1f} ms = {tot/1e6/4:.1f} ms/step; un-profiled default run (`python bench.py --steps 20 --warmup 5`) "
       f"{line['ms_per_step']:.1f} ms/step = {line['value']:.0f} mel-frames/s (`{tag}_c3_bench_line.log`): the stream is never idle.\n",
       "| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|"]
for r in rows[:26]:
    out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.1f} |")
out.append("")
out.append("Kernel names: `wino43_tn_kernel`: conv2 / conv3 weight gradient, Winograd F(4,3), transforms applied at LDS-staging time "
           "(new this round; finalised by `wino43_wgrad_finalize_kernel`); `wino43_nt_kernel<0, 2, 1>` = <DIRECT loader, POOL epilogue>: conv2 / conv3 "
           "forward (Winograd F(4,3)) + bias + LeakyReLU + max-pool + arg-max and sign bits; `wino43_nt_kernel<1, 3, 1>` = <UNPOOL, MASK>: conv3 input "
           "gradient; `wino43_nt_kernel<1, 4, 1>` = <UNPOOL, C1WGRAD>: conv2 input gradient whose epilogue contracts the result with the raw signal into "
           "the conv1 weight gradient (G1 is never stored); `nt_window_kernel<128,...>` / `tn_window_kernel<.>`: direct-form MFMA kernels for conv4, "
           "conv5, the 1x1 stack and the Linear layer; `nt_window_kernel<32, 0, 0>` / `tn_skinny_kernel`: the h.W_hh^T / dgates.W_hh passes over the "
           "5.4 GB LSTM weight; `nadam_lowrank_kernel`: NAdam on that weight from its gradient factors.\n")
wg = (pl['conv2_wgrad']['ms'] + pl['conv3_wgrad']['ms']) / 2
out.append("Agreement with bench.py's HIP-event timers (roofline.per_launch of the bench line): the rocprof average of a kernel name is the mean "
           f"over its launches, e.g. weight gradient ({pl['conv2_wgrad']['ms']:.1f} + {pl['conv3_wgrad']['ms']:.1f})/2 = {wg:.1f} ms vs the "
           "`wino43_tn_kernel` row (profiled runs clock 1-3 % lower).\n")
out.append("HBM-side traffic per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of the same bench command, reduced by "
           "`scripts/pmc_traffic.py` into `profiles/pmc_traffic.json`; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md):\n")
out.append("| kernel | read GB | write GB | algorithmic GB read / written |\n|---|---|---|---|")
alg = {"C1WGRAD": "G2 6.6 + arg-max / sign / conv1 bits 1.0 + x 0.05 = 7.7 / partial sums 0.1",
       "UNPOOL,MASK": "G3 3.3 + bits 0.3 = 3.6 / G2 6.6",
       "DIRECT,POOL": "(13.4 + 6.6)/2 = 10.0 / (6.6 + 3.3)/2 + bits = 5.3",
       "wino43_tn": "(13.4 + 6.6 + 0.2 + 6.6 + 3.3 + 0.1)/2 = 15.1 / split-K slabs (128 / 64 splits) 0.8",
       "wino_tn": "(13.4 + 6.6 + 0.2 + 6.6 + 3.3 + 0.1)/2 = 15.1 / split-K slabs 0.13"}
for k, v in tr.items():
    key = [a for a in alg if a in k][0]
    out.append(f"| {k} | {v['read_bytes']/1e9:.1f} | {v['write_bytes']/1e9:.1f} | {alg[key]} |")
out.append("\nAttribution of the NT kernels' excess reads, issued MFMA rates and SQ counters: `" + tag + "_kernel_notes.md`.")
open(P(f"{tag}_c3_step_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
