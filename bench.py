"""Headline benchmark: mel-frames/sec of the SynthesisModelCNN train step on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md section 8d, "C3"): SynthesisModelCNN 128 ch x 400
samples -> 80 mel bins, 4 tones x 2 syllables (L = 5 dynamics), batch 256 per GPU, reference
defaults (dropout 0.5, NAdam lr 5e-4, coupled weight decay 0.004), LogisticRegression tone /
syllable classifiers on 8-channel slices, synthetic N(0,1) ECoG and 10*N(0,1) targets,
random-init weights.  A "step" is the whole body of the reference's batch loop
(models/synthesis_trainer.py:201-229): classifier forwards + argmax, label dynamics, forward,
L1 on truncated targets, backward, (DP: gradient reduction), NAdam, loss + MCD accumulation.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns 256 windows.

Extra objects on the JSON line:
  roofline     the dominant kernel of the step (the rocprofv3 kernel name with the largest total
               time; one name covers its conv2 and conv3 launches): mean algorithmic FLOPs per
               launch / mean launch time, measured with HIP events on the launch stream, against
               the dense fp32 MFMA peak (157.3 TFLOP/s); `traffic` = HBM bytes per launch from the
               committed rocprofv3 PMC passes (profiles/pmc_traffic.json).  Algorithmic FLOPs are
               those of the direct convolution; the Winograd F(2,3) kernels issue 2/3
               of them as MFMA work, so `achieved` can exceed `peak` - `mfma_pipe_frac` is the
               utilisation of the matrix pipe itself.
  cpu_baseline the CPU oracle (oracle/synthesis_oracle.py, PyTorch-CPU fp32 restatement of the
               reference) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TONE_MAP = {"0": [3, 3, 3, 3, 3], "1": [1, 2, 3, 4, 5], "2": [3, 2, 1, 2, 4], "3": [5, 4, 3, 2, 1]}
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, dense fp32 matrix peak


def conv_flops(eng, stage_idx: int, B: int) -> float:
    """Algorithmic FLOPs of one conv stage pass (2 FLOP/MAC, valid rows only), SURVEY 8d."""
    st = eng.stages[stage_idx - 2]
    return 2.0 * B * eng.C * st.tc * st.k * st.cin * st.cout


def host_threads() -> int:
    """Threads the host leg may use: the cgroup CPU quota / affinity of this box, not the
    machine-wide core count (a 1-GPU box owns a 16-core share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("TL_BENCH_CPU_THREADS", min(n, 32)))


def cpu_baseline(model, B_cpu: int, C: int, T: int, out_dim: int, budget_s: float = 40.0):
    """Time the oracle's train step on the host cores (bounded sample)."""
    from oracle import synthesis_oracle as so
    threads = host_threads()
    torch.set_num_threads(threads)
    params = {k: v.detach().to("cpu", copy=True) for k, v in model.named_parameters()}
    state = so.NAdamState(params)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(B_cpu, C, T, generator=gen)
    tones = torch.randint(0, 4, (B_cpu,), generator=gen)
    syls = torch.randint(0, 2, (B_cpu,), generator=gen)
    lab = torch.tensor([[[int(s)] * 5, TONE_MAP[str(int(t))]] for t, s in zip(tones, syls)], dtype=torch.float32)
    tgt = 10 * torch.randn(B_cpu, out_dim, generator=gen)
    mask = (torch.rand(B_cpu, 64, model.latent_len, C, generator=gen) >= 0.5).float() * 2.0   # Dropout(0.5)
    times = []
    t_all = time.perf_counter()
    for _ in range(1):
        t0 = time.perf_counter()
        so.train_step("cnn", params, None, state, x, lab, tgt, dropout_mask=mask)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s:
            break
    best = min(times)
    return {"value": B_cpu / best, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} train step(s) of the CPU oracle at micro-batch {B_cpu} (same model, "
                      f"C={C}, T={T}; best of {len(times)}: {best:.2f} s/step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="windows per GPU")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--timepoints", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--dropout", type=float, default=None, help="default: the model's own default (0.5 / 0.3)")
    ap.add_argument("--model", choices=["full", "lite"], default="full",
                    help="full = SynthesisModelCNN (headline, C3); lite = SynthesisLite (use --channels 32 "
                         "--timepoints 200 --batch 64 for BASELINE config C2)")
    args = ap.parse_args()

    from decode_tonal_langauge_amd import parallel
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite, SynthesisModelCNN
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    B, C, T, D = args.batch, args.channels, args.timepoints, 80

    # every rank draws the same 1.38 G initial weights on the host: share the cores between ranks
    torch.set_num_threads(max(1, host_threads() // max(world, 1)))
    torch.manual_seed(1234)
    if args.model == "lite":
        args.no_cpu_baseline = True
        model = SynthesisLite(D, C, T, **({} if args.dropout is None else {"dropout": args.dropout}))
    else:
        model = SynthesisModelCNN(D, C, T, **({} if args.dropout is None else {"dropout": args.dropout}))
    drop = args.dropout if args.dropout is not None else (0.3 if args.model == "lite" else 0.5)
    tone_m = LogisticRegressionClassifier(8 * T, 4)
    syl_m = LogisticRegressionClassifier(8 * T, 2)
    trainer = SynthesisTrainer(model, tone_m, syl_m, TONE_MAP, device=dev, verbose=False)
    eng = model._engine

    # synthetic data, resident in HBM: every rank builds the same global batches and takes its shard
    gen = torch.Generator(device=dev).manual_seed(1234)
    nb = 4
    GB = B * world
    data = []
    for _ in range(nb):
        data.append((torch.randn(GB, C, T, device=dev, generator=gen),
                     torch.randn(GB, 8, T, device=dev, generator=gen),
                     torch.randn(GB, 8, T, device=dev, generator=gen),
                     10 * torch.randn(GB, D, device=dev, generator=gen)))
    model.train()

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.train_step(*data[i % nb])
    if not args.no_kernel_timers and args.model == "full":
        eng.enable_timers(True)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.train_step(*data[(args.warmup + i) % nb])
    sync()
    dt = time.perf_counter() - t0
    tsum = eng.timer_summary() if getattr(eng, "timers", None) is not None else {}
    if hasattr(eng, "enable_timers"):
        eng.enable_timers(False)
    if dist.is_initialized():
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = GB * args.steps / dt

    if rank == 0:
        roof = None
        if tsum:
            # kernel families as rocprofv3 names them: one name covers the launches of several stages
            if getattr(eng, "wino", False):
                nt, nt_frac = ("wino43_nt_kernel", "F(4,3)") if getattr(eng, "wino43", False) else ("wino_nt_kernel", "F(2,3)")
                fused = getattr(eng, "fuse_c1", False) and eng._c1_fusable()
                fams = {f"{nt}<DIRECT,POOL> (conv2/conv3 forward, Winograd {nt_frac})": ["conv2_fwd", "conv3_fwd"],
                        "wino_tn_kernel (conv2/conv3 weight gradient, Winograd F(2,3))": ["conv2_wgrad", "conv3_wgrad"]}
                if fused:       # the stage-2 launch carries the fused conv1 weight-gradient epilogue: its own kernel name
                    fams[f"{nt}<UNPOOL,C1WGRAD> (conv2 input gradient + conv1 weight gradient, Winograd {nt_frac})"] = ["conv2_dgrad"]
                    fams[f"{nt}<UNPOOL,MASK> (conv3 input gradient, Winograd {nt_frac})"] = ["conv3_dgrad"]
                else:
                    fams[f"{nt}<UNPOOL,MASK> (conv2/conv3 input gradient, Winograd {nt_frac})"] = ["conv2_dgrad", "conv3_dgrad"]
                # MFMA FLOPs issued per algorithmic (direct-convolution) FLOP of each family
                issued_of = {k: (0.5 if "F(4,3)" in k else 2.0 / 3.0) for k in fams}
            else:
                fams = {"nt_window_kernel<128,UNPOOL,MASK> (conv input-gradient)": ["conv2_dgrad", "conv3_dgrad", "conv4_dgrad"],
                        "nt_window_kernel<128,DIRECT,POOL> (conv forward)": ["conv2_fwd", "conv3_fwd", "conv4_fwd"],
                        "tn3_kernel<UNPOOL> (conv weight-gradient)": ["conv2_wgrad", "conv3_wgrad"]}
                issued_of = {k: 1.0 for k in fams}
            stats = {}
            for fam, tags in fams.items():
                tags = [t for t in tags if t in tsum]
                if not tags:
                    continue
                fl = [conv_flops(eng, int(t[4]), B) for t in tags]
                ms = [tsum[t][1] for t in tags]
                stats[fam] = {"ms_per_step": sum(ms), "launches_per_step": len(tags),
                              "flops_per_launch": sum(fl) / len(tags), "avg_launch_ms": sum(ms) / len(tags),
                              "tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12}
            dom = max(stats, key=lambda k: stats[k]["ms_per_step"])
            d = stats[dom]
            issued = issued_of[dom]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get(dom)
            roof = {"bound": "mfma", "achieved": round(d["tflops"], 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(d["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "kernel": dom,
                    "flops_per_launch": d["flops_per_launch"], "avg_launch_ms": round(d["avg_launch_ms"], 3),
                    "launches_per_step": d["launches_per_step"],
                    "flop_convention": "algorithmic = direct convolution, 2 FLOP/MAC over valid rows (SURVEY 8d)",
                    "mfma_issued_per_algorithmic_flop": round(issued, 4),
                    "mfma_issued_tflops": round(d["tflops"] * issued, 2),
                    "mfma_pipe_frac": round(d["tflops"] * issued / PEAK_FP32_MFMA_TFLOPS, 4),
                    "families": {k: {"tflops": round(v["tflops"], 2), "ms_per_step": round(v["ms_per_step"], 2)}
                                 for k, v in stats.items()},
                    "per_launch": {t: {"ms": round(tsum[t][1], 3),
                                       "tflops": round(conv_flops(eng, int(t[4]), B) / (tsum[t][1] * 1e-3) / 1e12, 2)}
                                   for t in sorted(tsum)}}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs on rank 0 of the 1-GPU run only
            try:
                cpu = cpu_baseline(model, 2, C, T, D)
            except Exception as e:      # noqa: BLE001 - the baseline leg must not kill the bench line
                cpu = {"value": None, "error": repr(e)}
        line = {
            "metric": "mel-frames/sec (train step) SynthesisModelCNN, 128ch x 400t batch256" if args.model == "full"
                      else "mel-frames/sec (train step) SynthesisLite",
            "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{type(model).__name__} {C}ch x {T}t -> {D} mel, 4 tones x 2 syllables, L=5, "
                                   f"batch {B} per GPU (global {GB}), dropout {drop}, NAdam, "
                                   f"{model.get_nparams():,} params",
                       "per_gpu_batch": B, "global_batch": GB, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
