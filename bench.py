"""Headline benchmark: mel-frames/sec of the SynthesisModelCNN train step on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...            (no WORLD_SIZE in the environment: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md section 8d, "C3"): SynthesisModelCNN 128 ch x 400
samples -> 80 mel bins, 4 tones x 2 syllables (L = 5 dynamics), global batch 256, reference
defaults (dropout 0.5, NAdam lr 5e-4, coupled weight decay 0.004), LogisticRegression tone /
syllable classifiers on 8-channel slices, synthetic N(0,1) ECoG and 10*N(0,1) targets,
random-init weights.  A "step" is the whole body of the reference's batch loop
(models/synthesis_trainer.py:201-229): classifier forwards + argmax, label dynamics, forward,
L1 on truncated targets, backward, (DP: gradient exchange over RCCL), NAdam, loss + MCD.
Inputs are resident in HBM before the timed region.

Scaling (``--scaling``): ``strong`` (default, what BASELINE config C4 names: the SAME global batch
of 256 windows sharded by rows over the ranks) or ``weak`` (256 windows per rank).

Objects on the JSON line beside the contract fields:
  roofline     the dominant kernel family of the step (largest total time; one rocprofv3 kernel name
               covers its conv2 and conv3 launches).  ``achieved`` = MFMA FLOPs the kernel ISSUES per
               launch (the Winograd forms need 1/2 or 2/3 of the direct convolution's multiplies) /
               mean launch time measured with HIP events on the launch stream; ``frac`` = achieved /
               dense fp32 MFMA peak (157.3 TFLOP/s), always <= 1.  ``algorithmic_tflops`` is the same
               time priced at the direct convolution's FLOPs.  ``step_mfma_issued_frac`` =
               all MFMA FLOPs issued in one step / step time / peak.  ``traffic`` = HBM bytes per
               launch from the rocprofv3 PMC passes committed under profiles/ (static file, named);
               ``held_clock`` = the clock that kernel ran at in a committed GRBM_GUI_ACTIVE pass of this
               command and ``frac`` re-priced at it (static file, named; ``peak`` stays the nominal figure).
  cpu_baseline the CPU oracle (oracle/synthesis_oracle.py, PyTorch-CPU fp32 restatement of the
               reference) timed on this box's host cores at two micro-batches inside --cpu-budget seconds.
               ``value`` is MEASURED (largest micro-batch / its median step time); the rate at the metric's
               batch of 256 is an extrapolation and sits in ``extrapolated_to_batch_256``.
  gpu_state    shader clock / power cap / power draw of the card read from sysfs before and after the
               timed region (boxes of the pool differ by more than most per-kernel leads).
The timed region carries no instrumentation: the per-kernel HIP-event timers behind ``roofline`` are
collected in a second, untimed pass (--timer-steps, default 3).
  c2_lite      BASELINE config C2 (SynthesisLite 32 ch x 200, batch 64) on the same GPU.
  c5           BASELINE config C5 rehearsed on this GPU at its per-rank batch (64): signal stage -> deep classifiers (train
               mode) -> the headline model's train step; step and classifier milliseconds.
  signal_c5    the preprocess/signal stage of config C5 (256 ch x 24 000 samples @ 400 Hz): Hilbert
               envelope, filtfilt, FIR - kernel time by HIP events, algorithmic GB/s against the HBM
               peak and (Hilbert: the bound that applies) fp64 vector FLOP/s against its peak.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TONE_MAP = {"0": [3, 3, 3, 3, 3], "1": [1, 2, 3, 4, 5], "2": [3, 2, 1, 2, 4], "3": [5, 4, 3, 2, 1]}
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_FP64_VALU_TFLOPS = 78.6           # MI355X_MICROARCH.md, fp64 vector peak
PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md, HBM3E


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256,
                    help="global batch (strong scaling) / windows per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--timepoints", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="smaller micro-batch of the CPU-oracle leg (the larger is twice it)")
    ap.add_argument("--cpu-budget", type=float, default=130.0,
                    help="seconds of host time the CPU-oracle leg aims at (default 130: five timed steps at micro-batch 16 - "
                         "always - then 2-5 at micro-batch 8 for the extrapolation); >= 400 times micro-batches 32 and 64")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="skip the untimed second pass that collects the per-kernel HIP-event timers")
    ap.add_argument("--timer-steps", type=int, default=3, help="steps of that second pass")
    ap.add_argument("--init", choices=["auto", "each", "broadcast"], default="auto",
                    help="initial weights under data parallelism: every rank draws all 1.38 G of them (each), or rank 0 "
                         "draws and broadcasts (broadcast; auto = broadcast when more than one rank) - same bits either way")
    ap.add_argument("--lstm-shard", choices=["auto", "on", "off"], default="auto",
                    help="data parallel: the label LSTM sharded by gate rows (ten small dependent collectives per step replace "
                         "(N-1)/N of the W_hh streams and of its NAdam pass) or whole on every rank (one 9.4 MB all-reduce of the "
                         "factor rows); auto times both for two steps during warm-up and keeps the faster (max over ranks)")
    ap.add_argument("--no-extras", action="store_true", help="skip the c2_lite / signal_c5 sub-results")
    ap.add_argument("--dropout", type=float, default=None, help="default: the model's own default (0.5 / 0.3)")
    ap.add_argument("--model", choices=["full", "lite"], default="full",
                    help="full = SynthesisModelCNN (headline, C3); lite = SynthesisLite (use --channels 32 "
                         "--timepoints 200 --batch 64 for BASELINE config C2)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N without torchrun
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """Start one child process per GPU.  The parent never touches the GPU runtime: the devices are counted in a
    short-lived child (torch.cuda.device_count() may fall back to hipGetDeviceCount, which initialises the runtime in
    the calling process).  Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and run this file; rank 0 prints
    the JSON line on the inherited stdout.  The rendezvous port is picked by bind-and-release (a race with other
    processes on the box is possible in principle; a failed rendezvous ends the run non-zero, it cannot hang it:
    the first rank to fail takes its peers down)."""
    n = args.gpus
    try:
        have = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                  capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1])
    except (ValueError, IndexError, subprocess.SubprocessError):
        have = 0
    share = os.environ.get("TONAL_BENCH_SHARE_GPU") == "1"      # rehearsal: all ranks on device 0 over gloo
    if have < n and not share:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if share:
            env["TONAL_DIST_BACKEND"] = "gloo"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0 and rc == 0:
                    rc = r
                    for q in pending:            # a rank died: its peers would block in the next collective
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
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


def gpu_state(index: int = 0) -> dict:
    """Shader clock and power cap of the card as sysfs shows them (one read; the box-to-box spread of this pool is larger
    than most per-kernel leads, so a driver-run regression must be tellable from a slow box).  Best effort: absent files
    give an empty dict.  (The in-kernel clock of an MFMA-dense loop can read up to ~10 % below pp_dpm_sclk:
    MI355X_MICROARCH.md, DVFS give-back item 6.)"""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return out
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])
    try:
        for ln in open(os.path.join(dev, "pp_dpm_sclk")):
            if "*" in ln:
                out["sclk_mhz"] = int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
    except (OSError, ValueError, IndexError):
        pass
    for name, key, scale in (("power1_cap", "power_cap_w", 1e-6), ("power1_average", "power_avg_w", 1e-6),
                             ("power1_input", "power_input_w", 1e-6)):
        for p in glob.glob(os.path.join(dev, "hwmon", "hwmon*", name)):
            try:
                out[key] = round(int(open(p).read().strip()) * scale, 1)
            except (OSError, ValueError):
                pass
    return out


def host_ram_gb() -> float:
    """Memory this process may use: cgroup limit if set, else MemTotal."""
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            return int(lim) / 2 ** 30
    except (OSError, ValueError):
        pass
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemTotal"):
                return int(ln.split()[1]) / 2 ** 20
    except OSError:
        pass
    return float("nan")


def cpu_baseline_at_256():
    """profiles/cpu_baseline_b256.json (scripts/cpu_baseline_b256.py, one gpurun session): the oracle at batch 256, or None."""
    path = os.path.join(ROOT, "profiles", "cpu_baseline_b256.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    rec["source"] = "profiles/cpu_baseline_b256.json (static: measured once on a GPU box's host cores, not in this run)"
    return rec


def cpu_baseline(model, B_cpu: int, C: int, T: int, out_dim: int, B_big: int = 0, budget_s: float = 130.0):
    """The oracle's train step on the host cores, timed at two micro-batches so that the batch-independent part of
    a step (5.5 GB of LSTM weights streamed per LSTM step + NAdam over 1.38 G parameters, ~9 s) is separated from the
    per-window part: s/step = a + b * B.  ``value`` is the rate that fit gives at the metric's batch of 256 (batch 256
    itself needs ~190 GB of eager activations and minutes per step); the raw micro-batch rates are reported beside it.
    Bounded by ``budget_s`` seconds of CPU time (the default bench run must finish within minutes): one warm-up step,
    FIVE timed steps at the larger micro-batch (the measured ``value``: BASELINE.md's >= 5), then 2-5 at the smaller one
    (the second point of the extrapolation) as the budget allows - ~11 + 5 x 15 + 2 x 11 s on 16 host threads."""
    import torch
    from oracle import synthesis_oracle as so
    threads = host_threads()
    torch.set_num_threads(threads)
    min_steps = 2
    if budget_s >= 400.0 and not B_big:        # opt-in (--cpu-budget): larger micro-batches, >= 5 timed steps each
        B_cpu, B_big, min_steps = 32, 64, 5
    B_big = B_big or 2 * B_cpu                 # powers of two: 8 and 16 by default
    params = {k: v.detach().to("cpu", copy=True) for k, v in model.named_parameters()}
    state = so.NAdamState(params)
    gen = torch.Generator().manual_seed(1234)

    def one(Bc):
        x = torch.randn(Bc, C, T, generator=gen)
        tones = torch.randint(0, 4, (Bc,), generator=gen)
        syls = torch.randint(0, 2, (Bc,), generator=gen)
        lab = torch.tensor([[[int(s)] * 5, TONE_MAP[str(int(t))]] for t, s in zip(tones, syls)], dtype=torch.float32)
        tgt = 10 * torch.randn(Bc, out_dim, generator=gen)
        mask = (torch.rand(Bc, 64, model.latent_len, C, generator=gen) >= 0.5).float() * 2.0   # Dropout(0.5)
        t0 = time.perf_counter()
        so.train_step("cnn", params, None, state, x, lab, tgt, dropout_mask=mask)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    warm = one(B_cpu)
    times = {B_cpu: [], B_big: []}
    # the measured value first: FIVE timed steps at the larger micro-batch whatever the budget (>= 5), then the
    # smaller one - which only feeds the extrapolation - with what is left of it (at least min_steps)
    while len(times[B_big]) < 5:
        times[B_big].append(one(B_big))
    while len(times[B_cpu]) < 5:
        spent = time.perf_counter() - t_begin
        nxt = times[B_cpu][-1] if times[B_cpu] else warm
        if len(times[B_cpu]) >= min_steps and spent + nxt > budget_s:
            break
        times[B_cpu].append(one(B_cpu))
    m_small, m_big = statistics.median(times[B_cpu]), statistics.median(times[B_big])
    b = max((m_big - m_small) / (B_big - B_cpu), 1e-9)
    a = max(m_small - b * B_cpu, 0.0)
    s256 = a + 256 * b
    return {"value": round(B_big / m_big, 4), "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "value_is": f"MEASURED: micro-batch {B_big} / median step time (the largest batch timed); the rate at the "
                        "metric's batch of 256 is an extrapolation and is reported apart",
            "extrapolated_to_batch_256": {"value": round(256 / s256, 3), "unit": "mel-frames/s",
                                          "from": "s/step = a + b*B through the two micro-batch medians (two points: no residual)",
                                          "fit_a_s": round(a, 3), "fit_b_s_per_window": round(b, 4),
                                          "s_per_step_at_256": round(s256, 1)},
            "micro_batches": {str(k): {"timed_steps": len(v), "s_per_step_median": round(statistics.median(v), 3),
                                       "s_per_step_min": round(min(v), 3), "s_per_step_max": round(max(v), 3),
                                       "mel_frames_per_s": round(k / statistics.median(v), 3)} for k, v in times.items()},
            "warmup_steps": 1, "warmup_s": round(warm, 3), "budget_s": budget_s,
            "host_ram_gb": round(host_ram_gb(), 1), "torch": torch.__version__,
            "sample": f"CPU oracle (same model, C={C}, T={T}, dropout mask 0.5, NAdam on all 1.38 G parameters): 1 warm-up "
                      f"step, then {len(times[B_big])} timed steps at micro-batch {B_big} and {len(times[B_cpu])} at {B_cpu}, bounded by a "
                      f"{budget_s:.0f} s budget (a full batch of 256 needs ~190 GB of eager activations and minutes per "
                      f"step; the default bench run has to finish within minutes).  A step costs a = {a:.1f} s that does not "
                      f"depend on the batch (LSTM weight + NAdam traffic) plus b = {b:.2f} s per window, so the measured "
                      f"micro-batch rate understates the CPU at batch 256 (see extrapolated_to_batch_256)"}


def event_ms(fn, iters: int, warm: int = 2):
    """Mean wall time of ``fn`` in ms by HIP events on torch's current stream (where the kernels launch)."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def lite_subresult(dev, steps: int = 200, warm: int = 20):
    """BASELINE config C2: SynthesisLite 32 ch x 200 samples -> 80 mel, batch 64, full trainer step."""
    import torch
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    B, C, T, D = 64, 32, 200, 80
    torch.manual_seed(1234)
    model = SynthesisLite(D, C, T)
    tr = SynthesisTrainer(model, LogisticRegressionClassifier(8 * T, 4), LogisticRegressionClassifier(8 * T, 2),
                          TONE_MAP, device=dev, verbose=False)
    gen = torch.Generator(device=dev).manual_seed(1234)
    data = [(torch.randn(B, C, T, device=dev, generator=gen), torch.randn(B, 8, T, device=dev, generator=gen),
             torch.randn(B, 8, T, device=dev, generator=gen), 10 * torch.randn(B, D, device=dev, generator=gen))
            for _ in range(4)]
    model.train()
    for i in range(warm):
        tr.train_step(*data[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step(*data[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / steps * 1e3
    flop = 13.85e6 * B                               # 13.85 MFLOP per window per train step
    return {"workload": "SynthesisLite 32ch x 200t -> 80 mel, batch 64, dropout 0.3, NAdam, 919,312 params (C2)",
            "value": round(B / (ms * 1e-3), 1), "unit": "mel-frames/s", "ms_per_step": round(ms, 4), "steps": steps,
            "warmup": warm, "bound": "launch latency (~50 small kernels per step; tensors are KB-MB)",
            "algorithmic_gflops": round(flop / (ms * 1e-3) / 1e9, 1),
            "cpu_reference_ms_per_step": 19.0, "cpu_reference_note": "SURVEY section 6: reference trainer on 8 host cores"}


def with_kernels(setting: str, fn):
    """Run ``fn`` under TONAL_KERNELS=<setting> (appended to what is set), restoring the variable afterwards."""
    old = os.environ.get("TONAL_KERNELS")
    os.environ["TONAL_KERNELS"] = setting if not old else old + "," + setting
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("TONAL_KERNELS", None)
        else:
            os.environ["TONAL_KERNELS"] = old


def signal_subresult(dev, with_cpu: bool):
    """C5 signal stage: frequency_filter methods on 256 ch x 24 000 float32 samples @ 400 Hz, in HBM."""
    import numpy as np
    import torch
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    C, T, FS = 256, 24000, 400
    x_np = np.random.default_rng(0).standard_normal((C, T)).astype(np.float32)
    x = torch.from_numpy(x_np).to(dev)
    out = {"shape": [C, T], "fs": FS, "dtype_in": "f32", "dtype_out": "f64"}
    def hilbert_dft():
        return with_kernels("hilbert=fft", lambda: ff.hilbert_filter(x, FS, [70., 150.]))

    def hilbert_f32():
        return with_kernels("hilbert_f32=1", lambda: ff.hilbert_filter(x, FS, [70., 150.]))

    cases = [("hilbert", lambda: ff.hilbert_filter(x, FS, [70., 150.]), 8),
             ("hilbert_f32_math", hilbert_f32, 8),               # opt-in: fp32 transforms end to end (ols_bank_bl_kernel<float,8,float>)
             ("hilbert_dft_domain_path", hilbert_dft, 8),       # the path long (low-band / raw-rate) kernels take
             ("butter_filtfilt", lambda: ff.butter_filter(x, [0.3, 100], FS), 8),
             # opt-in (TONAL_KERNELS=butter=scan): time-parallel block scan, 2e-8 - 5e-8 from the sequential kernel
             ("butter_filtfilt_scan", lambda: with_kernels("butter=scan", lambda: ff.butter_filter(x, [0.3, 100], FS)), 8),
             ("fir390", lambda: ff.fir_bandpass_filter(x, FS, 390, [100.]), 4)]
    for name, fn, s_out in cases:
        ms = event_ms(fn, 10)
        gb = C * T * (4 + s_out) / 1e9
        rec = {"ms": round(ms, 4), "algorithmic_GBps": round(gb / ms * 1e3, 1),
               "frac_of_hbm_peak": round(gb / ms * 1e3 / PEAK_HBM_GBPS, 4),
               "channel_samples_per_s": round(C * T / ms * 1e3)}
        if name in ("hilbert", "hilbert_f32_math"):
            f32m = name == "hilbert_f32_math"
            taps = 209                                 # analytic_taps() at fs = 400 Hz, 70-150 Hz, tol 1e-13
            try:
                cfs, sds = ff.gaussian_bank([70., 150.], FS)[:2]
                taps = int(ff.analytic_taps(T, FS, cfs, sds)[0].shape[1])
            except Exception:                          # noqa: BLE001 - helper names are internal
                pass
            # Default path for this bank: band-limited overlap-save (tl_hilbert_ols_bl): per segment of 1024 - (taps - 1) outputs
            # one forward 1024-point transform (5 radix-4 stages of 256 butterflies, ~40 FLOP each) and, per band, four
            # wave-private 256-point inverses (4 stages of 64 butterflies each), the spectrum product and the magnitudes.
            # The time-domain bank it replaces: 8 bands x taps complex FMAs.
            nfft = 1024
            segs = C * -(-T // (nfft - (taps - 1)))
            fl = segs * ((5 + 8 * 4) * (nfft // 4) * 40.0 + 8 * nfft * (8 + 5))
            fl_plain = 2.0 * 2 * 8 * taps * C * T      # 8 bands x taps complex FMAs (2 real FMA = 4 FLOP) per sample
            prec, peak = ("fp32", PEAK_FP32_MFMA_TFLOPS) if f32m else ("fp64", PEAK_FP64_VALU_TFLOPS)   # (fp32 vector peak = 157.3 too)
            rec.update({"kernel": f"ols_bank_bl_kernel<float, 8, {'float' if f32m else 'double'}>", "math": prec,
                        "bound": f"{prec} VALU issue + LDS round trips of the in-LDS FFTs (band-limited overlap-save: 1024-point forward, "
                                 "four wave-private 256-point inverses per band, last exchange by lane swaps; 4 workgroups per CU)"
                                 + ("" if f32m else "; the time-domain form of the same convolution is fp64-VALU bound at 0.47 ms"),
                        "taps": taps, f"{prec}_tflops_issued": round(fl / (ms * 1e-3) / 1e12, 2),
                        f"{prec}_tflops_plain_bank_equivalent": round(fl_plain / (ms * 1e-3) / 1e12, 2),
                        f"{prec}_vector_peak_tflops": peak,
                        f"frac_of_{prec}_peak": round(fl / (ms * 1e-3) / 1e12 / peak, 4)})
        elif name == "hilbert_dft_domain_path":
            rec["bound"] = ("HBM / L2 (Bluestein chirp-z over radix-2 Stockham passes in fp64: 18 FFTs of 65 536 points per "
                            "channel, one pass over a 1 MB buffer per radix-2 stage); not the default for this band")
        elif name == "butter_filtfilt":
            rec["bound"] = "latency (fp64 IIR recurrence, sequential in time; bit-identical to scipy's loop: the default)"
        elif name == "butter_filtfilt_scan":
            rec["bound"] = ("fp64 VALU issue of the per-block recurrences (two passes over every block) + the per-channel scan of the "
                            "block-start states (compensated 8 x 8 products); opt-in, 2e-8 - 5e-8 from the default")
            rec["speedup_over_sequential"] = round(out["butter_filtfilt"]["ms"] / ms, 1)
        else:
            rec["bound"] = ("LDS round trips and barriers of the in-LDS FFT (391-tap causal FIR by overlap-save on the 1024-point "
                            "fp64 transform of tl_hilbert_ols; the time-domain kernel: 0.32 ms, fp64-VALU bound)")
        out[name] = rec
    # ---- the same kernels where occupancy is not the limit: 256 ch x 2.4 M samples (100 minutes at 400 Hz), 7.4 GB algorithmic ----
    try:
        TL = 2_400_000
        xl = torch.randn(C, TL, device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(1))
        long_cases = [("hilbert", lambda: ff.hilbert_filter(xl, FS, [70., 150.]), 8, "fp64 VALU issue + LDS round trips of the in-LDS FFTs"),
                      ("fir390", lambda: ff.fir_bandpass_filter(xl, FS, 390, [100.]), 4, "LDS round trips and barriers of the in-LDS FFT"),
                      ("butter_filtfilt_scan", lambda: with_kernels("butter=scan", lambda: ff.butter_filter(xl, [0.3, 100], FS)), 8,
                       "fp64 VALU issue of the block recurrences; time-major work buffers (2 x 9.8 GB) written and read twice")]
        lg = {"shape": [C, TL], "dtype_in": "f32"}
        for name, fn, s_out, bound in long_cases:
            ms = event_ms(fn, 3)
            gb = C * TL * (4 + s_out) / 1e9
            lg[name] = {"ms": round(ms, 3), "algorithmic_GBps": round(gb / ms * 1e3, 1),
                        "frac_of_hbm_peak": round(gb / ms * 1e3 / PEAK_HBM_GBPS, 4), "binding_unit": bound}
        out["signal_long"] = lg
        del xl
        torch.cuda.empty_cache()
    except Exception as e:                          # noqa: BLE001 - optional sub-result (a smaller card, a path with a size limit)
        out["signal_long"] = {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}
    if with_cpu:
        from oracle import signal_oracle as sg
        rows = 32                                   # bounded sample: 32 of the 256 channels
        t0 = time.perf_counter()
        sg.hilbert_filter(x_np[:rows], FS, [70., 150.])
        t1 = time.perf_counter()
        sg.butter_filter(x_np[:rows], [0.3, 100], FS)
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": f"{rows} of {C} channels, whole recording",
                               "hilbert_channel_samples_per_s": round(rows * T / (t1 - t0)),
                               "butter_channel_samples_per_s": round(rows * T / (t2 - t1))}
    return out


def rank_batch_probe(trainer, data, batches=(128, 64, 32), steps: int = 3):
    """Single-GPU train-step time at the per-rank batches of the 2 / 4 / 8-GPU strong-scaling points (the whole label LSTM on
    the rank, no exchange): what a measured scaling curve decomposes into - per-rank compute here, exchange + sharding effects as
    the difference (review item 9b).  Runs after everything that is timed; the model moves on by a few updates."""
    import torch
    out = {}
    for b in batches:
        shard = tuple(t[:b].contiguous() for t in data[0])
        trainer.train_step(*shard)                         # (re-allocates the engine's workspaces at this batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            trainer.train_step(*shard)
        torch.cuda.synchronize()
        out[str(b)] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    return {"ms_per_step_at_batch": out, "steps": steps,
            "note": "one GPU, whole LSTM on the rank, no exchange step: the compute part of the N = 256 / batch strong-scaling point"}


def c5_subresult(dev, model, steps: int = 3):
    """BASELINE config C5 rehearsed on ONE GPU at the per-rank batch of its 8-GPU run (512 / 8 = 64): raw 256-channel ECoG
    (24 000 samples @ 400 Hz) -> frequency_filter.run (Hilbert 70-150 Hz envelope) -> 400-sample windows -> CNNClassifier
    (syllable) + CNNRNNClassifier (tone, lstm_dim 800) on 64 + 64 channels, in train mode as the reference CLI runs them
    (train_synthesizer.py:275-284) -> the SynthesisModelCNN train step on 128 channels.  Uses the headline model (its weights
    move on by `steps` + 1 updates: this runs after everything that is timed) with a trainer of its own."""
    import torch
    from argparse import Namespace
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier, SynthesisTrainer
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    B, T = 64, 400
    g = torch.Generator(device=dev).manual_seed(0)
    raw = torch.randn(256, 24000, device=dev, generator=g)
    prm = Namespace(signal_freq=400, bands=[{"method": "hilbert", "params": {"freq_ranges": [70., 150.], "envelope": True}}])
    ff.run(raw, prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hg = ff.run(raw, prm)
    torch.cuda.synchronize()
    t_sig = time.perf_counter() - t0
    starts = torch.arange(B, device=dev) * ((24000 - T) // (B - 1))
    win = torch.stack([hg[:, int(s):int(s) + T] for s in starts.tolist()]).float()
    x_non, x_syl, x_tone = win[:, :128].contiguous(), win[:, 128:192].contiguous(), win[:, 192:].contiguous()
    tgt = 10 * torch.randn(B, 80, device=dev, generator=g)
    torch.manual_seed(0)
    syl = CNNClassifier(input_channels=64, input_length=T, n_classes=2)
    tone = CNNRNNClassifier(input_channels=64, input_length=T, n_classes=4, lstm_dim=800)
    tr = SynthesisTrainer(model, tone, syl, TONE_MAP, device=dev, verbose=False, train_classifiers=True)
    model.train()
    tr.train_step(x_non, x_syl, x_tone, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(x_non, x_syl, x_tone, tgt)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        tr._labels(x_tone, x_syl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr._labels(x_tone, x_syl)
        torch.cuda.synchronize()
        t_cls = (time.perf_counter() - t0) / steps
    return {"workload": "C5 rehearsal on 1 GPU at the per-rank batch 64 of the 8-GPU run: 256 ch x 24 000 raw -> Hilbert envelope -> "
                        "CNNClassifier + CNNRNNClassifier (train mode, 64 + 64 ch) -> SynthesisModelCNN 128 ch x 400 train step",
            "per_gpu_batch": B, "steps": steps, "train_step_ms": round(t_step * 1e3, 2),
            "of_which_classifier_forwards_ms": round(t_cls * 1e3, 2), "signal_ms_256x24000": round(t_sig * 1e3, 3),
            "value": round(B / t_step, 1), "unit": "mel-frames/s", "loss": float(tr._stats[2])}


def step_issued_flops(eng, B: int, U: int, L: int) -> float:
    """MFMA FLOPs one train step issues (forward + input gradient + weight gradient)."""
    tot = 0.0
    for st in eng.stages:
        alg = conv_flops(eng, st.idx, B)
        if getattr(eng, "wino63", False) and eng._f63(st):
            f_nt = eng.f63_issue_factor(st)
        elif eng._v43(st):
            f_nt = 0.5
        else:
            f_nt = 1.0
        f_tn = eng.wgrad_issue_factor(st)
        tot += alg * (2 * f_nt + f_tn)
    rows5 = B * eng.C * eng.lat
    for cin_t, cin_ld, cout_t, cout_ld in eng.concat_dims:
        tot += 3 * 2.0 * rows5 * cin_ld * cout_ld
    tot += 3 * 2.0 * B * eng.kflat * eng.out_dim
    # W_hh passes run on the U distinct label rows, padded to the 32-row tile
    tot += 2 * (L - 1) * 2.0 * 32 * 4 * eng.H * eng.H
    return tot


def replica_spread(modules, parallel, dist) -> float:
    """max over every parameter (and buffer) of (MAX over ranks - MIN over ranks) of two checksums of it - the fp64 sum
    and the fp64 2-norm.  0.0 exactly when all ranks hold the same values (up to the astronomically unlikely collision of
    both checksums); NaN anywhere gives NaN (reported, and != 0)."""
    import torch
    sums = []
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            t = t.detach()
            if not t.is_floating_point():
                t = t.double()
            sums.append(torch.sum(t, dtype=torch.float64))
            sums.append(torch.linalg.vector_norm(t.reshape(-1), dtype=torch.float64))
    v = torch.stack(sums)
    # (a NaN / inf checksum on ANY rank must read as NaN on EVERY rank - MIN / MAX reductions need not propagate NaN - or the
    # ranks would take different branches behind this call: the flag travels as a number of its own)
    bad = (~torch.isfinite(v)).any().to(torch.float64).reshape(1)
    v = torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)
    lo, hi = v.clone(), torch.cat([v, bad])
    parallel.all_reduce_(lo, op=dist.ReduceOp.MIN)
    parallel.all_reduce_(hi, op=dist.ReduceOp.MAX)
    if float(hi[-1].item()) != 0.0:
        return float("nan")
    return float((hi[:-1] - lo).abs().max().item())


def replica_spread_tl(modules, parallel) -> float:
    """The same equality check carried by the C-ABI RCCL handle (``tl_allreduce`` MIN / MAX, fp32 only): every fp64 checksum
    travels as three fp32 pieces (x = hi + mid + lo exactly, a deterministic function of x), so the ranks hold the same
    checksums exactly when every piece has spread 0.  The value returned is the largest piece spread - a yes / no figure
    (0.0 or not), not the size of the disagreement; non-finite checksums return NaN.  Validates the handle's MIN / MAX
    path on the first real multi-GPU run (review item 9a)."""
    import torch
    sums = []
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            t = t.detach()
            if not t.is_floating_point():
                t = t.double()
            sums.append(torch.sum(t, dtype=torch.float64))
            sums.append(torch.linalg.vector_norm(t.reshape(-1), dtype=torch.float64))
    v = torch.stack(sums)
    bad = (~torch.isfinite(v)).any().to(torch.float32).reshape(1)
    v = torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)
    hi = v.float()
    mid = (v - hi.double()).float()
    lo_ = (v - hi.double() - mid.double()).float()
    pieces = torch.cat([hi, mid, lo_]).contiguous()
    mn, mx = pieces.clone(), torch.cat([pieces, bad]).contiguous()
    import torch.distributed as dist
    parallel.all_reduce_(mn, op=dist.ReduceOp.MIN)
    parallel.all_reduce_(mx, op=dist.ReduceOp.MAX)
    if float(mx[-1].item()) != 0.0:
        return float("nan")
    return float((mx[:-1] - mn).abs().max().item())


def choose_lstm_mode(trainer, data, want: str, sync, parallel, dist, dev) -> dict:
    """--lstm-shard: run the label LSTM sharded by gate rows or whole on every rank.  ``auto`` times both for two steps
    each (after one step that allocates), max over ranks, and keeps the faster - DESIGN.md section 7's falsifier (i): the
    sharded form's ten dependent small collectives per step must cost less than the (N-1)/N of 14.6 ms it saves."""
    import torch
    rec = {"requested": want}
    if want in ("on", "off"):
        rec["sharded"] = trainer.set_lstm_shard(want == "on")
        return rec
    ms = {}
    for mode in (True, False):
        if trainer.set_lstm_shard(mode) != mode:
            continue                                   # (the model / label table does not allow the sharded form)
        trainer.train_step(*data[1 % len(data)])
        sync()
        t0 = time.perf_counter()
        for i in range(2):
            trainer.train_step(*data[(2 + i) % len(data)])
        sync()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tt, op=dist.ReduceOp.MAX)
        ms["sharded" if mode else "whole"] = float(tt.item()) / 2 * 1e3
    pick = min(ms, key=ms.get) if ms else "whole"
    rec["sharded"] = trainer.set_lstm_shard(pick == "sharded")
    rec["probe_ms_per_step"] = {k: round(v, 3) for k, v in ms.items()}
    return rec


def step0_record(step0, args, C, T, GB, drop):
    """The first step's loss / MCD beside the committed single-GPU value for the same seeded global batch."""
    if step0 is None:
        return None
    key = f"{args.model}:{C}x{T}:batch{GB}:dropout{drop}"
    rec = {"loss": step0[0], "mcd": step0[1], "key": key, "golden": None}
    path = os.path.join(ROOT, "profiles", "bench_step0_golden.json")
    if os.path.exists(path):
        with open(path) as f:
            g = json.load(f).get(key)
        if g:
            rel = max(abs(step0[0] - g["loss"]) / abs(g["loss"]), abs(step0[1] - g["mcd"]) / abs(g["mcd"]))
            rec.update(golden=g, rel_diff=rel, ok=bool(rel < 1e-4),
                       source="profiles/bench_step0_golden.json (1-GPU run of this command, committed)")
    return rec


# ------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    from decode_tonal_langauge_amd import parallel
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite, SynthesisModelCNN
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer

    if os.environ.get("TONAL_BENCH_SHARE_GPU") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # rehearsal on a one-GPU box (also under `python -m torch.distributed.run`, the driver's launch form): every rank on
        # device 0, collectives over gloo (RCCL refuses two ranks per device)
        os.environ["LOCAL_RANK"] = "0"
        os.environ.setdefault("TONAL_DIST_BACKEND", "gloo")
    rank, world, local = parallel.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    if os.environ.get("TONAL_BENCH_FAIL_RANK") == str(rank) and world > 1:
        # test hook (tests/test_gpu_dp.py): a rank that dies must end the whole run non-zero, not hang its peers
        print(f"bench.py: rank {rank} exits on request (TONAL_BENCH_FAIL_RANK)", file=sys.stderr)
        os._exit(3)
    C, T, D = args.channels, args.timepoints, 80
    if args.scaling == "strong":
        GB = args.batch
        if GB < world:
            raise SystemExit(f"global batch {GB} smaller than {world} ranks")
    else:
        GB = args.batch * world
    B = len(range(*parallel.shard_rows(GB, rank, world).indices(GB)))       # rows of this rank

    # initial weights: every rank draws the same 1.38 G values on the host (sharing its cores), or rank 0 draws with all
    # cores and broadcasts after the move to the GPU (5.5 GB over xGMI) - 8 x 40 s of host RNG become one
    bcast_init = world > 1 and args.init in ("auto", "broadcast")
    torch.set_num_threads(host_threads() if (bcast_init and rank == 0) else max(1, host_threads() // max(world, 1)))
    torch.manual_seed(1234)
    import contextlib
    with (parallel.skip_param_init() if (bcast_init and rank != 0) else contextlib.nullcontext()):
        if args.model == "lite":
            args.no_cpu_baseline = True
            model = SynthesisLite(D, C, T, **({} if args.dropout is None else {"dropout": args.dropout}))
        else:
            model = SynthesisModelCNN(D, C, T, **({} if args.dropout is None else {"dropout": args.dropout}))
    drop = args.dropout if args.dropout is not None else (0.3 if args.model == "lite" else 0.5)
    # The classifiers are drawn from a seed of their own: under --init broadcast the ranks other than 0 skip the model's
    # draws, so their global RNG stands elsewhere than rank 0's behind the model - and classifiers drawn from THAT state
    # would differ from rank to rank (different predicted labels per shard: no longer the single-process computation).
    torch.manual_seed(1234 + 7)
    tone_m = LogisticRegressionClassifier(8 * T, 4)
    syl_m = LogisticRegressionClassifier(8 * T, 2)
    trainer = SynthesisTrainer(model, tone_m, syl_m, TONE_MAP, device=dev, verbose=False)
    if bcast_init:
        parallel.broadcast_parameters_(model, src=0)
        parallel.broadcast_parameters_(trainer.tone_model, src=0)
        parallel.broadcast_parameters_(trainer.syllable_model, src=0)
    eng = model._engine
    dp_check = None
    if world > 1:
        # every rank must start from the same bits (model AND classifiers), whichever way they were initialised
        spread0 = replica_spread([model, trainer.tone_model, trainer.syllable_model], parallel, dist)
        dp_check = {"init_checksum_spread": spread0}
        if parallel.tl_active():
            dp_check["init_checksum_spread_tl_handle"] = replica_spread_tl([model, trainer.tone_model, trainer.syllable_model], parallel)
            spread0 = spread0 if dp_check["init_checksum_spread_tl_handle"] == 0.0 else float("nan")
        if spread0 != 0.0:
            raise SystemExit(f"bench.py: rank {rank}: initial parameters differ between the ranks (checksum spread {spread0})")

    # synthetic data, resident in HBM: every rank builds the same global batches and takes its shard
    gen = torch.Generator(device=dev).manual_seed(1234)
    nb = 4
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

    step0 = None
    lstm_mode = None
    for i in range(args.warmup):
        trainer.train_step(*data[i % nb])
        if i == 0:
            # loss / MCD of the first step on the seeded first global batch: under strong scaling the same number whatever
            # the rank count (rows sharded, masks hashed by the global element) - checked against the committed 1-GPU value
            s0 = trainer._stats[2:4].detach().clone().double()
            if dist.is_initialized():
                parallel.all_reduce_(s0)               # per-rank statistics carry their weight in the global mean
            step0 = [float(v) for v in s0.tolist()]
            if world > 1 and args.model == "full":
                lstm_mode = choose_lstm_mode(trainer, data, args.lstm_shard, sync, parallel, dist, dev)
    if world > 1:
        trainer.exchange_events = []
        trainer.exchange_wait_events = []
    state0 = gpu_state(local if world > 1 else 0)
    sync()
    # ---- the timed region: exactly args.steps train steps, nothing else (no per-kernel instrumentation) ----
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.train_step(*data[(args.warmup + i) % nb])
    sync()
    dt = time.perf_counter() - t0
    state1 = gpu_state(local if world > 1 else 0)
    exch_ms = exposed_ms = None
    if world > 1:
        ev = trainer.exchange_events
        trainer.exchange_events = None
        exch_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
        wev = getattr(trainer, "exchange_wait_events", None) or []
        trainer.exchange_wait_events = None
        exposed_ms = sum(a.elapsed_time(b) for a, b in wev) / max(args.steps, 1) if wev else None
    # ---- second, UNTIMED pass: per-launch HIP-event timers of the conv kernels (roofline / families) ----
    tsum = {}
    if not args.no_kernel_timers and args.model == "full":
        eng.enable_timers(True)
        for i in range(max(1, args.timer_steps)):
            trainer.train_step(*data[(args.warmup + args.steps + i) % nb])
        tsum = eng.timer_summary()
        eng.enable_timers(False)
    if world > 1:
        # Self-check of the data-parallel run: after re-assembling the shard-wise updated W_hh every rank must hold the same
        # bits (the replicas see identical reduced gradients and run deterministic kernels) - a spread means the ranks did
        # not stay in lockstep and the throughput above is not that of the single-process computation.
        trainer.sync_parameters()
        dp_check["param_checksum_spread"] = replica_spread([model], parallel, dist)
        if parallel.tl_active():
            # the same verdict through tl_allreduce MIN / MAX: a wrong result of the handle's reductions shows here
            dp_check["param_checksum_spread_tl_handle"] = replica_spread_tl([model], parallel)
            if dp_check["param_checksum_spread_tl_handle"] != 0.0:
                dp_check["param_checksum_spread"] = float("nan")
        dp_check["ranks_seen"] = dist.get_world_size()
        dp_check["backend"] = dist.get_backend()
        dp_check["lstm"] = lstm_mode
    if dist.is_initialized():
        tt = torch.tensor([dt, exch_ms or 0.0, exposed_ms or 0.0], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tt, op=dist.ReduceOp.MAX)
        dt, exch_ms = float(tt[0].item()), float(tt[1].item())
        exposed_ms = float(tt[2].item()) if exposed_ms is not None else None
    ms_per_step = dt / args.steps * 1e3
    value = GB * args.steps / dt

    if rank == 0:
        roof = None
        if tsum:
            # kernel families as rocprofv3 names them: one name covers the launches of several stages
            fams, issued_of = eng.kernel_families() if hasattr(eng, "kernel_families") else ({}, {})
            stats = {}
            for fam, tags in fams.items():
                tags = [t for t in tags if t in tsum]
                if not tags:
                    continue
                share = getattr(eng, "_fam_share", {}).get(fam, 1.0)     # a family may compute only part of its stages' FLOPs
                fl = [conv_flops(eng, int(t[4]), B) * share for t in tags]
                ms = [tsum[t][1] for t in tags]
                stats[fam] = {"ms_per_step": sum(ms), "launches_per_step": len(tags),
                              "flops_per_launch": sum(fl) / len(tags), "avg_launch_ms": sum(ms) / len(tags),
                              "tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12}
            # timer tag -> share of its stage's algorithmic FLOPs the launch computes (a family may cover part of an op)
            tag_share = {t: getattr(eng, "_fam_share", {}).get(fam, 1.0) for fam, tags in fams.items() for t in tags
                         if getattr(eng, "_fam_share", {}).get(fam, 1.0) != 1.0}
            dom = max(stats, key=lambda k: stats[k]["ms_per_step"])
            d = stats[dom]
            issued = issued_of[dom]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                if dom in tj:
                    traffic = dict(tj[dom], source="profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / "
                                                   "WRITE_SIZE passes of this command, committed; not re-measured here)")
            issued_tf = d["tflops"] * issued
            # the clock the chip held in this kernel under the same command (static, from a committed counter pass): the peak
            # above is priced at the nominal 2.4 GHz, which an MFMA-dense launch does not get
            held = None
            cpath = os.path.join(ROOT, "profiles", "effective_clock.json")
            if os.path.exists(cpath):
                with open(cpath) as f:
                    cj = json.load(f)
                ghz = cj.get("effective_clock_ghz", {}).get(dom)
                if ghz:
                    held = {"effective_clock_ghz": ghz, "nominal_clock_ghz": cj.get("nominal_clock_ghz", 2.4),
                            "frac_at_held_clock": round(issued_tf / PEAK_FP32_MFMA_TFLOPS / (ghz / cj.get("nominal_clock_ghz", 2.4)), 4),
                            "source": "profiles/effective_clock.json (static: GRBM_GUI_ACTIVE pass of this command, committed; "
                                      "not re-measured here)"}
            U = getattr(eng, "_U", 8)
            L = getattr(eng, "_L", 5)
            step_issued = step_issued_flops(eng, B, U, L)
            roof = {"bound": "mfma", "achieved": round(issued_tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(issued_tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "held_clock": held, "kernel": dom,
                    "flops_issued_per_launch": d["flops_per_launch"] * issued,
                    "avg_launch_ms": round(d["avg_launch_ms"], 3), "launches_per_step": d["launches_per_step"],
                    "achieved_is": "MFMA FLOPs issued per launch / HIP-event launch time (<= peak by construction)",
                    "algorithmic_tflops": round(d["tflops"], 2),
                    "algorithmic_flops_per_launch": d["flops_per_launch"],
                    "flop_convention": "algorithmic = direct convolution, 2 FLOP/MAC over valid rows (SURVEY 8d)",
                    "mfma_issued_per_algorithmic_flop": round(issued, 4),
                    "step_mfma_issued_tflop": round(step_issued / 1e12, 3),
                    "step_mfma_issued_frac": round(step_issued / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "families": {k: {"issued_tflops": round(v["tflops"] * issued_of[k], 2),
                                     "frac": round(v["tflops"] * issued_of[k] / PEAK_FP32_MFMA_TFLOPS, 4),
                                     "algorithmic_tflops": round(v["tflops"], 2), "ms_per_step": round(v["ms_per_step"], 2)}
                                 for k, v in stats.items()},
                    "per_launch": {t: ({"ms": round(tsum[t][1], 3),
                                        "algorithmic_tflops": round(conv_flops(eng, int(t[4]), B) * tag_share.get(t, 1.0)
                                                                    / (tsum[t][1] * 1e-3) / 1e12, 2)}
                                       if t.rsplit("_", 1)[-1] in ("fwd", "dgrad", "wgrad") or t in tag_share
                                       else {"ms": round(tsum[t][1], 3)})
                                   for t in sorted(tsum)}}
            # an op split over two kernel names (the weight gradient: the Vd-writing C_in tile + the other tiles): the whole op
            ops = {}
            for t in tsum:
                if t.endswith("_wgrad_vd") and t[:-3] in tsum:
                    ops[t[:-3]] = (conv_flops(eng, int(t[4]), B), tsum[t][1] + tsum[t[:-3]][1])
            if ops:
                fl, ms = sum(v[0] for v in ops.values()), sum(v[1] for v in ops.values())
                iss = issued_of[next(k for k, tags in fams.items() if next(iter(ops)) in tags)]
                roof["whole_ops"] = {"conv2/conv3 weight gradient (both launches)": {
                    "ms_per_step": round(ms, 2), "issued_tflops": round(fl * iss / (ms * 1e-3) / 1e12, 2),
                    "frac": round(fl * iss / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "per_stage_frac": {k: round(v[0] * iss / (v[1] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) for k, v in sorted(ops.items())}}}
        extras = {}
        if world == 1 and args.model == "full" and not args.no_extras:
            for key, fn in (("rank_batch_probe", lambda: rank_batch_probe(trainer, data)),
                            ("c2_lite", lambda: lite_subresult(dev)),
                            ("signal_c5", lambda: signal_subresult(dev, not args.no_cpu_baseline)),
                            ("c5", lambda: c5_subresult(dev, model))):
                try:
                    extras[key] = fn()
                except Exception as e:      # noqa: BLE001 - a sub-result must not kill the headline line
                    extras[key] = {"error": repr(e)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the LIVE CPU leg runs on rank 0 of the 1-GPU run only
            try:
                cpu = cpu_baseline(model, args.cpu_batch, C, T, D, budget_s=args.cpu_budget)
            except Exception as e:      # noqa: BLE001 - the baseline leg must not kill the bench line
                cpu = {"value": None, "error": repr(e)}
        static256 = cpu_baseline_at_256()
        if static256 is not None and args.model == "full" and not args.no_cpu_baseline:
            # the same oracle timed ONCE at the metric's own batch on a GPU box's host (static, committed file): carried on every
            # line, multi-GPU ones included (the live leg is skipped there)
            if cpu is None:
                cpu = {"value": None, "unit": "mel-frames/s", "kind": "port", "cores": static256.get("cores"),
                       "sample": "live leg skipped (world > 1): see measured_at_batch_256"}
            cpu["measured_at_batch_256"] = static256
            ex = cpu.get("extrapolated_to_batch_256")
            if isinstance(ex, dict) and static256.get("value"):
                ex["vs_measured_at_batch_256"] = round(ex["value"] / static256["value"], 3)
        line = {
            "metric": "mel-frames/sec (train step) SynthesisModelCNN, 128ch x 400t batch256" if args.model == "full"
                      else "mel-frames/sec (train step) SynthesisLite",
            "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{type(model).__name__} {C}ch x {T}t -> {D} mel, 4 tones x 2 syllables, L=5, "
                                   f"global batch {GB} ({args.scaling} scaling: {B} windows on rank 0), dropout {drop}, "
                                   f"NAdam, {model.get_nparams():,} params",
                       "per_gpu_batch": B, "global_batch": GB, "parallelism": f"dp{world}",
                       "backend": (dist.get_backend() if dist.is_initialized() else None),
                       "exchange_ms_per_step": None if exch_ms is None else round(exch_ms, 3),
                       "exchange_exposed_ms_per_step": None if exposed_ms is None else round(exposed_ms, 3),
                       "init": ("broadcast from rank 0" if bcast_init else "drawn on every rank")},
            "gpu_state": {"before_timed_region": state0, "after_timed_region": state1,
                          "note": "sysfs pp_dpm_sclk / hwmon power of this rank's card, one read each"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if roof is not None:
            roof["timers_from"] = f"a second, untimed pass of {max(1, args.timer_steps)} steps (the timed region carries no instrumentation)"
        line["step0"] = step0_record(step0, args, C, T, GB, drop)
        if dp_check is not None:
            dp_check["loss_step0_vs_dp1_golden"] = line["step0"]
            dp_check["ok"] = dp_check["param_checksum_spread"] == 0.0 and dp_check["init_checksum_spread"] == 0.0
            line["dp_check"] = dp_check
        line.update(extras)
        print(json.dumps(line), flush=True)
    bad = dp_check is not None and not (dp_check.get("param_checksum_spread") == 0.0)
    if dist.is_initialized():
        dist.barrier()
        parallel.tl_comm_destroy()                 # (no-op unless TONAL_DIST_BACKEND=tl created the C-ABI communicator)
        dist.destroy_process_group()
    if bad:
        print(f"bench.py: rank {rank}: replicas diverged (parameter checksum spread {dp_check.get('param_checksum_spread')})",
              file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
