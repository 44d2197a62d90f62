#!/usr/bin/env python3
"""The CPU oracle's train step AT THE METRIC'S BATCH (256 windows of 128 ch x 400 samples), once: 1 warm-up + 3 timed steps
on the host cores of the GPU box (no GPU is touched).  BASELINE.md section 3 asks for the largest power-of-two micro-batch
that fits; batch 256 needs ~190 - 200 GB of eager activations (stage 1 alone keeps 80 GB: pre-activation, LeakyReLU output
and int64 pool indices) and the box has 300 GB.  The default bench run cannot afford this (minutes per step) - its live
`cpu_baseline` stays at micro-batches 8 / 16 and carries this file's figure as a static field.

    gpurun -- python scripts/cpu_baseline_b256.py            -> gpurun_out/r06/cpu_baseline_b256.json (copy to profiles/)

A watchdog thread ends the process (exit 3, no JSON) if its resident set passes --rss-limit-gb or the box's available memory
falls below 8 GB: an out-of-memory kill would take the box down with it."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def meminfo():
    d = {}
    for ln in open("/proc/meminfo"):
        k, v = ln.split(":")
        d[k] = int(v.split()[0]) / 2 ** 20
    return d


def rss_gb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"):
            return int(ln.split()[1]) / 2 ** 20
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rss-limit-gb", type=float, default=262.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06", "cpu_baseline_b256.json"))
    args = ap.parse_args()
    import bench
    need = 215.0
    ram, avail = bench.host_ram_gb(), meminfo().get("MemAvailable", 0.0)
    print(f"host: {ram:.0f} GB limit, {avail:.0f} GB available, {bench.host_threads()} threads", flush=True)
    if min(ram, avail) < need and args.batch >= 256:
        print(f"not enough memory for batch {args.batch} (~{need:.0f} GB needed): not started", flush=True)
        sys.exit(2)
    peak = [0.0]

    def watchdog():
        while True:
            r = rss_gb()
            peak[0] = max(peak[0], r)
            if r > args.rss_limit_gb or meminfo().get("MemAvailable", 1e9) < 8.0:
                print(f"watchdog: RSS {r:.0f} GB / available {meminfo().get('MemAvailable', 0):.0f} GB - giving up", flush=True)
                os._exit(3)
            time.sleep(0.5)
    threading.Thread(target=watchdog, daemon=True).start()

    import torch
    from oracle import synthesis_oracle as so
    threads = bench.host_threads()
    torch.set_num_threads(threads)
    C, T, D, B = 128, 400, 80, args.batch
    t0 = time.perf_counter()
    torch.manual_seed(0)
    params = so.init_cnn_params(D, C, T)
    state = so.NAdamState(params)
    lat = so.latent_length(T)
    print(f"parameters drawn in {time.perf_counter() - t0:.0f} s, RSS {rss_gb():.0f} GB", flush=True)
    gen = torch.Generator().manual_seed(1234)

    def one():
        x = torch.randn(B, C, T, generator=gen)
        tones = torch.randint(0, 4, (B,), generator=gen)
        syls = torch.randint(0, 2, (B,), generator=gen)
        lab = torch.tensor([[[int(s)] * 5, bench.TONE_MAP[str(int(t))]] for t, s in zip(tones, syls)], dtype=torch.float32)
        tgt = 10 * torch.randn(B, D, generator=gen)
        mask = (torch.rand(B, 64, lat, C, generator=gen) >= 0.5).float() * 2.0
        t = time.perf_counter()
        loss, mcd = so.train_step("cnn", params, None, state, x, lab, tgt, dropout_mask=mask)
        return time.perf_counter() - t, loss

    warm, loss0 = one()
    print(f"warm-up step {warm:.1f} s, loss {loss0:.4f}, peak RSS {peak[0]:.0f} GB", flush=True)
    times = []
    for i in range(args.steps):
        dt, loss = one()
        times.append(dt)
        print(f"step {i + 1}: {dt:.1f} s, loss {loss:.4f}, peak RSS {peak[0]:.0f} GB", flush=True)
    import statistics
    med = statistics.median(times)
    rec = {"value": round(B / med, 4), "unit": "mel-frames/s", "batch": B, "cores": threads, "kind": "port",
           "timed_steps": len(times), "s_per_step": [round(v, 2) for v in times], "s_per_step_median": round(med, 2),
           "warmup_s": round(warm, 2), "peak_rss_gb": round(peak[0], 1), "host_ram_gb": round(ram, 1),
           "torch": torch.__version__, "measured_unix": int(time.time()),
           "what": f"CPU oracle (oracle/synthesis_oracle.py, PyTorch-CPU fp32 restatement of the reference's step, "
                   f"models/synthesis_trainer.py:201-229) at the metric's own batch of {B} windows of {C} ch x {T} samples, dropout "
                   f"mask 0.5, NAdam on all 1.38 G parameters: 1 warm-up + {len(times)} timed steps on {threads} host threads of a GPU box"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
