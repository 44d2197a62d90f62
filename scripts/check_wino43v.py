"""A/B of the V-form F(4,3) kernels (tonal_wino43v.hip: input transform hoisted into the producer, operands by
LDS-DMA) against the F(4,3) kernels that transform in the GEMM loop - same engine, same buffers.

    python scripts/check_wino43v.py [--batch 4] [--channels 16] [--iters 5] [--passes fwd,wgrad]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--channels", type=int, default=16)
ap.add_argument("--timepoints", type=int, default=400)
ap.add_argument("--iters", type=int, default=0)
ap.add_argument("--stages", default="2,3")
ap.add_argument("--passes", default="fwd,wgrad,dgrad")
args = ap.parse_args()
dev = torch.device("cuda:0")
stages_def = [(512, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]
eng = CnnEngine(80, args.channels, args.timepoints, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
eng.wino_vout = False          # stage kernels one at a time: every stage reads P
B = args.batch
eng.fuse_c1 = False
eng._alloc(B, dev)
eng._alloc_bwd()
g = torch.Generator(device=dev).manual_seed(1)
for k in eng.P:
    eng.P[k].normal_(generator=g)
for k in eng.G:
    eng.G[k].normal_(generator=g)
for k in eng.bits:
    eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)


def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


passes = args.passes.split(",")
bad = False
for si in [int(s) for s in args.stages.split(",")]:
    st = eng.stages[si - 2]
    # zero the pad rows of the input like the producing epilogue does (t >= tin)
    Pin = eng.P[si - 1].view(eng.S, st.tp_in, -1)
    Pin[:, st.tin:, :] = 0
    w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
    b = torch.randn(st.cout, device=dev, generator=g) * 0.1
    res = {}
    for mode in (False, True):
        eng.wino_v = mode
        eng._v_ready = {}
        bits_in = eng.bits[si].clone()
        out = {}
        if "wgrad" in passes:
            gw, gb = torch.zeros_like(w), torch.zeros_like(b)
            if mode:
                eng._v_ready = {}
            eng.stage_wgrad(st, gw, gb)
            out["gw"], out["gb"] = gw.clone(), gb.clone()
        if "dgrad" in passes:
            if mode:                       # the V-form input gradient reads the Vd its weight-gradient kernel wrote
                gw2, gb2 = torch.zeros_like(w), torch.zeros_like(b)
                eng.stage_wgrad(st, gw2, gb2)
            eng.G[si - 1].zero_()
            eng.stage_dgrad(st, w)
            out["dg"] = eng.G[si - 1].clone()
        if "fwd" in passes:
            eng.P[si].zero_()
            eng.stage_forward(st, w, b)
            out["P"], out["bits"], out["sbits"] = eng.P[si].clone(), eng.bits[si].clone(), eng.sbits[si].clone()
            eng.bits[si].copy_(bits_in)
        res[mode] = out
    d, v = res[False], res[True]
    msg = f"conv{si}:"
    if "fwd" in passes:
        nb = d["bits"] ^ v["bits"]
        nflip = sum(int(((nb >> k) & 1).sum()) for k in range(32))
        ns = d["sbits"] ^ v["sbits"]
        nsf = sum(int(((ns >> k) & 1).sum()) for k in range(32))
        r = rel(v["P"], d["P"])
        msg += f" fwd rel {r:.3e} max {float((v['P'] - d['P']).abs().max()):.3e} argmax flips {nflip} sign flips {nsf} of {d['bits'].numel() * 32}"
        bad |= not (r < 5e-6)
    if "dgrad" in passes:
        r3 = rel(v["dg"], d["dg"])
        msg += f"  dgrad rel {r3:.3e} max {float((v['dg'] - d['dg']).abs().max()):.3e}"
        bad |= not (r3 < 5e-6)
    if "wgrad" in passes:
        r1, r2 = rel(v["gw"], d["gw"]), rel(v["gb"], d["gb"])
        msg += f"  wgrad rel {r1:.3e} bias {r2:.3e}"
        bad |= not (r1 < 5e-5 and r2 < 5e-5)
    print(msg, flush=True)

if args.iters:
    for si in [int(s) for s in args.stages.split(",")]:
        st = eng.stages[si - 2]
        w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
        b = torch.randn(st.cout, device=dev, generator=g) * 0.1
        gw, gb = torch.empty_like(w), torch.empty_like(b)
        fl = 2.0 * B * eng.C * st.tc * st.k * st.cin * st.cout
        for mode in (False, True):
            eng.wino_v = mode
            eng._v_ready = {}
            if mode:
                eng.enable_timers(True)
                for _ in range(args.iters):
                    eng._input_transform(st)
                ts = eng.timer_summary(); eng.enable_timers(False)
                print(f"V-form conv{si}_xform  {ts[f'conv{si}_xform'][1]:8.3f} ms (stand-alone input transform)", flush=True)
            def dgrad():
                if mode:
                    eng.stage_wgrad(st, gw, gb)
                eng.stage_dgrad(st, w)
            for name, fn in (("fwd", lambda: eng.stage_forward(st, w, b)), ("wgrad", lambda: eng.stage_wgrad(st, gw, gb)),
                             ("dgrad", dgrad)):
                if name not in passes:
                    continue
                fn(); torch.cuda.synchronize()
                eng.enable_timers(True)
                for _ in range(args.iters):
                    fn()
                ts = eng.timer_summary(); eng.enable_timers(False)
                ms = ts[f"conv{si}_{name}"][1]
                extra = ""
                print(f"{'V-form' if mode else 'in-loop'} conv{si}_{name:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s algorithmic, "
                      f"{50 * fl / ms / 1e9 / 157.3:.1f}% of the fp32 MFMA peak issued{extra}", flush=True)
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
