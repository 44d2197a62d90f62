"""A/B of the Winograd F(2,3) conv kernels against the direct MFMA kernels (same engine, same buffers).

    python scripts/check_wino.py [--batch 4] [--channels 16] [--time] 
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
ap.add_argument("--f43", action="store_true", help="F(4,3) NT kernels as the Winograd arm")
ap.add_argument("--only-wino", action="store_true", help="timing of the Winograd kernels only")
args = ap.parse_args()
dev = torch.device("cuda:0")
stages_def = [(512, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]
eng = CnnEngine(80, args.channels, args.timepoints, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
eng.wino_vout = False          # stage kernels one at a time: every stage reads P
B = args.batch
eng.wino43 = args.f43
eng.fuse_c1 = False          # stage kernels in isolation: keep G1 as a tensor
eng._alloc(B, dev)
eng._alloc_bwd()
g = torch.Generator(device=dev).manual_seed(1)


def fill():
    for k in eng.P:
        eng.P[k].normal_(generator=g)
    for k in eng.G:
        eng.G[k].normal_(generator=g)
    for k in eng.bits:
        eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)


def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


fill()
for si in (() if args.only_wino else (2, 3)):
    st = eng.stages[si - 2]
    w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
    b = torch.randn(st.cout, device=dev, generator=g) * 0.1
    res = {}
    for mode in (False, True):
        eng.wino = mode
        Pin = eng.P[si - 1].clone(); Gin = eng.G[si].clone(); bits_in = eng.bits[si].clone()
        eng.P[si].zero_(); eng.G[si - 1].zero_()
        gw, gb = torch.zeros_like(w), torch.zeros_like(b)
        eng.stage_dgrad(st, w)
        dg = eng.G[si - 1].clone()
        eng.stage_wgrad(st, gw, gb)
        eng.stage_forward(st, w, b)           # overwrites P[si] and bits[si]
        res[mode] = (eng.P[si].clone(), eng.bits[si].clone(), dg, gw.clone(), gb.clone())
        eng.P[si - 1].copy_(Pin); eng.G[si].copy_(Gin); eng.bits[si].copy_(bits_in)
    d, wn = res[False], res[True]
    nb = (d[1] ^ wn[1])
    nflip = sum(int(((nb >> k) & 1).sum()) for k in range(32))
    print(f"conv{si}: fwd rel {rel(wn[0], d[0]):.3e}  max {float((wn[0]-d[0]).abs().max()):.3e}  bit flips {nflip} of {d[1].numel()*32}"
          f"  dgrad rel {rel(wn[2], d[2]):.3e}  wgrad rel {rel(wn[3], d[3]):.3e}  bias {rel(wn[4], d[4]):.3e}", flush=True)

if args.iters:
    for si in ((2,) if args.only_wino else (2, 3)):
        st = eng.stages[si - 2]
        w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
        b = torch.randn(st.cout, device=dev, generator=g) * 0.1
        gw, gb = torch.empty_like(w), torch.empty_like(b)
        fl = 2.0 * B * eng.C * st.tc * st.k * st.cin * st.cout
        for mode in ((True,) if args.only_wino else (False, True)):
            eng.wino = mode
            for name, fn in (("fwd", lambda: eng.stage_forward(st, w, b)), ("dgrad", lambda: eng.stage_dgrad(st, w)),
                             ("wgrad", lambda: eng.stage_wgrad(st, gw, gb))):
                fn(); torch.cuda.synchronize()
                eng.enable_timers(True)
                for _ in range(args.iters):
                    fn()
                ts = eng.timer_summary(); eng.enable_timers(False)
                ms = ts[f"conv{si}_{name}"][1]
                print(f"{'wino  ' if mode else 'direct'} conv{si}_{name:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s (direct-conv FLOPs)", flush=True)
