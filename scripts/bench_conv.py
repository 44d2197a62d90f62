"""Isolated timing of the conv-stage MFMA kernels at the north-star geometry (no model, no LSTM).

    python scripts/bench_conv.py [--batch 256] [--stages 2,3] [--iters 5]
Prints ms and TFLOP/s per kernel (algorithmic FLOPs of the valid rows, 2 FLOP/MAC).
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--stages", default="2,3")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--passes", default="fwd,dgrad,wgrad")
args = ap.parse_args()
dev = torch.device("cuda:0")
stages_def = [(512, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]
eng = CnnEngine(80, 128, 400, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
B = args.batch
eng.fuse_c1 = False          # stage kernels in isolation: keep G1 as a tensor
eng._alloc(B, dev)
eng._alloc_bwd()
for st in eng.stages:       # (a stage whose forward epilogue writes V for its successor keeps no raw pooled rows: give the
    if st.idx not in eng.P:  # stand-alone transform of the next stage something to read when that stage is timed alone)
        eng.P[st.idx] = torch.empty(eng.S * st.tp_out, st.cout, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
for k in eng.P:
    eng.P[k].normal_(generator=g)
for k in eng.G:
    eng.G[k].normal_(generator=g)
for k in eng.bits:
    eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
for si in [int(s) for s in args.stages.split(",")]:
    st = eng.stages[si - 2]
    w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
    b = torch.randn(st.cout, device=dev, generator=g) * 0.1
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    fl = 2.0 * B * eng.C * st.tc * st.k * st.cin * st.cout
    def dgrad():
        if eng._use_wino_vd(st):          # the V-form input gradient reads the Vd its weight-gradient kernel leaves
            eng.stage_wgrad(st, gw, gb)
        eng.stage_dgrad(st, w)
    for name, fn in (("fwd", lambda: eng.stage_forward(st, w, b)), ("dgrad", dgrad),
                     ("wgrad", lambda: eng.stage_wgrad(st, gw, gb))):
        if name not in args.passes.split(","):
            continue
        fn(); torch.cuda.synchronize()
        eng.enable_timers(True)
        for _ in range(args.iters):
            fn()
        ts = eng.timer_summary(); eng.enable_timers(False)
        ms = ts[f"conv{si}_{name}"][1]
        print(f"conv{si}_{name:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s  ({100 * fl / ms / 1e9 / 157.3:.1f}% of fp32 MFMA peak)", flush=True)
