"""A/B of the V-form F(4,3) weight-gradient kernels of tonal_wino43v.hip - same engine, same buffers: C_in tile 64
(wino43v_tn_kernel<., 2>: 4 waves, two workgroups per CU), 127 (wino43v_tn_kernel<., 4>: 128-wide tile, 8 waves, Y staged
through registers) and 128 (wino43v_tn8_kernel: 128-wide tile, the Y side by LDS-DMA, one launch whose workgroups take
turns at Vd).  They must agree bit for bit (same k order per accumulator): weight gradient, bias gradient and the Vd.

    python scripts/check_tn_bm.py [--batch 4] [--channels 16] [--iters 5]
"""
import argparse, os, sys
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TONAL_TN_TARGET", "4096")     # the same reduction splits for every tiling (bit-identity needs that)
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--channels", type=int, default=16)
ap.add_argument("--timepoints", type=int, default=400)
ap.add_argument("--iters", type=int, default=0)
ap.add_argument("--stages", default="2,3")
ap.add_argument("--bms", default="64,128", help="tilings to compare: 64 (4 waves), 127 (8 waves, Y through registers), 128 (8 waves, Y by LDS-DMA)")
args = ap.parse_args()
bms = [int(b) for b in args.bms.split(",")]
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

bad = False
for si in [int(s) for s in args.stages.split(",")]:
    st = eng.stages[si - 2]
    Pin = eng.P[si - 1].view(eng.S, st.tp_in, -1)
    Pin[:, st.tin:, :] = 0
    w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
    res = {}
    for bm in bms:
        eng.tn_bm = bm
        eng._v_ready = {}
        if si in eng.Vd:
            eng.Vd[si].fill_(float("nan"))
        gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
        eng.stage_wgrad(st, gw, gb)
        nq = eng.S * st.tp_in // 4
        res[bm] = (gw.clone(), gb.clone(), eng.Vd[si][:nq].clone())
    a, b = res[bms[0]], res[bms[-1]]
    eq = [bool(torch.equal(x, y)) for x, y in zip(a, b)]
    fin = bool(torch.isfinite(b[2]).all())
    print(f"conv{si}: weight gradient equal {eq[0]}  bias gradient equal {eq[1]}  Vd equal {eq[2]} (finite {fin})  "
          f"|gw| {float(b[0].norm()):.4e}", flush=True)
    bad |= not (all(eq) and fin)

if args.iters:
    for si in [int(s) for s in args.stages.split(",")]:
        st = eng.stages[si - 2]
        w = torch.randn(st.cout, st.cin, st.k, 1, device=dev, generator=g) * 0.02
        gw, gb = torch.empty_like(w), torch.empty(st.cout, device=dev)
        fl = 2.0 * B * eng.C * st.tc * st.k * st.cin * st.cout
        for bm in bms + bms:
            eng.tn_bm = bm
            eng.stage_wgrad(st, gw, gb)
            torch.cuda.synchronize()
            eng.enable_timers(True)
            for _ in range(args.iters):
                eng.stage_wgrad(st, gw, gb)
            ts = eng.timer_summary()
            eng.enable_timers(False)
            a, b = ts.get(f"conv{si}_wgrad_vd", (0, 0.0))[1], ts[f"conv{si}_wgrad"][1]     # (bm 128: one launch)
            print(f"bm {bm:3d} conv{si}: Vd tile {a:7.3f} ms + other tiles {b:7.3f} ms = {a + b:7.3f} ms  "
                  f"{50 * fl / (a + b) / 1e9 / 157.3:.1f}% of the fp32 MFMA peak issued", flush=True)
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
