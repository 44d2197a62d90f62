"""Isolated timing of the F(6,3) conv-stage kernels (TONAL_WINO=6) at the north-star geometry (no model, no LSTM).

    TONAL_WINO=6 python scripts/bench_conv63.py [--batch 256] [--iters 5] [--passes fwd,wgrad,dgrad]
Prints ms, algorithmic TFLOP/s (direct-convolution FLOPs of the valid rows) and the MFMA TFLOP/s issued.
"""
import argparse, os, sys
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
os.environ["TONAL_WINO"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--passes", default="fwd,wgrad,dgrad")
ap.add_argument("--stages", default="2,3")
ap.add_argument("--stamps", action="store_true",
                help="diagnostic build (TONAL_HIP_LIB=build/variants/lib63_stamp1.so): print the in-kernel phase timeline of the NT launches")
args = ap.parse_args()


def stamp_report(eng, tag):
    """Phase timeline of the last wino63v_nt launch from the stamps of a -DV6_STAMP=1 build (wave 0 of every workgroup)."""
    import ctypes as C
    import numpy as np
    fn = getattr(eng.lib, "tl_debug_v6_stamps", None)
    if fn is None:
        print("   (no tl_debug_v6_stamps in this library: not a stamp build)")
        return
    fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
    torch.cuda.synchronize()
    buf = np.zeros((256, 96, 16), dtype=np.uint64)
    assert fn(buf.ctypes.data, 1) == 0
    s = buf.astype(np.int64)
    live = (s[:, :, 7] != 0) & (s[:, :, 0] != 0)
    nt = live.sum(1)
    full = min(int(nt.min()), 96)
    if full < 4:
        print(f"   ({tag}: fewer than 4 stamped tiles per workgroup)")
        return
    t = s[:, 1:full - 1]                                   # drop the first and the last stamped tile of every workgroup
    cyc = (t[:, :, 6] - t[:, :, 1]).astype(np.float64)
    rt = (t[:, :, 7] - t[:, :, 0]).astype(np.float64) * 10e-9      # 100 MHz
    ghz = cyc.sum() / rt.sum() / 1e9
    seg = {"first two K-steps": t[:, :, 2] - t[:, :, 1], "steady-state loop": t[:, :, 3] - t[:, :, 2],
           "last two K-steps + carried MFMAs": t[:, :, 4] - t[:, :, 3], "epilogue instruction stream": t[:, :, 5] - t[:, :, 4],
           "closing wait": t[:, :, 6] - t[:, :, 5], "whole tile": t[:, :, 6] - t[:, :, 1]}
    gap = s[:, 2:full - 1, 1] - s[:, 1:full - 2, 6]
    print(f"   {tag}: {full} stamped tiles per workgroup, in-kernel clock {ghz:.3f} GHz")
    tot = float(seg["whole tile"].mean())
    for k, v in seg.items():
        v = v.astype(np.float64)
        print(f"     {k:36s} {v.mean():10.0f} cycles ({100 * v.mean() / tot:5.1f} %)  p5 {np.percentile(v, 5):9.0f}  p95 {np.percentile(v, 95):9.0f}"
              f"  = {v.mean() / ghz / 1e3:7.2f} us")
    print(f"     {'between tiles (barrier)':36s} {gap.mean():10.0f} cycles")
    # inside the epilogue (slots 8..13, where the epilogue has them): cycles from the end of the K loop
    inner = [(k, (t[:, :, k] - t[:, :, 4]).astype(np.float64)) for k in range(8, 14) if (t[:, :, k] != 0).all()]
    if inner:
        print("     inside the epilogue, cycles after the K loop: " + "  ".join(f"[{k}] {v.mean():.0f}" for k, v in inner)
              + f"  [end] {(t[:, :, 5] - t[:, :, 4]).mean():.0f}")
    # how much in step the workgroups are: spread over the workgroups of the time (100 MHz) their j-th epilogue begins
    e = s[:, 1:full - 1, 0].astype(np.float64) * 0.01      # tile start in us
    ph = e - e.mean(0, keepdims=True)
    print(f"     tile start across workgroups: std {ph.std(0).mean():.1f} us, range {np.ptp(ph, axis=0).mean():.1f} us "
          f"(tile {tot / ghz / 1e3:.1f} us)")
dev = torch.device("cuda:0")
stages_def = [(512, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]
eng = CnnEngine(80, 128, 400, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
assert eng.wino63
B = args.batch
eng._alloc(B, dev)
eng._alloc_bwd()
S = eng.S
g = torch.Generator(device=dev).manual_seed(1)
eng._x = torch.randn(B, 128, 400, device=dev, generator=g)
eng.generation += 1
V1 = eng._v_hex_buffer(eng.V, 1, S * eng.tp1, 512)
V1.normal_(generator=g)
eng._v_ready = {1: V1}
for k in eng.G:
    eng.G[k].normal_(generator=g)
if eng.f63_yprod:         # stage 2's backward operands come out of stage 3's input gradient: give the isolated passes something to read
    for store in (eng.Yt, eng.Vd):
        eng._v_hex_buffer(store, 2, S * eng.tp1, 512).normal_(generator=g)
if eng.gy4:               # ... and stage 3's out of stage 4's
    for store in (eng.Yt, eng.Vd):
        eng._v_hex_buffer(store, 3, S * eng.stages[1].tp_in, 512).normal_(generator=g)
for k in eng.bits:
    eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
    eng.sbits[k].random_(-2**31, 2**31 - 1, generator=g)
for si in [int(s) for s in args.stages.split(",")]:
    st = eng.stages[si - 2]
    if si == 3 and 2 not in eng._v_ready:
        V2 = eng._v_hex_buffer(eng.V, 2, S * st.tp_in, 512)
        V2.normal_(generator=g)
        eng._v_ready[2] = V2
    if si == 4:
        # the one-tap stage behind conv3: only its input gradient runs on the NT63 kernel (six batched GEMMs, epilogue 7)
        assert eng.gy4, "stage 4's input gradient is not on the NT63 kernel (TONAL_KERNELS conv4_dgrad)"
        w = torch.randn(st.cout, st.cin, 1, 1, device=dev, generator=g) * 0.02
        fl = 2.0 * B * eng.C * st.tc * st.cin * st.cout
        eng.stage_dgrad(st, w); torch.cuda.synchronize()
        eng.enable_timers(True)
        for _ in range(args.iters):
            eng.stage_dgrad(st, w)
        ts = eng.timer_summary(); eng.enable_timers(False)
        ms = ts["conv4_dgrad"][1]
        print(f"F63 conv4_dgrad  {ms:8.3f} ms (operand producer + NT63 launch + fix-up)  {fl / ms / 1e9:7.2f} TFLOP/s algorithmic "
              f"({100 * fl / ms / 1e9 / 157.3:.1f}% of fp32 MFMA peak)", flush=True)
        if args.stamps:
            stamp_report(eng, "conv4_dgrad")
        continue
    w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.02
    b = torch.randn(st.cout, device=dev, generator=g) * 0.1
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    fl = 2.0 * B * eng.C * st.tc * 3 * st.cin * st.cout
    iss = eng.f63_issue_factor(st)

    def dgrad():
        eng._vd_ready[st.idx] = eng.generation       # (Vd left by the last weight-gradient launch / by stage 3's input gradient)
        eng.stage_dgrad(st, w)

    def wgrad():
        if (eng.f63_yprod and st.idx == 2) or (eng.gy4 and st.idx == 3):
            eng._y_ready[st.idx] = eng.generation
        eng.stage_wgrad(st, gw, gb)
    for name, fn in (("fwd", lambda: eng.stage_forward(st, w, b)), ("wgrad", wgrad), ("dgrad", dgrad)):
        if name not in args.passes.split(","):
            continue
        if name == "dgrad" and st.idx not in eng.Vd:
            eng.stage_wgrad(st, gw, gb)
        fn(); torch.cuda.synchronize()
        eng.enable_timers(True)
        for _ in range(args.iters):
            fn()
        ts = eng.timer_summary(); eng.enable_timers(False)
        ms = ts[f"conv{si}_{name}"][1]
        print(f"F63 conv{si}_{name:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s algorithmic  {fl * iss / ms / 1e9:7.2f} issued "
              f"({100 * fl * iss / ms / 1e9 / 157.3:.1f}% of fp32 MFMA peak)", flush=True)
        if args.stamps and name in ("fwd", "dgrad"):
            stamp_report(eng, f"conv{si}_{name}")
