"""Per-tile fixed cost of the Winograd NT kernels: time the stage-2 forward / input gradient at K = 512
and K = 256 (same rows, same columns); T(K) = tiles * (steps(K) * t_step + fixed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine
dev = torch.device("cuda:0")
B = 256
res = {}
for c1 in (512, 256):
    defs = [(c1, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]
    eng = CnnEngine(80, 128, 400, 6, 64, 0.0, 0.01, defs, [128, 128, 128, 128, 64])
    eng.wino_vout = False          # stage kernels one at a time: every stage reads P
    eng.fuse_c1 = False
    eng._alloc(B, dev); eng._alloc_bwd()
    g = torch.Generator(device=dev).manual_seed(1)
    eng.P[1].normal_(generator=g); eng.G[2].normal_(generator=g)
    eng.bits[2].random_(-2**31, 2**31 - 1, generator=g); eng.sbits[1].random_(-2**31, 2**31 - 1, generator=g)
    st = eng.stages[0]
    w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.02
    b = torch.randn(st.cout, device=dev, generator=g) * 0.1
    for name, fn in (("fwd", lambda: eng.stage_forward(st, w, b)), ("dgrad", lambda: eng.stage_dgrad(st, w))):
        fn(); torch.cuda.synchronize()
        eng.enable_timers(True)
        for _ in range(4):
            fn()
        ms = eng.timer_summary()[f"conv2_{name}"][1]; eng.enable_timers(False)
        res[(name, c1)] = ms
        print(name, "K(fwd)/N(dgrad) =", c1, f"{ms:.3f} ms", flush=True)
    del eng
    torch.cuda.empty_cache()
tiles = 12800 * 8 / 256      # rounds of 256 resident workgroups
t512, t256 = res[("fwd", 512)], res[("fwd", 256)]
fixed = (2 * t256 - t512) / tiles * 1e3
print(f"forward: per-round fixed cost {fixed:.2f} us of {t512 / tiles * 1e3:.1f} us per round ({100 * fixed * tiles / 1e3 / t512:.1f} %)")
