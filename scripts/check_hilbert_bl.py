"""A/B of the two overlap-save forms of the Gaussian Hilbert bank on the C5 array (256 ch x 24 000 samples @ 400 Hz):
tl_hilbert_ols (1024-point inverse per band, workgroup-wide) and tl_hilbert_ols_bl (band-limited: four wave-private
256-point inverses per band).  Same call, interleaved; prints the largest difference and both times.

    python scripts/check_hilbert_bl.py [--iters 20]
"""
import argparse, os, sys
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
C, T, FS = 256, 24000, 400
dev = torch.device("cuda:0")
x64 = torch.from_numpy(np.random.default_rng(0).standard_normal((C, T))).to(dev)
x32 = x64.float()


def run(x, bl, envelope=True):
    os.environ["TONAL_HILBERT_BL"] = "1" if bl else "0"
    return ff.hilbert_filter(x, FS, [70., 150.], envelope=envelope)


def timed(x, bl):
    run(x, bl); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run(x, bl)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


bad = False
for name, x in (("f64", x64), ("f32", x32)):
    for env in (True, False):
        a, b = run(x, False, env), run(x, True, env)
        d = float((a - b).abs().max()) / float(a.abs().max())
        print(f"{name} envelope={env}: max |ols - band-limited| / max |ols| = {d:.3e}", flush=True)
        bad |= not d < 1e-10
for rep in range(2):
    for name, x in (("f64", x64), ("f32", x32)):
        print(f"{name}: ols {timed(x, False):.4f} ms   band-limited {timed(x, True):.4f} ms", flush=True)

# cost split: t(nb) = (window load + forward transform + output) + nb x (band); straight through the C ABI
if os.environ.get("TONAL_HILBERT_SPLIT", "1") != "0":
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import check, ptr
    cfs, sds = ff.gaussian_bank([70., 150.], FS, 0.018, 1 / 7, np.log10(0.39), 0.5)
    tp, ntap, half, sym, ols = ff._device_taps(T, FS, cfs, sds, dev)
    y = torch.empty(C, T, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for nb in (1, 2, 4, 7):
        def call():
            check(_lib.load().tl_hilbert_ols_bl(ptr(x32), 0, ptr(ols[3][0]), ptr(ols[3][1]), ptr(ols[1]), ptr(y), C, T, nb, half,
                                                ols[2], 1, st), "tl_hilbert_ols_bl")
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            call()
        e1.record(); torch.cuda.synchronize()
        print(f"nb = {nb}: {e0.elapsed_time(e1) / args.iters:.4f} ms", flush=True)
print("FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
