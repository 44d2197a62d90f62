"""Timing of the one-tap NT GEMM launches of the train step in isolation (1x1 stack layer, its input gradient, conv4 / conv5
forward) - for build variants selected with TONAL_HIP_LIB (scripts/build_variant.sh tonal_gemm.hip 'name|-DG_ABL=..|')."""
import os, sys
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._classifier_engine import _launch_nt
from decode_tonal_langauge_amd._lib import LOAD_DIRECT, EPI_STORE, EPI_LRELU, EPI_MASK, EPI_POOL, ptr
lib = _lib.load()
dev = torch.device("cuda:0")
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rows5 = 256 * 128 * 24
for name, M, N, K, epi in (("stack fwd 128->128 LRELU", rows5, 128, 128, EPI_LRELU), ("stack dgrad 128->128 MASK", rows5, 128, 128, EPI_MASK),
                           ("stack fwd 128->64 LRELU", rows5, 64, 128, EPI_LRELU), ("conv5 fwd 256->64 LRELU", rows5, 64, 256, EPI_LRELU),
                           ("conv4 fwd 512->256 STORE", 256 * 128 * 50, 256, 512, EPI_STORE)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev); aux = torch.randn(M, N, device=dev) if epi == EPI_MASK else None
    kw = dict(A=ptr(A), Bw=ptr(W), M=M, A_rows=M, N=N, K=K, lda=K, ldb=K, loader=LOAD_DIRECT, out=ptr(out), ldo=N, epilogue=epi, slope=0.1)
    if epi == EPI_LRELU: kw.update(bias=ptr(b))
    if epi == EPI_MASK: kw.update(aux=ptr(aux), ldaux=N)
    t = ev(lambda: _launch_nt(lib, **kw))
    gb = (M * K + M * N * (2 if epi == EPI_MASK else 1)) * 4 / 1e9
    print(f"{name:28s} {t*1e3:8.1f} us   {gb / t:7.2f} GB/ms (= TB/s) {2.0*M*N*K/t/1e9:7.1f} TFLOP/s", flush=True)
    del A, W, out, aux
