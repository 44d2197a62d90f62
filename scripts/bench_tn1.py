"""Timing of the weight-gradient launches of the 1x1 stack / conv5 (tl_gemm_tn_window, direct loader, split-K slabs) in isolation."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import TnParams, LOAD_DIRECT, check, ptr
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rows = 256 * 128 * 24
for name, M, N in (("stack 128 x 128", 128, 128), ("stack 72 x 128", 72, 128), ("stack 128 x 64", 128, 64), ("conv5 256 x 64", 256, 64)):
    A = torch.randn(rows, M, device=dev); B = torch.randn(rows, N, device=dev)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    for target in (2048, 1024, 512):
        sk = int(max(1, min((rows + 31) // 32, (target + tiles - 1) // tiles, 1024)))
        slab = torch.empty(sk, M, N, device=dev)
        p = TnParams()
        p.splitk, p.J, p.Tp, p.Tvalid = sk, 1, 24, 24
        p.A, p.B, p.slab = ptr(A), ptr(B), ptr(slab)
        p.Krows, p.A_rows, p.B_rows, p.Mdim, p.Ndim, p.lda, p.ldb, p.ldc, p.loader, p.slab_stride = rows, rows, rows, M, N, M, N, N, LOAD_DIRECT, M * N
        t = ev(lambda: check(lib.tl_gemm_tn_window(C.byref(p), st), "tn"))
        gb = rows * (M + N) * 4 / 1e9
        print(f"{name:18s} splitk {sk:5d}  {t*1e3:8.1f} us   {gb / t:6.2f} TB/s   {2.0*rows*M*N/t/1e9:7.1f} TFLOP/s", flush=True)
