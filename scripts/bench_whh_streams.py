#!/usr/bin/env python3
"""The recurrent products of the label LSTM at the north-star shape (W_hh 73 728 x 18 432, U = 8): the streaming kernel of round 6
(tl_lstm_gw: dgates . W) for several row-block sizes, and a library GEMM on the same operands for both products as the yardstick
(the MFMA forms: 1.03 ms forward, 1.05 ms backward, profiles/r06_c3_step_summary.md).  HIP events, 10 launches each."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import check, ptr

lib = _lib.load()
dev = torch.device("cuda:0")
N, K, U = 73728, 18432, 8
W = torch.randn(N, K, device=dev) * 0.01
h = torch.randn(U, K, device=dev)
g = torch.randn(U, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
gb = N * K * 4 / 1e9


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t = timed(lambda: h @ W.t())
print(f"library GEMM  h @ W^T               {t:7.3f} ms  {gb / t:5.2f} TB/s")
ref_b = (g.double() @ W.double())
for rpb in (256, 512, 1024):
    nb = -(-N // rpb)
    slab = torch.empty(nb, U, K, device=dev)
    t = timed(lambda: check(lib.tl_lstm_gw(ptr(g), ptr(W), ptr(slab), U, N, K, N, K, rpb, st), "gw"))
    red = torch.empty(U, K, device=dev)
    t2 = timed(lambda: torch.sum(slab, dim=0, out=red))
    err = float((slab.double().sum(0) - ref_b).abs().max() / ref_b.abs().max())
    print(f"tl_lstm_gw rows/block {rpb:5d}         {t:7.3f} ms  {gb / t:5.2f} TB/s  + slab sum {t2:.3f} ms ({slab.numel() * 4 / 1e6:.0f} MB)  err {err:.1e}")
t = timed(lambda: g @ W)
print(f"library GEMM  g @ W                 {t:7.3f} ms  {gb / t:5.2f} TB/s")
