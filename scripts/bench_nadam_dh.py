#!/usr/bin/env python3
"""W_hh update at the north-star shape (73 728 x 18 432, factor rank 32, U = 8): tl_nadam_lowrank + the separate dh_1 pass
(tl_gemm_tn_window, skinny form) against tl_nadam_lowrank_dh for several row-tile counts.  HIP events, 5 launches each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import check, ptr

lib = _lib.load()
dev = torch.device("cuda:0")
R, Cc, kr, U = 73728, 18432, 32, 8
p = torch.randn(R, Cc, device=dev) * 0.01
m = torch.zeros_like(p)
v = torch.zeros_like(p)
fa = torch.randn(kr, R, device=dev) * 1e-3
fb = torch.randn(kr, Cc, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


args = (5e-5, 4e-4, 0.9, 0.999, 1e-3, 1e-8, 0.004, 1.0)
base = timed(lambda: check(lib.tl_nadam_lowrank(ptr(p), ptr(m), ptr(v), ptr(fa), ptr(fb), kr, R, Cc, R, Cc, *args, st), "nadam"))
print(f"tl_nadam_lowrank            {base:7.3f} ms  ({6 * 4 * R * Cc / base / 1e9:6.2f} TB/s)")
for rt in (2, 4, 8, 16, 32, 64):
    nslab = -(-(R // 32) // rt)
    slab = torch.empty(nslab, U, Cc, device=dev)
    t = timed(lambda: check(lib.tl_nadam_lowrank_dh(ptr(p), ptr(m), ptr(v), ptr(fa), ptr(fb), kr, R, Cc, R, Cc, *args, ptr(slab), U, rt, st), "dh"))
    red = torch.empty(U, Cc, device=dev)
    t2 = timed(lambda: torch.sum(slab, dim=0, out=red))
    print(f"tl_nadam_lowrank_dh rt={rt:3d}  {t:7.3f} ms  + slab sum ({nslab} slabs, {slab.numel() * 4 / 1e6:.0f} MB) {t2:6.3f} ms")
