"""Timing of tl_conv1_fwd_v6 alone at the north-star geometry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import check, ptr
lib = _lib.load()
dev = torch.device("cuda:0")
B, C, T, c1 = 256, 128, 400, 512
S = B * C
tp1, tout1 = 204, 199
x = torch.randn(S, T, device=dev)
w = torch.randn(c1, 3, device=dev); b = torch.randn(c1, device=dev)
nh = S * tp1 // 6
V = torch.zeros((nh + 24 + 127) // 128 * 128, 8, c1, device=dev)
bits = torch.zeros(S * tp1, c1 // 32, dtype=torch.int32, device=dev); sb = torch.zeros_like(bits)
st = torch.cuda.current_stream().cuda_stream
def run():
    check(lib.tl_conv1_fwd_v6(ptr(x), ptr(w), ptr(b), None, ptr(V), ptr(bits), ptr(sb), S, T, 3, c1, tp1, tout1, 0.01, st), "conv1")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
print(f"conv1_fwd_v6 {e0.elapsed_time(e1) / 3:.3f} ms  ({V.numel() * 4 / 1e9:.1f} GB of V)", flush=True)
idx = torch.arange(0, V.numel(), 9973, device=dev)
print(f"  checksum {V.view(-1)[idx].double().mul(torch.arange(idx.numel(), device=dev).double().remainder(17.0) + 1).sum().item():.9e}"
      f" abs {V.view(-1)[idx].double().abs().sum().item():.9e}", flush=True)
if len(sys.argv) > 1:
    # (b) with a large resident footprint, (c) with real model weights / data scale
    big = [torch.zeros(8 * 1024**3 // 4, device=dev) for _ in range(int(sys.argv[1]))]
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    print(f"  with {len(big) * 8} GB more resident: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
    w.mul_(0.01); b.mul_(0.01)
    x.mul_(1e-3)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    print(f"  small weights / inputs: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
    x.zero_()
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    print(f"  zero input: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
