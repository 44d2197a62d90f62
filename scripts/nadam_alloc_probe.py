import sys
sys.path.insert(0, "/root/repo")
import torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import check, ptr
lib = _lib.load(); dev = torch.device("cuda:0")
R, Cc, kr, U = 73728, 18432, 32, 8
n = R * Cc
fa = torch.randn(kr, R, device=dev) * 1e-3; fb = torch.randn(kr, Cc, device=dev)
st = torch.cuda.current_stream().cuda_stream
args = (5e-5, 4e-4, 0.9, 0.999, 1e-3, 1e-8, 0.004, 1.0)
slab = torch.empty(144, U, Cc, device=dev)
def timed(p, m, v, it=5):
    fn = lambda: check(lib.tl_nadam_lowrank_dh(ptr(p), ptr(m), ptr(v), ptr(fa), ptr(fb), kr, R, Cc, R, Cc, *args, ptr(slab), U, 16, st), "dh")
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def seg(t):
    for s in torch.cuda.memory_snapshot():
        if s["address"] <= t.data_ptr() < s["address"] + s["total_size"]:
            return (hex(s["address"]), s["total_size"] >> 20)
p = torch.zeros(R, Cc, device=dev)
mv = torch.zeros(2 * n, device=dev)
m, v = mv[:n].view(R, Cc), mv[n:].view(R, Cc)
print("p separate, m + v in one allocation:", f"{timed(p, m, v):.3f} ms", seg(p), seg(m), seg(v))
del m, v, mv
junk = torch.zeros(3 * n, device=dev); del junk            # a 16 GB block goes back to the cache
m = torch.zeros(R, Cc, device=dev); v = torch.zeros(R, Cc, device=dev)
print("m, v carved from a cached 16 GB block:", f"{timed(p, m, v):.3f} ms", seg(p), seg(m), seg(v))
del m, v
torch.cuda.empty_cache()
m = torch.zeros(R, Cc, device=dev); v = torch.zeros(R, Cc, device=dev)
print("after empty_cache (fresh allocations):", f"{timed(p, m, v):.3f} ms", seg(p), seg(m), seg(v))
