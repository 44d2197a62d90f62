import torch, time
dev = torch.device("cuda:0")
W = torch.randn(73728, 18432, device=dev)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = W.numel() * 4 / 1e9
v = torch.randn(18432, device=dev); h = torch.randn(8, 18432, device=dev); g = torch.randn(8, 73728, device=dev)
for name, fn in (("sum", lambda: W.sum()), ("abs().max", lambda: W.abs().max()), ("mv W@v", lambda: torch.mv(W, v)),
                 ("mm h@W^T (8 rows)", lambda: h @ W.t()), ("mm g@W (8 rows)", lambda: g @ W), ("clone (r+w)", lambda: W.clone())):
    ms = timed(fn)
    print(f"{name:22s} {ms:7.3f} ms  {gb / ms:6.2f} TB/s read" + ("  (x2 bytes moved)" if "clone" in name else ""))
