#!/usr/bin/env python3
"""What a pure store stream reaches on this chip (the yardstick for conv1's 18.3 GB of V1): torch fill / copy on an 18 GB buffer."""
import torch
dev = torch.device("cuda:0")
n = 18_300_000_000 // 4
a = torch.empty(n, device=dev)
def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
gb = n * 4 / 1e9
t = timed(lambda: a.fill_(1.0)); print(f"fill_            {t:7.3f} ms  {gb / t:5.2f} TB/s written")
t = timed(lambda: a.zero_()); print(f"zero_            {t:7.3f} ms  {gb / t:5.2f} TB/s written")
b = torch.empty(n // 2, device=dev)
t = timed(lambda: b.copy_(a[: n // 2])); print(f"copy (r + w)     {t:7.3f} ms  {gb / t:5.2f} TB/s moved")
t = timed(lambda: torch.mul(a[: n // 2], 2.0, out=b)); print(f"mul out (r + w)  {t:7.3f} ms  {gb / t:5.2f} TB/s moved")
