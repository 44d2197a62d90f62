#!/usr/bin/env python3
"""Block length of the time-parallel filtfilt at the C5 size: HIP-event time per call for several L (module constant _SCAN_L)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TONAL_KERNELS"] = "butter=scan"
import torch
import bench
from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
x = torch.randn(256, 24000, device="cuda:0")
os.environ.pop("TONAL_KERNELS")
seq = bench.event_ms(lambda: ff.butter_filter(x, [0.3, 100], 400), 10)
os.environ["TONAL_KERNELS"] = "butter=scan"
print(f"sequential {seq:.4f} ms")
for L in (64, 96, 128, 160, 192, 256):
    ff._SCAN_L = L
    ff._BUTTER_CACHE.clear()
    ms = bench.event_ms(lambda: ff.butter_filter(x, [0.3, 100], 400), 20)
    print(f"L = {L:4d}  {ms:.4f} ms  {seq / ms:5.1f} x")
