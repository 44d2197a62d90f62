#!/usr/bin/env python3
"""20 calls of the time-parallel filtfilt at the C5 size (256 ch x 24 000, float32) - for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TONAL_KERNELS"] = os.environ.get("TONAL_KERNELS", "butter=scan")
import torch
from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
x = torch.randn(256, 24000, device="cuda:0")
for _ in range(20):
    ff.butter_filter(x, [0.3, 100], 400)
torch.cuda.synchronize()
