"""Signal-path measurement (SURVEY.md section 8d, C5 shape): frequency_filter on 256 ch x 24 000
samples float32 @ 400 Hz, resident in HBM.  Prints one JSON line per method with kernel time
(HIP events on the launch stream), algorithmic GB/s = C*T*(s_in + s_out) / t against the HBM peak,
and the CPU oracle timed on the same array."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
from oracle import signal_oracle as sg

C, T, FS = 256, 24000, 400
dev = torch.device("cuda:0")
x_np = np.random.default_rng(0).standard_normal((C, T)).astype(np.float32)
x = torch.from_numpy(x_np).to(dev)


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, y


cases = [
    ("hilbert 70-150 Hz envelope (8 Gaussian bands)", lambda: ff.hilbert_filter(x, FS, [70., 150.]),
     lambda: sg.hilbert_filter(x_np, FS, [70., 150.]), 8, 1e-5),
    ("butter order-4 band-pass filtfilt 0.3-100 Hz", lambda: ff.butter_filter(x, [0.3, 100], FS),
     lambda: sg.butter_filter(x_np[:16], [0.3, 100], FS), 8, 1e-6),
    ("fir order-390 band-pass bank [100 Hz]", lambda: ff.fir_bandpass_filter(x, FS, 390, [100.]),
     lambda: sg.fir_bandpass_filter(x_np[:32], FS, 390, [100.]), 4, 1e-5),
]
for name, gpu_fn, cpu_fn, s_out, tol in cases:
    ms, y = timed(gpu_fn)
    t0 = time.perf_counter(); ref = cpu_fn(); cpu_s = time.perf_counter() - t0
    rows = ref.shape[0]
    err = float(np.max(np.abs(y[:rows].double().cpu().numpy() - ref)) / np.max(np.abs(ref)))
    gb = C * T * (4 + s_out) / 1e9
    print(json.dumps({"op": name, "shape": [C, T], "ms": round(ms, 4), "algorithmic_GBps": round(gb / ms * 1e3, 1),
                      "hbm_peak_GBps": 8000, "frac_of_hbm_peak": round(gb / ms * 1e3 / 8000, 4),
                      "channel_samples_per_s": round(C * T / ms * 1e3), "max_rel_err_vs_oracle": err,
                      "cpu_oracle_s_for_rows": [rows, round(cpu_s, 3)],
                      "cpu_channel_samples_per_s": round(rows * T / cpu_s)}), flush=True)
    assert err < tol, (name, err)
