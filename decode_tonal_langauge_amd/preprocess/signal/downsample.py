"""FFT down-sampling step (mirror of reference preprocess/signal/downsample.py:6-29): resample the
whole recording to ``downsample_freq`` (default 400 Hz) with ``scipy.signal.resample`` semantics and
update ``params.signal_freq``.

The DFT of the (arbitrary) recording length is evaluated on the GPU with Bluestein's chirp-z identity
over power-of-two Stockham FFTs in fp64 (``tl_fft_resample``); the chirps, the chirp-filter spectra
and the twiddle table are coefficient data built once per length on the host and cached."""
from argparse import Namespace

import numpy as np
import torch

from ... import _lib
from ..._lib import check, ptr
from ._common import ret, stream, to_device

_COEF_CACHE = {}


def _bluestein_coeffs(n: int, dev):
    key = (n, str(dev))
    hit = _COEF_CACHE.get(key)
    if hit is None:
        m2 = 1 << int(np.ceil(np.log2(max(2 * n - 1, 2))))
        m = np.arange(n, dtype=np.int64)
        ang = np.pi * ((m * m) % (2 * n)).astype(np.float64) / n          # exact reduction of m^2 mod 2n
        w = np.exp(1j * ang)
        b = np.zeros(m2, dtype=np.complex128)
        b[:n] = w
        b[m2 - n + 1:] = w[1:][::-1]
        bf = np.fft.fft(b)
        tw = np.exp(-2j * np.pi * np.arange(m2 // 2) / m2)
        pack = lambda z: torch.from_numpy(np.ascontiguousarray(np.stack([z.real, z.imag], axis=-1))).to(dev)
        if len(_COEF_CACHE) > 16:
            _COEF_CACHE.clear()
        hit = _COEF_CACHE[key] = (pack(w), pack(bf), pack(tw), m2)
    return hit


def resample(data, num: int):
    """``scipy.signal.resample(data, num, axis=1)`` for real (C, T) data; output dtype = input dtype."""
    x, was_np = to_device(data, "downsample")
    C, nx = x.shape
    if num < 1:
        raise ValueError("downsample: the target number of samples must be positive")
    w1, bf1, tw1, m2a = _bluestein_coeffs(nx, x.device)
    w2, bf2, tw2, m2b = _bluestein_coeffs(num, x.device)
    work = torch.empty(C * (2 * max(m2a, m2b) + m2b), 2, dtype=torch.float64, device=x.device)
    y = torch.empty(C, num, dtype=x.dtype, device=x.device)
    check(_lib.load().tl_fft_resample(ptr(x), int(x.dtype == torch.float64), ptr(y), C, nx, num, ptr(w1), ptr(bf1),
                                      ptr(tw1), m2a, ptr(w2), ptr(bf2), ptr(tw2), m2b, ptr(work), stream()),
          "tl_fft_resample")
    return ret(y, was_np)


def run(data, params: Namespace):
    target_freq = getattr(params, "downsample_freq", 400)
    factor = target_freq / params.signal_freq
    n_samples = int(data.shape[1] * factor)
    out = resample(data, n_samples)
    params.signal_freq = target_freq
    return out
