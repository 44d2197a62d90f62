"""Band extraction step on MI355X (mirror of reference preprocess/signal/frequency_filter.py).

Same plugin ABI: ``run(data (C,T), params) -> (C', T)`` with ``params.bands`` / ``params.signal_freq``
(:9-77) and the same three public functions with the same keyword arguments and error behaviour:
``hilbert_filter`` (:80-184), ``butter_filter`` (:187-229), ``fir_bandpass_filter`` (:232-274).

The arithmetic runs in HIP kernels (``csrc/tonal_signal.hip``):
 * the Gaussian-bank analytic signal is evaluated - when its band kernels are short, as for high gamma on the
   400 Hz recordings of the pipeline - as an *exact* circular convolution: the host
   takes the inverse DFT of the reference's frequency-domain kernel ``H_b * hilbert_mult``
   (including its ``H[0] = 0`` and the one-sided multiplier) and keeps every tap above the fp64
   round-off floor of that inverse DFT (1e-13 of the kernel peak); the kernel then computes |sum_n h_b[n] x[(t-n) mod T]| per band and the mean
   over bands in one pass over the recording (the reference loops n_bands x C Python-level iFFTs); bands whose
   kernels are longer than that kernel's LDS window (low bands at a raw recording rate: sigma_t of hundreds to
   thousands of samples) go through the DFT domain exactly as the reference writes it - forward DFT, per-band
   multiplier, inverse DFT - with the arbitrary-length DFT as a Bluestein chirp-z over power-of-two Stockham
   passes (``tl_hilbert_fft``; ``TONAL_HILBERT=fft`` forces that path, ``=taps`` forbids it).  Eight-band banks with
   truncated kernels of up to 513 taps - the high-gamma bank of the pipeline - take the fastest form of the same
   convolution: overlap-save on LDS-resident fp64 FFTs (the spectra of the SAME truncated kernels) - ``tl_hilbert_ols_bl``
   when every band's kernel spectrum fits a window of 256 of the 1024 bins (the Gaussian bank does: four wave-private
   256-point inverses per band), ``tl_hilbert_ols`` with all bins otherwise or under ``TONAL_HILBERT_BL=0``;
   ``TONAL_HILBERT=sym`` keeps the time-domain kernel, which uses the kernels' Hermitian symmetry;
 * ``filtfilt`` / ``sosfilt`` are fp64 direct-form-II-transposed recurrences, sequential in time exactly like scipy's loop
   (``filtfilt``: the state spread over the lanes of a DPP row per channel, bit-identical; ``sosfilt``: one lane per channel);
 * the FIR bank is a causal convolution with zero initial state (recordings of >= 1024 samples: overlap-save on the same
   LDS FFT as the Hilbert bank, ``tl_fir_bank_ols``).
Filter *design* (``butter``, ``lfilter_zi``, ``firwin``) stays on scipy: coefficient generation,
not the hot path - done once per filter, the device copies are cached.

Inputs may be NumPy arrays (copied to the GPU and back, like a drop-in step must) or CUDA torch
tensors (stay resident; a CUDA tensor is returned).
"""
from __future__ import annotations

import math
import os
from argparse import Namespace
from typing import List, Tuple, Union

import numpy as np
import torch
from scipy.signal import butter, firwin, lfilter_zi

from ... import _kernels, _lib
from ..._lib import check, ptr

_MAX_TAPS_LDS = 7169      # (1024 + ntap - 1) * 8 B <= 64 KiB
_OLS_N = 1024             # FFT length of the overlap-save path (tl_hilbert_ols)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("frequency_filter (MI355X build): no GPU visible; this package has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_device(data) -> Tuple[torch.Tensor, bool]:
    """(C,T) float32/float64 device tensor and whether the caller passed a NumPy array."""
    if isinstance(data, torch.Tensor):
        _lib.require_gpu(data, "frequency_filter")
        t = data
        was_np = False
    else:
        arr = np.asarray(data)
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(_device())
        was_np = True
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    if t.dim() != 2:
        raise ValueError("expected data of shape (n_channels, n_timepoints)")
    return t.contiguous(), was_np


def _ret(t: torch.Tensor, was_np: bool):
    return t.cpu().numpy() if was_np else t


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gaussian_bank(freq_ranges, sampling_rate, f0=0.018, octspace=1 / 7, filterbank_bias=math.log10(0.39),
                  filterbank_slope=0.5):
    """Centre frequencies / widths, including the reference's argument normalisation (:121-153)."""
    if isinstance(freq_ranges, tuple):
        freq_ranges = [freq_ranges]
    if isinstance(freq_ranges[0], float):
        freq_ranges = [tuple(freq_ranges)]
    cfs, sds = [], []
    for fr in freq_ranges:
        if len(fr) != 2:
            raise ValueError("Each frequency range must be a tuple of (min_freq, max_freq).")
        min_freq = fr[0] if fr else 0
        max_freq = fr[1] if fr else sampling_rate // 2
        max_oct = math.log2(max_freq / f0)
        f = f0
        while math.log2(f / f0) < max_oct:
            if f >= min_freq:
                cfs.append(f)
                sds.append(10 ** (filterbank_bias + filterbank_slope * math.log10(f)))
            f = f * (2 ** octspace)
    return np.array(cfs), np.array(sds) * np.sqrt(2)


def band_multipliers(T: int, sampling_rate: float, cfs: np.ndarray, sds: np.ndarray) -> np.ndarray:
    """(nb, T) real DFT-domain kernels H_b x analytic multiplier of the reference (:155-175)."""
    freqs = np.fft.fftfreq(T, d=1.0 / sampling_rate)
    mult = np.zeros(T)
    if T % 2 == 0:
        mult[0] = 1
        mult[1:T // 2] = 2
        mult[T // 2] = 1
    else:
        mult[0] = 1
        mult[1:(T + 1) // 2] = 2
    ker = np.empty((len(cfs), T))
    for i, (fc, sf) in enumerate(zip(cfs, sds)):
        H = np.exp(-0.5 * ((freqs - fc) / sf) ** 2)
        H[0] = 0
        ker[i] = H * mult
    return ker


def analytic_taps(T: int, sampling_rate: float, cfs: np.ndarray, sds: np.ndarray, tol: float = 1e-13):
    """Time-domain kernels of the reference's per-band DFT multiplier (:155-175).

    Returns (taps (nb, ntap) complex128, half) with h_b[n], n = k - half.  ``ntap == T, half == 0``
    means the full circular kernel is used (short recordings / slowly decaying kernels)."""
    freqs = np.fft.fftfreq(T, d=1.0 / sampling_rate)
    mult = np.zeros(T)
    if T % 2 == 0:
        mult[0] = 1
        mult[1:T // 2] = 2
        mult[T // 2] = 1
    else:
        mult[0] = 1
        mult[1:(T + 1) // 2] = 2
    ker = np.empty((len(cfs), T), dtype=np.complex128)
    for i, (fc, sf) in enumerate(zip(cfs, sds)):
        H = np.exp(-0.5 * ((freqs - fc) / sf) ** 2)
        H[0] = 0
        ker[i] = np.fft.ifft(H * mult)
    # Keep every tap above the round-off floor of the DFT-domain evaluation itself: entries below
    # `tol` x the kernel peak are indistinguishable from the reference's own fp64 FFT rounding.
    mag = np.abs(ker)
    dist = np.minimum(np.arange(T), T - np.arange(T))      # circular distance of index n from 0
    live = mag > tol * mag.max(axis=1, keepdims=True)
    half = int(dist[live.any(axis=0)].max())
    if 2 * half + 1 >= T:
        return ker, 0
    idx = np.arange(-half, half + 1) % T
    return np.ascontiguousarray(ker[:, idx]), half


_TAPS_CACHE = {}
_TW_CACHE = {}


def _ols_twiddles(dev):
    """(cos, -sin)(2 pi m / N), m < N, of the overlap-save transforms."""
    hit = _TW_CACHE.get(str(dev))
    if hit is None:
        ang = 2.0 * np.pi * np.arange(_OLS_N) / _OLS_N
        hit = _TW_CACHE[str(dev)] = torch.from_numpy(np.ascontiguousarray(np.stack([np.cos(ang), -np.sin(ang)], axis=-1))).to(dev)
    return hit


def _band_limited(G: np.ndarray, dev, tol: float = 1e-10):
    """Windows of nfft / 4 bins that hold each band's kernel spectrum (tl_hilbert_ols_bl): (Gp (nb, 4, nfft / 4, 2), k0 (nb))
    on the device, or None when some band leaves more than ``tol`` x its peak (summed magnitude) outside its window - the
    error that dropping those bins adds to a unit-scale output is of that order, two decades inside the 1e-9 the golden holds."""
    nb, nfft = G.shape
    q = nfft // 4
    mag = np.abs(G)
    k0 = ((mag.argmax(axis=1) - q // 2) & ~3) % nfft                         # multiples of 4: the kernel's LDS reads stay aligned
    idx = (k0[:, None] + np.arange(q)[None, :]) % nfft                      # (nb, q)
    win = np.take_along_axis(G, idx, axis=1)
    if np.any(mag.sum(axis=1) - np.abs(win).sum(axis=1) > tol * mag.max(axis=1)):
        return None
    ph = np.exp(2j * np.pi * np.arange(q)[None, :] * np.arange(4)[:, None] / nfft)   # (4, q): residue r's twiddle
    Gp = win[:, None, :] * ph[None, :, :]
    return (torch.from_numpy(np.ascontiguousarray(np.stack([Gp.real, Gp.imag], axis=-1))).to(dev),
            torch.from_numpy(k0.astype(np.int32)).to(dev))


def _device_taps(T, sampling_rate, cfs, sds, dev):
    """Device copy of the band kernels, cached per (length, rate, bank): the host-side inverse DFT
    is coefficient generation and must not sit in front of every call."""
    key = (int(T), float(sampling_rate), cfs.tobytes(), sds.tobytes(), str(dev))
    hit = _TAPS_CACHE.get(key)
    if hit is None:
        taps, half = analytic_taps(T, sampling_rate, cfs, sds)
        tp = torch.from_numpy(np.ascontiguousarray(np.stack([taps.real, taps.imag], axis=-1))).to(dev)
        sym = None
        if half > 0 and taps.shape[0] == 8 and taps.shape[1] == 2 * half + 1:
            # the DFT multiplier is real, so h[-n] = conj(h[n]) up to the rounding of the inverse DFT (1e-17): the
            # Hermitian part, n = 0..half, for tl_gauss_envelope_sym (0.56 x the fp64 operations)
            fw, bw = taps[:, half:], taps[:, half::-1]
            h = (0.5 * (fw + np.conj(bw))).T                # (half + 1, 8): tap-major
            sym = torch.from_numpy(np.ascontiguousarray(np.stack([h.real, h.imag], axis=-1))).to(dev)
        ols = None
        nfft = _OLS_N
        if 0 < half <= nfft // 4 and taps.shape[0] == 8 and taps.shape[1] == 2 * half + 1 and T >= nfft:
            # overlap-save on an LDS-resident FFT (tl_hilbert_ols): the spectra of the same truncated kernels, / N, and the
            # twiddle table (cos, -sin)
            g = np.zeros((8, nfft), dtype=np.complex128)
            g[:, :2 * half + 1] = taps
            G = np.fft.fft(g, axis=1) / nfft
            ols = (torch.from_numpy(np.ascontiguousarray(np.stack([G.real, G.imag], axis=-1))).to(dev), _ols_twiddles(dev), nfft,
                   _band_limited(G, dev))
        if len(_TAPS_CACHE) > 32:
            _TAPS_CACHE.clear()
        hit = _TAPS_CACHE[key] = (tp, taps.shape[1], half, sym, ols)
    return hit


_MULT_CACHE = {}


def _hilbert_dft(x: torch.Tensor, sampling_rate, cfs, sds, envelope: bool) -> torch.Tensor:
    """The bank in the DFT domain (``tl_hilbert_fft``), channels in chunks that bound the workspace to ~4 GB."""
    from .downsample import _bluestein_coeffs
    C, T = x.shape
    dev = x.device
    w, bf, tw, m2 = _bluestein_coeffs(T, dev)
    key = (int(T), float(sampling_rate), cfs.tobytes(), sds.tobytes(), str(dev))
    kd = _MULT_CACHE.get(key)
    if kd is None:
        if len(_MULT_CACHE) > 8:
            _MULT_CACHE.clear()
        kd = _MULT_CACHE[key] = torch.from_numpy(band_multipliers(T, sampling_rate, cfs, sds)).to(dev)
    y = torch.empty(C, T, dtype=torch.float64, device=dev)
    per_ch = (2 * m2 + T) * 16
    chunk = int(max(1, min(C, (4 << 30) // per_ch)))
    work = torch.empty(chunk * (2 * m2 + T), 2, dtype=torch.float64, device=dev)
    lib = _lib.load()
    for c0 in range(0, C, chunk):
        n = min(chunk, C - c0)
        check(lib.tl_hilbert_fft(ptr(x[c0:c0 + n]), int(x.dtype == torch.float64), ptr(y[c0:c0 + n]), n, T, ptr(kd),
                                 len(cfs), ptr(w), ptr(bf), ptr(tw), m2, int(envelope), ptr(work), _stream()),
              "tl_hilbert_fft")
    return y


def hilbert_filter(data, sampling_rate: int, freq_ranges: Union[List[Tuple[float, float]], Tuple[float, float]],
                   f0: float = 0.018, octspace: float = 1 / 7, filterbank_bias: float = math.log10(0.39),
                   filterbank_slope: float = 0.5, envelope: bool = True):
    """Gaussian Hilbert filter bank, mean over bands; float64 out (reference :80-184)."""
    cfs, sds = gaussian_bank(freq_ranges, sampling_rate, f0, octspace, filterbank_bias, filterbank_slope)
    x, was_np = _to_device(data)
    C, T = x.shape
    if len(cfs) == 0:
        # the reference's mean over an empty band axis yields NaN
        return _ret(torch.full((C, T), float("nan"), dtype=torch.float64, device=x.device), was_np)
    mode = _kernels.get("hilbert")
    tp, ntap, half, sym, ols = (None, 0, 0, None, None) if mode == "fft" else _device_taps(T, sampling_rate, cfs, sds, x.device)
    if mode == "fft" or ntap > _MAX_TAPS_LDS:
        if mode in ("taps", "sym"):
            raise ValueError(f"hilbert_filter: the band kernels need {ntap} taps at this sampling rate; the time-domain "
                             f"kernel supports up to {_MAX_TAPS_LDS} (TONAL_KERNELS hilbert={mode} forbids the DFT-domain path)")
        return _ret(_hilbert_dft(x, sampling_rate, cfs, sds, bool(envelope)), was_np)
    y = torch.empty(C, T, dtype=torch.float64, device=x.device)
    # hilbert = auto: the first form below that the bank / recording allows; ols, ols_full, sym, taps force one
    if ols is not None and ols[3] is not None and mode in ("auto", "ols"):
        # band-limited overlap-save.  float32 recordings: fp64 math on the fp32 samples by default.  In the reference only
        # the forward transform stays single precision (scipy.fft(float32) -> complex64, frequency_filter.py:167); the product
        # with the float64 kernel promotes to complex128 and the inverse transform, |.| and the band mean run in fp64
        # (:170-184).  fp64 throughout is at least that precise.  hilbert_f32=1 opts into fp32 transforms end to end (0.135
        # instead of ~0.16 ms at 256 x 24 000; ~1e-6 relative, golden G6 hilbert_f32_odd holds it to 1e-5)
        mode_x = 1 if x.dtype == torch.float64 else (0 if _kernels.get("hilbert_f32") == "1" else 2)
        check(_lib.load().tl_hilbert_ols_bl(ptr(x), mode_x, ptr(ols[3][0]), ptr(ols[3][1]), ptr(ols[1]), ptr(y),
                                            C, T, len(cfs), half, ols[2], int(bool(envelope)), _stream()), "tl_hilbert_ols_bl")
        return _ret(y, was_np)
    if ols is not None and mode in ("auto", "ols", "ols_full"):          # overlap-save with all 1024 bins per band
        check(_lib.load().tl_hilbert_ols(ptr(x), int(x.dtype == torch.float64), ptr(ols[0]), ptr(ols[1]), ptr(y), C, T, len(cfs),
                                         half, ols[2], int(bool(envelope)), _stream()), "tl_hilbert_ols")
        return _ret(y, was_np)
    if sym is not None and mode != "taps":                               # Hermitian-symmetric time-domain kernel
        check(_lib.load().tl_gauss_envelope_sym(ptr(x), int(x.dtype == torch.float64), ptr(sym), ptr(y), C, T, len(cfs), half,
                                                int(bool(envelope)), _stream()), "tl_gauss_envelope_sym")
        return _ret(y, was_np)
    check(_lib.load().tl_gauss_envelope(ptr(x), int(x.dtype == torch.float64), ptr(tp), ptr(y), C, T, len(cfs), ntap,
                                        half, int(bool(envelope)), _stream()), "tl_gauss_envelope")
    return _ret(y, was_np)


def butter_filter(data, freqs: Union[Tuple[float, float], float], fs: float, order: int = 4, causal: bool = False,
                  filter_type: str = 'bandpass'):
    """Butterworth filter: zero-phase ``filtfilt`` or causal ``sosfilt`` (reference :187-229)."""
    nyquist = 0.5 * fs
    wn = np.asarray(freqs, dtype=float) / nyquist
    squeeze = False
    if not isinstance(data, torch.Tensor) and np.asarray(data).ndim == 1:
        data = np.asarray(data)[None, :]
        squeeze = True
    x, was_np = _to_device(data)
    C, T = x.shape
    lib = _lib.load()
    y = torch.empty(C, T, dtype=torch.float64, device=x.device)
    key = (int(order), tuple(np.atleast_1d(wn).tolist()), str(filter_type), bool(causal), str(x.device), _kernels.get("butter"))
    coef = _BUTTER_CACHE.get(key)
    if coef is None:
        # coefficient design (scipy, as the reference calls it) and its host-to-device copies, once per filter
        if causal:
            sos = np.ascontiguousarray(butter(order, wn, btype=filter_type, output='sos'), dtype=np.float64)
            coef = (torch.from_numpy(sos).to(x.device), sos.shape[0])
        else:
            b, a = butter(order, wn, btype=filter_type)
            ntaps = max(len(a), len(b))
            bb = np.zeros(ntaps)
            aa = np.zeros(ntaps)
            bb[:len(b)] = b / a[0]
            aa[:len(a)] = a / a[0]
            zi = lfilter_zi(bb, aa)
            coef = tuple(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(x.device) for v in (bb, aa, zi)) + (ntaps,)
            if _kernels.get("butter") == "scan" and ntaps <= 9:
                coef = coef + (torch.from_numpy(_scan_matrices(aa, _SCAN_L, _SCAN_LEVELS)).to(x.device),)
        if len(_BUTTER_CACHE) > 32:
            _BUTTER_CACHE.clear()
        _BUTTER_CACHE[key] = coef
    if causal:
        sd, nsec = coef
        check(lib.tl_sosfilt_f64(ptr(x), int(x.dtype == torch.float64), ptr(sd), ptr(y), C, T, nsec, _stream()),
              "tl_sosfilt_f64")
    else:
        bd, ad, zd, ntaps = coef[:4]
        edge = 3 * ntaps
        if T <= edge:
            raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
        work = torch.empty(2, T + 2 * edge, C, dtype=torch.float64, device=x.device)
        if len(coef) > 4:
            # opt-in (TONAL_KERNELS=butter=scan): time-parallel form, 2e-8 - 5e-8 from the sequential kernel - the size of the
            # reference's own rounding (tl_filtfilt_scan_f64's header comment); blocks of _SCAN_L samples, longer blocks for
            # recordings beyond 65 535 of them
            L = _SCAN_L
            while (T + 2 * edge + L - 1) // L > 65535:
                L *= 2
            md = coef[4] if L == _SCAN_L else torch.from_numpy(_scan_matrices(ad.cpu().numpy(), L, _SCAN_LEVELS)).to(x.device)
            nb = (T + 2 * edge + L - 1) // L
            swork = torch.empty(2, nb, 8, C, dtype=torch.float64, device=x.device)
            check(lib.tl_filtfilt_scan_f64(ptr(x), int(x.dtype == torch.float64), ptr(bd), ptr(ad), ptr(zd), ptr(md), _SCAN_LEVELS,
                                           ptr(y), ptr(work), ptr(swork), C, T, ntaps, L, _stream()), "tl_filtfilt_scan_f64")
            out = _ret(y, was_np)
            return out[0] if squeeze else out
        check(lib.tl_filtfilt_f64(ptr(x), int(x.dtype == torch.float64), ptr(bd), ptr(ad), ptr(zd), ptr(y), ptr(work),
                                  C, T, ntaps, _stream()), "tl_filtfilt_f64")
    out = _ret(y, was_np)
    return out[0] if squeeze else out


_BUTTER_CACHE = {}
_SCAN_L = 128            # samples per block of the time-parallel filtfilt
_SCAN_LEVELS = 7         # A^(L 2^m), m < 7: the scan runs over chunks of 128 blocks per workgroup


def _scan_matrices(aa: np.ndarray, L: int, nlev: int) -> np.ndarray:
    """(nlev, 8, 8, 2) double-double pairs of A^(L 2^m), m < nlev: the m-fold squared L-step transition matrix of the
    direct-form-II-transposed recurrence scipy's ``lfilter`` runs (homogeneous part: y = z_0, z_k <- z_{k+1} - a_{k+1} z_0) for
    the normalised denominator ``aa``.  The powers are formed in 600-bit fixed point on Python integers (the entries stay
    below ~1e7, the coefficients are exact binary fractions), then each entry is rounded ONCE to a (hi, lo) pair: the
    device's compensated products see the exact matrix to ~1e-32."""
    from fractions import Fraction
    F = 600
    ns = 8
    one = 1 << F

    def fx(v: float) -> int:
        fr = Fraction(float(v))
        return (fr.numerator << F) // fr.denominator

    A = [[0] * ns for _ in range(ns)]
    for k in range(min(ns, len(aa) - 1)):
        A[k][0] -= fx(aa[k + 1])
        if k + 1 < ns:
            A[k][k + 1] += one

    def mul(X, Y):
        return [[sum(X[i][k] * Y[k][j] for k in range(ns)) >> F for j in range(ns)] for i in range(ns)]

    P = [[one if i == j else 0 for j in range(ns)] for i in range(ns)]
    base, e = A, L
    while e:
        if e & 1:
            P = mul(P, base)
        base = mul(base, base)
        e >>= 1
    out = np.zeros((nlev, ns, ns, 2))
    for m in range(nlev):
        for i in range(ns):
            for j in range(ns):
                fr = Fraction(P[i][j], one)
                hi = float(fr)
                out[m, i, j, 0] = hi
                out[m, i, j, 1] = float(fr - Fraction(hi))
        P = mul(P, P)
    return out


def fir_bandpass_filter(data, fs: float, order: int, center_frequencies: List[float]):
    """Causal FIR band-pass bank, mean over centre frequencies (reference :232-274), including its
    double normalisation of the cut-offs (:265-268).  Output dtype follows the input (:261)."""
    squeeze = False
    if not isinstance(data, torch.Tensor) and np.asarray(data).ndim == 1:
        data = np.asarray(data)[None, :]
        squeeze = True
    x, was_np = _to_device(data)
    C, T = x.shape
    use_ols = T >= _OLS_N                      # (shorter recordings: the time-domain kernel, tl_fir_bank)
    coef, nb, ntap, ols = _fir_coefficients(float(fs), int(order), tuple(float(f) for f in center_frequencies), bool(use_ols),
                                            x.device)
    y = torch.empty(C, T, dtype=x.dtype, device=x.device)
    if ols:
        # the same causal convolution by overlap-save on the LDS-resident FFT (tl_fir_bank_ols): ~5 x fewer fp64 operations
        # at 391 taps
        check(_lib.load().tl_fir_bank_ols(ptr(x), int(x.dtype == torch.float64), ptr(coef), ptr(_ols_twiddles(x.device)), ptr(y),
                                          int(y.dtype == torch.float64), C, T, nb, ntap, _stream()),
              "tl_fir_bank_ols")
    else:
        check(_lib.load().tl_fir_bank(ptr(x), int(x.dtype == torch.float64), ptr(coef), ptr(y),
                                      int(y.dtype == torch.float64), C, T, nb, ntap, _stream()),
              "tl_fir_bank")
    out = _ret(y, was_np)
    return out[0] if squeeze else out


_FIR_CACHE = {}


def _fir_coefficients(fs, order, cfs, use_ols, dev):
    """Device copy of the bank's coefficients, cached per (rate, order, centre frequencies, form): ``firwin`` (reference
    :265-268, with its double normalisation of the cut-offs) and, for the overlap-save form, the kernels' 1024-point spectra
    / 1024 - coefficient generation and its host-to-device copy do not sit in front of every call.
    Returns (tensor, n_bands, n_taps, is_overlap_save)."""
    key = (fs, order, cfs, use_ols, str(dev))
    hit = _FIR_CACHE.get(key)
    if hit is None:
        nyquist = 0.5 * fs
        taps = np.ascontiguousarray(np.array([firwin(order + 1, [fc * 0.9 / nyquist, fc * 1.1 / nyquist], pass_zero=False, fs=fs)
                                              for fc in cfs]), dtype=np.float64)
        if taps.shape[1] > _MAX_TAPS_LDS:
            raise ValueError(f"fir_bandpass_filter: order {order} exceeds the kernel limit {_MAX_TAPS_LDS - 1}")
        ols = use_ols and taps.shape[1] - 1 <= _OLS_N // 2
        if ols:
            g = np.zeros((taps.shape[0], _OLS_N), dtype=np.float64)
            g[:, :taps.shape[1]] = taps
            G = np.fft.fft(g, axis=1) / _OLS_N
            coef = torch.from_numpy(np.ascontiguousarray(np.stack([G.real, G.imag], axis=-1))).to(dev)
        else:
            coef = torch.from_numpy(taps).to(dev)
        if len(_FIR_CACHE) > 32:
            _FIR_CACHE.clear()
        hit = _FIR_CACHE[key] = (coef, taps.shape[0], taps.shape[1], ols)
    return hit


def run(data, params: Namespace):
    """Plugin entry: iterate ``params.bands`` and concatenate the extracted bands as channels
    (reference :9-77)."""
    if "bands" not in params or params.bands is None:
        raise ValueError("bands must be specified in params.")
    all_channels = []
    for freq_config in params.bands:
        method = freq_config.get("method", "hilbert")
        method_params = freq_config.get("params", {})
        if method == 'hilbert':
            if 'freq_ranges' not in method_params:
                raise ValueError("Hilbert filter requires 'freq_ranges' in params.")
            signals = hilbert_filter(data, params.signal_freq, **method_params)
        elif method == 'butter':
            if "freqs" not in method_params:
                raise ValueError("Butterworth filter requires 'freq_range' in params.")
            signals = butter_filter(data, fs=params.signal_freq, **method_params)
        elif method == 'fir':
            if "order" not in method_params or "center_frequencies" not in method_params:
                raise ValueError("FIR filter requires 'order' and 'center_frequencies' in params.")
            signals = fir_bandpass_filter(data, fs=params.signal_freq, order=method_params["order"],
                                          center_frequencies=method_params["center_frequencies"])
        all_channels.append(signals)
    if all_channels and isinstance(all_channels[0], torch.Tensor):
        return torch.cat([s.double() for s in all_channels], dim=0)
    return np.concatenate(all_channels, axis=0)
