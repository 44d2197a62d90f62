"""CPU oracle for the preprocess/signal band-extraction path.  TEST INFRASTRUCTURE ONLY.

NumPy float64 restatement of the reference's ``preprocess/signal/frequency_filter.py``.
Never imported by the product package; only tests, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it, as the checker.

Parity status: PINNED by ``oracle/make_golden.py`` (imports the real reference in the build
container, compares, and writes ``tests/golden/signal_*.npz``).

Third-party arithmetic behind the reference (scipy==1.11.4 in its requirements.txt:1; 1.15.3
here): ``scipy.fft.fft/ifft`` (restated with ``numpy.fft``), ``scipy.signal.filtfilt`` /
``lfilter`` / ``lfilter_zi`` / ``sosfilt`` (restated below as explicit direct-form-II-transposed
recurrences).  Filter *design* (``butter``, ``firwin``) is coefficient generation, not the hot
path: both this oracle and the product call scipy for it.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple, Union

import numpy as np
from scipy.signal import butter as _butter_design, firwin as _firwin_design


# ---------------------------------------------------------------------------
# Gaussian filter bank + analytic signal (frequency_filter.py:80-184)
# ---------------------------------------------------------------------------

def gaussian_bank(freq_ranges, sampling_rate: float, f0: float = 0.018, octspace: float = 1 / 7,
                  filterbank_bias: float = math.log10(0.39), filterbank_slope: float = 0.5
                  ) -> Tuple[np.ndarray, np.ndarray]:
    """Centre frequencies and Gaussian widths (frequency_filter.py:121-153).

    Mirrors the reference's argument normalisation, including its quirk that a *list of
    ints* is not recognised as a single range (frequency_filter.py:121-124)."""
    if isinstance(freq_ranges, tuple):
        freq_ranges = [freq_ranges]
    if isinstance(freq_ranges[0], float):
        freq_ranges = [tuple(freq_ranges)]
    cfs: List[float] = []
    sds: List[float] = []
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


def hilbert_filter(data: np.ndarray, sampling_rate: float, freq_ranges, f0: float = 0.018,
                   octspace: float = 1 / 7, filterbank_bias: float = math.log10(0.39),
                   filterbank_slope: float = 0.5, envelope: bool = True) -> np.ndarray:
    """frequency_filter.py:80-184.  Whole-recording FFT, one Gaussian x analytic multiplier
    per band, inverse FFT, |.| (or real part), mean over bands.  Always float64 out (the
    reference accumulates into ``np.zeros((C, T, n_banks))``, :170)."""
    C, T = data.shape
    cfs, sds = gaussian_bank(freq_ranges, sampling_rate, f0, octspace, filterbank_bias, filterbank_slope)
    freqs = np.fft.fftfreq(T, d=1.0 / sampling_rate)
    mult = np.zeros(T)
    if T % 2 == 0:
        mult[0] = 1
        mult[1:T // 2] = 2
        mult[T // 2] = 1
    else:
        mult[0] = 1
        mult[1:(T + 1) // 2] = 2
    # scipy.fft keeps single precision for float32 input (complex64); numpy.fft promotes.
    # The reference therefore computes in complex64 for float32 input; we restate the
    # float64 arithmetic and bound the float32 case by tolerance in the tests.
    X = np.fft.fft(np.asarray(data, dtype=np.float64), axis=1)
    acc = np.zeros((C, T))
    for fc, sf in zip(cfs, sds):
        H = np.exp(-0.5 * ((freqs - fc) / sf) ** 2)
        H[0] = 0
        sig = np.fft.ifft(X * (H * mult)[None, :], axis=1)
        acc += np.abs(sig) if envelope else sig.real
    return acc / len(cfs)


# ---------------------------------------------------------------------------
# IIR: lfilter / lfilter_zi / filtfilt / sosfilt restated
# ---------------------------------------------------------------------------

def lfilter_df2t(b: np.ndarray, a: np.ndarray, x: np.ndarray, zi: np.ndarray | None = None
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Direct-form II transposed recurrence along the last axis (scipy.signal.lfilter):
    y[n] = b0 x[n] + z0;  z_k = b_{k+1} x[n] + z_{k+1} - a_{k+1} y[n].  x: (C, T)."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    if a[0] != 1.0:
        b = b / a[0]
        a = a / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    x = np.asarray(x, dtype=np.float64)
    C, T = x.shape
    z = np.zeros((C, n - 1)) if zi is None else np.array(zi, dtype=np.float64, copy=True)
    y = np.empty_like(x)
    for t in range(T):
        xt = x[:, t]
        yt = b[0] * xt + (z[:, 0] if n > 1 else 0.0)
        for k in range(n - 2):
            z[:, k] = b[k + 1] * xt + z[:, k + 1] - a[k + 1] * yt
        if n > 1:
            z[:, n - 2] = b[n - 1] * xt - a[n - 1] * yt
        y[:, t] = yt
    return y, z


def lfilter_zi(b: np.ndarray, a: np.ndarray) -> np.ndarray:
    """Steady-state DF-II-T state for a unit step input (scipy.signal.lfilter_zi):
    solve (I - A^T) zi = B with A the companion matrix of a."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    if a[0] != 1.0:
        b = b / a[0]
        a = a / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    comp[1:, :-1] = np.eye(n - 2)
    IminusA = np.eye(n - 1) - comp.T
    B = b[1:] - a[1:] * b[0]
    return np.linalg.solve(IminusA, B)


def filtfilt(b: np.ndarray, a: np.ndarray, x: np.ndarray) -> np.ndarray:
    """scipy.signal.filtfilt defaults (padtype='odd', padlen=3*max(len(a),len(b)), method='pad')
    as the reference calls it (frequency_filter.py:226-227)."""
    x = np.asarray(x, dtype=np.float64)
    C, T = x.shape
    ntaps = max(len(a), len(b))
    edge = 3 * ntaps
    if T <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    left = 2 * x[:, :1] - x[:, edge:0:-1]
    right = 2 * x[:, -1:] - x[:, -2:-(edge + 2):-1]
    ext = np.concatenate([left, x, right], axis=1)
    zi = lfilter_zi(b, a)
    y, _ = lfilter_df2t(b, a, ext, zi=zi[None, :] * ext[:, :1])
    yr = y[:, ::-1]
    y2, _ = lfilter_df2t(b, a, yr, zi=zi[None, :] * yr[:, :1])
    y2 = y2[:, ::-1]
    return y2[:, edge:-edge]


def sosfilt(sos: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Cascade of biquads, zero initial state (scipy.signal.sosfilt)."""
    y = np.asarray(x, dtype=np.float64)
    for sec in np.asarray(sos, dtype=np.float64):
        y, _ = lfilter_df2t(sec[:3], sec[3:], y)
    return y


def butter_filter(data: np.ndarray, freqs, fs: float, order: int = 4, causal: bool = False,
                  filter_type: str = "bandpass") -> np.ndarray:
    """frequency_filter.py:187-229."""
    nyq = 0.5 * fs
    wn = np.asarray(freqs, dtype=float) / nyq
    if causal:
        sos = _butter_design(order, wn, btype=filter_type, output="sos")
        return sosfilt(sos, data)
    b, a = _butter_design(order, wn, btype=filter_type)
    return filtfilt(b, a, data)


# ---------------------------------------------------------------------------
# FIR band-pass bank (frequency_filter.py:232-274)
# ---------------------------------------------------------------------------

def fir_taps(fs: float, order: int, center_frequencies: Sequence[float]) -> np.ndarray:
    """Tap matrix (n_bands, order+1).  Reproduces the reference's double normalisation
    (cut-offs divided by Nyquist *and* ``fs=fs`` passed to firwin, :265-268)."""
    nyq = 0.5 * fs
    taps = []
    for fc in center_frequencies:
        low = fc * 0.9 / nyq
        high = fc * 1.1 / nyq
        taps.append(_firwin_design(order + 1, [low, high], pass_zero=False, fs=fs))
    return np.array(taps)


def fir_bandpass_filter(data: np.ndarray, fs: float, order: int,
                        center_frequencies: Sequence[float]) -> np.ndarray:
    """Causal FIR with zero initial state per band, mean over bands (:260-274).
    Output dtype follows ``np.zeros_like(data)`` (:261)."""
    x = np.asarray(data, dtype=np.float64)
    taps = fir_taps(fs, order, center_frequencies)
    out = np.zeros_like(data)
    C, T = x.shape
    for h in taps:
        y = np.empty((C, T))
        for c in range(C):
            y[c] = np.convolve(x[c], h)[:T]
        out += y.astype(out.dtype, copy=False) if out.dtype != np.float64 else y
    out /= len(center_frequencies)
    return out


# ---------------------------------------------------------------------------
# Plugin entry (frequency_filter.py:9-77)
# ---------------------------------------------------------------------------

def run(data: np.ndarray, params) -> np.ndarray:
    if "bands" not in params or params.bands is None:
        raise ValueError("bands must be specified in params.")
    chans = []
    for cfg in params.bands:
        method = cfg.get("method", "hilbert")
        mp = cfg.get("params", {})
        if method == "hilbert":
            if "freq_ranges" not in mp:
                raise ValueError("Hilbert filter requires 'freq_ranges' in params.")
            sig = hilbert_filter(data, params.signal_freq, **mp)
        elif method == "butter":
            if "freqs" not in mp:
                raise ValueError("Butterworth filter requires 'freq_range' in params.")
            sig = butter_filter(data, fs=params.signal_freq, **mp)
        elif method == "fir":
            if "order" not in mp or "center_frequencies" not in mp:
                raise ValueError("FIR filter requires 'order' and 'center_frequencies' in params.")
            sig = fir_bandpass_filter(data, fs=params.signal_freq, order=mp["order"],
                                      center_frequencies=mp["center_frequencies"])
        chans.append(sig)
    return np.concatenate(chans, axis=0)


# ---------------------------------------------------------------------------
# Other preprocess/signal steps (the plugin ABI's neighbours of the band extraction)
# ---------------------------------------------------------------------------

def channel_zscore(data: np.ndarray, preserve_nans: bool = True) -> np.ndarray:
    """preprocess/signal/channel_zscore.py:22-27: population std over the whole recording."""
    mean = np.mean(data, axis=1, keepdims=True)
    std = np.std(data, axis=1, keepdims=True)
    z = (data - mean) / std
    if not preserve_nans:
        z[np.isnan(z)] = 0
    return z


def zscore_rereference(data: np.ndarray, start: int, end: int) -> np.ndarray:
    """preprocess/signal/zscore_rereference.py:60-70: statistics of data[:, start:end]."""
    if start < 0 or end > data.shape[1]:
        raise ValueError("Reference time indices are out of bounds.")
    if start >= end:
        raise ValueError("Start time must be less than end time.")
    m = np.mean(data[:, start:end], axis=1, keepdims=True)
    s = np.std(data[:, start:end], axis=1, keepdims=True)
    return (data - m) / s


def car_rereference(data: np.ndarray, exclude_channels=()) -> np.ndarray:
    """preprocess/signal/car_rereference.py:34-39."""
    mask = np.ones(data.shape[0], dtype=bool)
    mask[list(exclude_channels)] = False
    return data - np.mean(data[mask, :], axis=0, keepdims=True)


def rolling_zscore(data: np.ndarray, window_size: int, preserve_nans: bool = True) -> np.ndarray:
    """preprocess/signal/rolling_zscore.py:36-49 restated without pandas: trailing window of
    ``window_size`` samples, min_periods=1, mean and *sample* std (ddof=1) of the non-NaN values."""
    x = np.asarray(data, dtype=np.float64)
    C, T = x.shape
    out = np.full((C, T), np.nan)
    for t in range(T):
        w = x[:, max(0, t - window_size + 1):t + 1]
        n = np.sum(~np.isnan(w), axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = np.nansum(w, axis=1) / n
            var = np.nansum((w - mean[:, None]) ** 2, axis=1) / (n - 1)
            out[:, t] = np.where(n >= 2, (x[:, t] - mean) / np.sqrt(var), np.nan)
    if not preserve_nans:
        out[np.isnan(out)] = 0
    return out


def resample_fft(x: np.ndarray, num: int) -> np.ndarray:
    """``scipy.signal.resample(x, num, axis=1)`` for real input, restated with numpy.fft: rfft, copy
    the bins up to the smaller Nyquist (doubling / halving a shared even-length Nyquist bin), irfft,
    scale by num/N.  (preprocess/signal/downsample.py:25 calls it.)  Like scipy.fft, float32 input
    is transformed in single precision."""
    x = np.asarray(x)
    nx = x.shape[1]
    dt = np.complex64 if x.dtype == np.float32 else np.complex128
    X = np.fft.rfft(x.astype(np.float64), axis=1)
    Y = np.zeros((x.shape[0], num // 2 + 1), dtype=np.complex128)
    N = min(num, nx)
    nyq = N // 2 + 1
    Y[:, :nyq] = X[:, :nyq]
    if N % 2 == 0:
        if num < nx:
            Y[:, N // 2] *= 2.0
        elif nx < num:
            Y[:, N // 2] *= 0.5
    y = np.fft.irfft(Y, num, axis=1) * (float(num) / float(nx))
    return y.astype(np.float32) if dt == np.complex64 else y


def downsample(data: np.ndarray, signal_freq: float, downsample_freq: float = 400):
    """preprocess/signal/downsample.py:21-27: returns (resampled data, new signal_freq)."""
    n_samples = int(data.shape[1] * (downsample_freq / signal_freq))
    return resample_fft(data, n_samples), downsample_freq
