"""Audio <-> mel-spectrogram helpers (mirror of reference utils/audio.py:7-87), without librosa.

The reference delegates to ``librosa.feature.melspectrogram`` + ``librosa.power_to_db`` (``audio_to_mel``, :36-43) and to
``librosa.db_to_power`` + ``librosa.feature.inverse.mel_to_audio`` (``mel_to_audio``, :76-87).  librosa is a third-party
dependency that is neither vendored in the reference nor installable in this image, so this module RESTATES the published
algorithms of librosa 0.10 with NumPy / SciPy:

* STFT: ``n_fft`` 2048, ``hop_length`` n_fft / 4 = 512 by default, periodic Hann window, ``center=True`` with zero
  ("constant") padding, power spectrogram ``|S|^2``;
* mel filter bank: Slaney scale (linear below 1 kHz, logarithmic above; ``htk=False``), triangular filters with area
  normalisation (``norm='slaney'``), ``fmin`` 0, ``fmax`` sr / 2, 128 bands by default;
* ``power_to_db(S, ref=np.max)``: ``10 log10(max(S, 1e-10)) - 10 log10(max(ref, 1e-10))`` clipped at ``top_db`` = 80 below the peak;
* inverse: ``db_to_power`` (``ref * 10^(dB / 10)``), mel -> linear magnitude by non-negative least squares against the
  filter bank, Griffin-Lim with momentum 0.99 and 32 iterations from random phases.

**Parity unpinned**: the reference holds no golden vector for these calls and librosa cannot be imported here to make one;
the tests check the published properties (filter-bank partition / normalisation, dB reference and floor, a tone landing in
its band, spectral convergence of the inversion).  This is CPU host code beside the hot path: the
synthesis trainer consumes its output (mel targets), nothing here runs per train step.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
from scipy.optimize import nnls


# ------------------------------------------------------------------------------------------ mel scale (Slaney)
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax: Optional[float] = None) -> np.ndarray:
    """(n_mels, 1 + n_fft // 2) triangular filters on the Slaney mel scale, each normalised to unit area in Hz."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


# ------------------------------------------------------------------------------------------ STFT
def _hann(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)          # periodic ("fftbins=True")


def stft(y: np.ndarray, n_fft: int = 2048, hop_length: Optional[int] = None, win_length: Optional[int] = None,
         center: bool = True) -> np.ndarray:
    hop = n_fft // 4 if hop_length is None else int(hop_length)
    wl = n_fft if win_length is None else int(win_length)
    win = np.zeros(n_fft)
    off = (n_fft - wl) // 2
    win[off:off + wl] = _hann(wl)
    y = np.asarray(y, dtype=np.float64)
    if center:
        y = np.pad(y, n_fft // 2, mode="constant")
    if y.shape[0] < n_fft:
        raise ValueError(f"audio of {y.shape[0]} samples is shorter than n_fft = {n_fft}")
    n_frames = 1 + (y.shape[0] - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    return np.fft.rfft(y[idx] * win[:, None], axis=0)                    # (1 + n_fft // 2, n_frames)


def istft(S: np.ndarray, hop_length: Optional[int] = None, win_length: Optional[int] = None, center: bool = True,
          length: Optional[int] = None) -> np.ndarray:
    n_fft = 2 * (S.shape[0] - 1)
    hop = n_fft // 4 if hop_length is None else int(hop_length)
    wl = n_fft if win_length is None else int(win_length)
    win = np.zeros(n_fft)
    off = (n_fft - wl) // 2
    win[off:off + wl] = _hann(wl)
    frames = np.fft.irfft(S, n=n_fft, axis=0) * win[:, None]
    n_frames = S.shape[1]
    out = np.zeros(n_fft + hop * (n_frames - 1))
    wsum = np.zeros_like(out)
    for t in range(n_frames):
        out[t * hop:t * hop + n_fft] += frames[:, t]
        wsum[t * hop:t * hop + n_fft] += win ** 2
    nz = wsum > np.finfo(np.float32).tiny
    out[nz] /= wsum[nz]
    if center:
        out = out[n_fft // 2:len(out) - n_fft // 2] if length is None else out[n_fft // 2:n_fft // 2 + length]
    elif length is not None:
        out = out[:length]
    return out


# ------------------------------------------------------------------------------------------ dB
def power_to_db(S: np.ndarray, ref=1.0, amin: float = 1e-10, top_db: Optional[float] = 80.0) -> np.ndarray:
    S = np.asarray(S)
    ref_value = ref(S) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def db_to_power(S_db: np.ndarray, ref: float = 1.0) -> np.ndarray:
    return ref * np.power(10.0, 0.1 * np.asarray(S_db))


# ------------------------------------------------------------------------------------------ the reference's two functions
def _split_kwargs(kw: Optional[dict]):
    kw = dict(kw or {})
    stft_kw = {k: kw.pop(k) for k in ("n_fft", "hop_length", "win_length", "center") if k in kw}
    power = kw.pop("power", 2.0)
    mel_kw = {k: kw.pop(k) for k in ("n_mels", "fmin", "fmax") if k in kw}
    if kw:
        raise TypeError(f"unsupported mel keyword(s) {sorted(kw)} (supported: n_fft, hop_length, win_length, center, "
                        "power, n_mels, fmin, fmax)")
    return stft_kw, power, mel_kw


def audio_to_mel(audio: np.ndarray, audio_sampling_rate: int, mel_in_db: bool = True,
                 mel_kwargs: Optional[dict] = None) -> np.ndarray:
    """Mel spectrogram of a 1-D signal, flattened to (n_mels * n_frames,) - reference utils/audio.py:7-43."""
    audio = np.asarray(audio)
    if audio.ndim > 1:
        raise ValueError("Audio input must be a 1D array.")
    stft_kw, power, mel_kw = _split_kwargs(mel_kwargs)
    n_fft = stft_kw.get("n_fft", 2048)
    S = np.abs(stft(audio, **{"n_fft": n_fft, **{k: v for k, v in stft_kw.items() if k != "n_fft"}})) ** power
    mel = mel_filterbank(audio_sampling_rate, n_fft, **mel_kw).astype(np.float64) @ S
    if mel_in_db:
        mel = power_to_db(mel, ref=np.max)
    return mel.astype(np.float32).reshape(-1)


def griffinlim(mag: np.ndarray, n_iter: int = 32, hop_length: Optional[int] = None, win_length: Optional[int] = None,
               momentum: float = 0.99, seed: int = 0, length: Optional[int] = None) -> np.ndarray:
    """Fast Griffin-Lim (Perraudin et al. 2013) from random initial phases."""
    rng = np.random.default_rng(seed)
    angles = np.exp(2j * np.pi * rng.random(mag.shape))
    n_fft = 2 * (mag.shape[0] - 1)
    rebuilt = tprev = None
    for _ in range(n_iter):
        inverse = istft(mag * angles, hop_length=hop_length, win_length=win_length, length=length)
        rebuilt = stft(inverse, n_fft=n_fft, hop_length=hop_length, win_length=win_length)
        rebuilt = rebuilt[:, :mag.shape[1]] if rebuilt.shape[1] >= mag.shape[1] else \
            np.pad(rebuilt, ((0, 0), (0, mag.shape[1] - rebuilt.shape[1])))
        angles = rebuilt - (momentum / (1 + momentum)) * tprev if tprev is not None else rebuilt.copy()
        angles /= np.abs(angles) + 1e-16
        tprev = rebuilt
    return istft(mag * angles, hop_length=hop_length, win_length=win_length, length=length)


def mel_to_audio(mel: np.ndarray, n_mels: int, audio_sampling_rate: int = 24414, mel_in_db: bool = True, **kwargs) -> np.ndarray:
    """Waveform from a flattened mel spectrogram by NNLS mel inversion + Griffin-Lim - reference utils/audio.py:46-87."""
    mel = np.asarray(mel, dtype=np.float64).reshape(n_mels, -1)
    if mel_in_db:
        mel = db_to_power(mel, ref=0.0001)
    n_fft = kwargs.pop("n_fft", 2048)
    hop_length, win_length = kwargs.pop("hop_length", None), kwargs.pop("win_length", None)
    power, n_iter = kwargs.pop("power", 2.0), kwargs.pop("n_iter", 32)
    length, seed = kwargs.pop("length", None), kwargs.pop("seed", 0)
    fb_kw = {k: kwargs.pop(k) for k in ("fmin", "fmax") if k in kwargs}
    if kwargs:
        raise TypeError(f"unsupported keyword(s) {sorted(kwargs)}")
    fb = mel_filterbank(audio_sampling_rate, n_fft, n_mels=n_mels, **fb_kw).astype(np.float64)
    lin = np.stack([nnls(fb, mel[:, t])[0] for t in range(mel.shape[1])], axis=1)      # power (or magnitude^power) spectrum
    mag = np.power(np.maximum(lin, 0.0), 1.0 / power)
    return griffinlim(mag, n_iter=n_iter, hop_length=hop_length, win_length=win_length, seed=seed, length=length).astype(np.float32)
