"""Seeded input generators shared by oracle/make_golden.py and the parity tests.

The goldens under tests/golden/ hold the reference's *outputs*; the inputs are regenerated
from these seeds (CPU torch / numpy generators are deterministic for a given version; each
golden also stores an input checksum so RNG drift is detected rather than silently accepted).
"""
from __future__ import annotations

import numpy as np
import torch

TONE_MAP = {"0": [3, 3, 3, 3, 3], "1": [1, 2, 3, 4, 5], "2": [3, 2, 1, 2, 4], "3": [5, 4, 3, 2, 1]}


def tone_dynamics(tones: torch.Tensor, syls: torch.Tensor, mapping=TONE_MAP) -> torch.Tensor:
    """(B,2,L) float tensor [[syl]*L, mapping[tone]] - what the trainer feeds the model
    (reference models/synthesis_trainer.py:212-218)."""
    rows = [[[int(s)] * len(mapping[str(int(t))]), list(mapping[str(int(t))])] for t, s in zip(tones, syls)]
    return torch.tensor(rows, dtype=torch.float32)


def train_batches(nsteps: int, B: int, C: int, T: int, seed: int = 1234, out_dim: int = 80):
    """Synthetic batches in the layout of SURVEY.md section 8d."""
    gen = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, C, T, generator=gen) for _ in range(nsteps)]
    tones = [torch.randint(0, 4, (B,), generator=gen) for _ in range(nsteps)]
    syls = [torch.randint(0, 2, (B,), generator=gen) for _ in range(nsteps)]
    labs = [tone_dynamics(t, s) for t, s in zip(tones, syls)]
    tg = [10 * torch.randn(B, out_dim, generator=gen) for _ in range(nsteps)]
    return xs, tones, syls, labs, tg


def g1_inputs():
    """After torch.manual_seed(0) and model construction: x, lab drawn from the global RNG."""
    x = torch.randn(3, 4, 100)
    lab = torch.randn(3, 2, 5)
    return x, lab


def g2_inputs():
    x = torch.randn(4, 32, 200)
    lab = torch.randn(4, 2, 5)
    return x, lab


def g6_inputs():
    x = np.random.default_rng(0).standard_normal((2, 1000))
    x2 = np.random.default_rng(1).standard_normal((3, 777)).astype(np.float32)
    return x, x2


def g9_dataset(N: int = 96, C: int = 32, T: int = 200, seed: int = 1234):
    gen = torch.Generator().manual_seed(seed)
    e_non = torch.randn(N, C, T, generator=gen)
    e_syl = torch.randn(N, 8, T, generator=gen)
    e_tone = torch.randn(N, 8, T, generator=gen)
    tgt = 10 * torch.randn(N, 80, generator=gen)
    return e_non, e_syl, e_tone, tgt


def checksum(*arrays) -> float:
    s = 0.0
    for a in arrays:
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        s += float(np.abs(a.astype(np.float64)).sum())
    return s


def update_rel_l2(final_a, final_b, init) -> float:
    """Relative L2 distance between two parameter *updates* (final - init).  NAdam turns a
    gradient that is numerically ~0 into a +-lr step, so element-wise comparison of final
    parameters is ill-conditioned; the update vector as a whole is not."""
    da = (np.asarray(final_a, dtype=np.float64) - np.asarray(init, dtype=np.float64)).ravel()
    db = (np.asarray(final_b, dtype=np.float64) - np.asarray(init, dtype=np.float64)).ravel()
    return float(np.linalg.norm(da - db) / max(np.linalg.norm(db), 1e-30))


def c1_subject(N: int = 400, C: int = 16, T: int = 100, seed: int = 2024):
    """BASELINE config C1: synthetic ``subject_<id>.npz`` content - N(0,1) ECoG plus a class pattern,
    4 tones x 2 syllables.  Returns a dict of arrays with the reference's sample-file keys."""
    rng = np.random.default_rng(seed)
    tone = rng.integers(0, 4, N)
    syllable = rng.integers(0, 2, N)
    ecog = rng.standard_normal((N, C, T)).astype(np.float32)
    for k in range(4):                       # tone k lifts channels 4k..4k+3, syllable 1 adds a slow ramp
        ecog[tone == k, 4 * k:4 * k + 4, :] += 0.5
    ecog[syllable == 1] += np.linspace(-0.4, 0.4, T, dtype=np.float32)
    return {"ecog": ecog, "ecog_sf": np.array(100), "tone": tone, "syllable": syllable,
            "ecog_rest": rng.standard_normal((N, C, T)).astype(np.float32)}


#: golden G13: the step list handed to the reference's preprocess_signal (module names as in the reference tree)
CHAIN_STEPS = [
    {"module": "preprocess.signal.downsample", "params": {"downsample_freq": 400}},
    {"module": "preprocess.signal.car_rereference", "params": {"exclude_channels": [2]}},
    {"module": "preprocess.signal.frequency_filter", "params": {"bands": [
        {"method": "hilbert", "params": {"freq_ranges": [70., 150.], "envelope": True}},
        {"method": "butter", "params": {"freqs": [0.3, 100], "filter_type": "bandpass"}}]}},
    {"module": "preprocess.signal.channel_zscore", "params": {}},
]


def chain_input():
    return np.random.default_rng(11).standard_normal((6, 3000)) * 2.0 + 0.3
