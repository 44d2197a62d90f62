"""Synthetic data in the reference's on-disk schemas (no recordings ship with either repository).

Schemas:
  * ``subject_<id>.npz``   ``ecog (N, C, T) float32``, ``ecog_sf``, ``audio (N, S)``, ``audio_sf``, ``mel (N, n_mels)``,
                           ``syllable (N,)``, ``tone (N,)``, ``ecog_rest (N, C, T)``
                           (reference data_loading/text_align.py:446-459; ``mel`` as consumed by train_synthesizer)
  * ``subject_<id>.json``  channel selections ``{active_channels, tone_discriminative, syllable_discriminative}``
                           (reference channel_selection_main.py:86-88)
  * ``config.json``        ``{mel_kwargs, n_syllables, n_tones, tone_dynamic_mapping}`` (reference train_synthesizer.py:171-177)
  * ``B<block>_<modality>.npz``  ``data (C, T)``, ``sf`` (reference preprocess/io/tdt_blocks.py:33-34)

The ECoG carries a learnable class pattern: tone k lifts a group of channels, syllable 1 adds a slow
ramp, so classifiers and the synthesiser have something to fit."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np

DEFAULT_TONE_DYNAMICS = {"0": [3, 3, 3, 3, 3], "1": [1, 2, 3, 4, 5], "2": [3, 2, 1, 2, 4], "3": [5, 4, 3, 2, 1]}


def make_subject(n_samples: int = 400, n_channels: int = 16, n_timepoints: int = 100, n_tones: int = 4,
                 n_syllables: int = 2, n_mels: int = 80, ecog_sf: int = 100, audio_sf: int = 16000,
                 seed: int = 2024) -> Dict[str, np.ndarray]:
    """Arrays of one ``subject_<id>.npz``."""
    rng = np.random.default_rng(seed)
    tone = rng.integers(0, n_tones, n_samples)
    syllable = rng.integers(0, n_syllables, n_samples)
    ecog = rng.standard_normal((n_samples, n_channels, n_timepoints)).astype(np.float32)
    group = max(1, n_channels // n_tones)
    for k in range(n_tones):
        ecog[tone == k, k * group:(k + 1) * group, :] += 0.5
    ecog[syllable == 1] += np.linspace(-0.4, 0.4, n_timepoints, dtype=np.float32)
    n_audio = int(n_timepoints / ecog_sf * audio_sf)
    t = np.arange(n_audio) / audio_sf
    f0 = 110.0 * (1 + tone[:, None]) * (1 + 0.1 * syllable[:, None])
    audio = (0.1 * np.sin(2 * np.pi * f0 * t[None, :])).astype(np.float32)
    mel = (10 * rng.standard_normal((n_samples, n_mels)) + 5 * tone[:, None]).astype(np.float32)
    return {"ecog": ecog, "ecog_sf": np.array(ecog_sf), "audio": audio, "audio_sf": np.array(audio_sf), "mel": mel,
            "syllable": syllable, "tone": tone,
            "ecog_rest": rng.standard_normal((n_samples, n_channels, n_timepoints)).astype(np.float32)}


def write_dataset(root: str, subject_ids=(1,), channels: Optional[Dict[str, List[int]]] = None,
                  tone_dynamic_mapping: Optional[Dict[str, List[int]]] = None, **subject_kwargs) -> Dict[str, str]:
    """Write ``samples/subject_<id>.npz``, ``channels/subject_<id>.json`` and ``samples/config.json`` under
    ``root``.  Returns the directories / files written."""
    sample_dir, chan_dir = os.path.join(root, "samples"), os.path.join(root, "channels")
    os.makedirs(sample_dir, exist_ok=True)
    os.makedirs(chan_dir, exist_ok=True)
    n_mels = subject_kwargs.get("n_mels", 80)
    for i, sid in enumerate(subject_ids):
        subj = make_subject(seed=subject_kwargs.pop("seed", 2024) + i, **subject_kwargs)
        np.savez(os.path.join(sample_dir, f"subject_{sid}.npz"), **subj)
        C = subj["ecog"].shape[1]
        sel = channels or {"active_channels": list(range(C)), "tone_discriminative": list(range(0, C // 2)),
                           "syllable_discriminative": list(range(C // 2, C))}
        with open(os.path.join(chan_dir, f"subject_{sid}.json"), "w") as f:
            json.dump(sel, f)
    cfg = {"mel_kwargs": {"n_mels": n_mels}, "n_syllables": subject_kwargs.get("n_syllables", 2),
           "n_tones": subject_kwargs.get("n_tones", 4),
           "tone_dynamic_mapping": tone_dynamic_mapping or DEFAULT_TONE_DYNAMICS}
    config_file = os.path.join(sample_dir, "config.json")
    with open(config_file, "w") as f:
        json.dump(cfg, f)
    return {"sample_dir": sample_dir, "channel_selection_dir": chan_dir, "config_file": config_file}


def write_raw_block(root: str, block: int = 1, modality: str = "ecog", n_channels: int = 256, seconds: float = 60.0,
                    sf: float = 400.0, seed: int = 0) -> str:
    """``B<block>_<modality>.npz`` with ``data (C, T) float32`` and ``sf`` - the input of the preprocess steps."""
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((n_channels, int(seconds * sf))).astype(np.float32)
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, f"B{block}_{modality}.npz")
    np.savez(path, data=data, sf=sf)
    return path
