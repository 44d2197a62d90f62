"""Label helpers of the train loop (mirror of reference data_loading/utils.py:32-79,119-156)."""
from typing import Dict, List

import numpy as np


def prepare_tone_dynamics(tone_dynamic_mapping: Dict[str, List[int]], tone_labels: np.ndarray,
                          syllable_labels: np.ndarray) -> np.ndarray:
    """Host version kept for API parity; the trainer uses the device gather ``tl_tone_dynamics``."""
    if len(tone_labels) != len(syllable_labels):
        raise ValueError("Length of tone labels and syllable labels must match.")
    dynamics = []
    for tone, syllable in zip(tone_labels, syllable_labels):
        try:
            tone_dynamic = tone_dynamic_mapping[str(tone)]
        except KeyError:
            raise ValueError(f"Tone {str(tone)} not found in tone_dynamic_mapping."
                             f"Available tones in mapping: {list(tone_dynamic_mapping.keys())}")
        dynamics.append(np.array([[syllable] * len(tone_dynamic), tone_dynamic]))
    return np.array(dynamics)


def select_non_discriminative_channels(channel_selections: dict, discriminative_keys: List[str]) -> list:
    keep = set(channel_selections['active_channels'])
    drop = set()
    for label in discriminative_keys:
        drop.update(channel_selections[label])
    return sorted(keep - drop)
