"""Loading of ``subject_<id>.npz`` sample files for classifier training
(counterpart of reference data_loading/sample_loading.py:9-194).

File schema (reference data_loading/text_align.py:446-459): ``ecog (N, C, T)``, one integer label array
per target (``syllable``, ``tone``), optional ``ecog_rest``, ``audio``, sampling rates.  The channel
JSON (reference channel_selection_main.py:86-88) maps ``<target>_discriminative`` to channel indices."""
from __future__ import annotations

import json
from argparse import Namespace
from itertools import product
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import TensorDataset


class ClassificationSampleHandler:
    def __init__(self, params: Namespace):
        self.params = params
        self.sample_path = params.sample_path
        self.channel_file = getattr(params, "channel_file", None)
        self.dataset = np.load(self.sample_path)
        self.channels = None
        targets = getattr(params, "targets", None)
        self.targets = [targets] if isinstance(targets, str) else targets

    # ------------------------------------------------------------------ data
    def load_data(self) -> dict:
        """features (N, C_sel, T), joint labels (N,), selected channel indices, classes per target.
        The joint label is mixed radix with the FIRST target least significant (reference :66-71)."""
        key = self.params.features
        if key not in self.dataset:
            raise KeyError(f"The dataset in {self.sample_path} does not contain {key}. "
                           f"Available keys: {', '.join(self.dataset.keys())}")
        features = self.dataset[key]
        per_target, n_classes_dict = [], {}
        for target in self.targets:
            if target not in self.dataset:
                raise KeyError(f"The dataset does not contain '{target}' key. "
                               f"Available keys: {', '.join(self.dataset.keys())}")
            values = self.dataset[target]
            per_target.append(values.flatten())
            n_classes_dict[target] = len(np.unique(values))
        labels = np.zeros_like(per_target[0], dtype=int)
        weight = 1
        for values in per_target:
            labels += values * weight
            weight *= len(np.unique(values))
        self.channels = self._filter_channels(features.shape[1])
        return {"features": features[:, self.channels, :], "labels": labels, "selected_channels": self.channels,
                "n_classes_dict": n_classes_dict}

    def _filter_channels(self, n_channels: int) -> np.ndarray:
        if self.channel_file is None:
            return np.arange(n_channels)
        with open(self.channel_file, "r") as f:
            selections = json.load(f)
        chosen = set()
        for target in self.targets:
            key = f"{target}_discriminative"
            if key not in selections:
                raise KeyError(f"Channel selection for '{key}' not found in the file {self.channel_file}. "
                               f"Available keys: {', '.join(selections.keys())}")
            chosen.update(selections[key])
        if not chosen:
            raise ValueError(f"No channels found for the targets: {', '.join(self.targets)}. "
                             f"Please check the channel file {self.channel_file}")
        return np.array(sorted(chosen))

    def prepare_torch_dataset(self, features: np.ndarray, labels: np.ndarray, device: str) -> TensorDataset:
        """float32 features and float32 labels on ``device`` (the trainer casts labels to int64)."""
        return TensorDataset(torch.tensor(features, dtype=torch.float32).to(device),
                             torch.tensor(labels, dtype=torch.float32).to(device))

    # ------------------------------------------------------------------ names
    def prepare_class_labels(self, n_classes_dict: Optional[Dict[str, int]] = None) -> List[str]:
        """Display names of the (joint) classes: configured ``class_labels`` per target, else "1".."n";
        several targets give the Cartesian product joined with "_"."""
        named = getattr(self.params, "class_labels", {}) or {}

        def names_of(target):
            if named.get(target) is not None:
                return named[target]
            if n_classes_dict is None or target not in n_classes_dict:
                raise ValueError(f"Number of classes for target '{target}' is not provided.")
            return np.arange(1, n_classes_dict[target] + 1).astype(str)

        if len(self.targets) > 1:
            return ["_".join(combo) for combo in product(*[names_of(t) for t in self.targets])]
        return names_of(self.targets[0])
