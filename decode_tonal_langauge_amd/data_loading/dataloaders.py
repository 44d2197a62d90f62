"""``split_dataset`` (mirror of reference data_loading/dataloaders.py:11-74): the seeded
``random_split`` + DataLoader construction that fixes batch composition for a seed."""
from typing import List

import torch
from torch.utils.data import DataLoader, TensorDataset, random_split


def split_dataset(dataset: TensorDataset, ratios: List[float], shuffling: List[bool], batch_size: int = 8,
                  seed: int = 42) -> List[DataLoader]:
    torch.manual_seed(seed)
    n_samples = len(dataset)
    sizes: List[int] = []
    for i, ratio in enumerate(ratios):
        if ratio <= 0 or ratio >= 1:
            raise ValueError("All ratios must be between 0 and 1 (exclusive).")
        sizes.append(n_samples - sum(sizes) if i == len(ratios) - 1 else int(n_samples * ratio))
    subsets = random_split(dataset, sizes)
    return [DataLoader(sub, batch_size=batch_size, shuffle=shuffling[i]) for i, sub in enumerate(subsets)]
