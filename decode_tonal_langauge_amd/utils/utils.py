"""Mirror of reference utils/utils.py:6-18."""
import numpy as np
import torch


def set_seeds(seed: int):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
