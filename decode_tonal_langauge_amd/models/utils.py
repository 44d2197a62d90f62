"""Mirror of reference models/utils.py (activation factory, decay groups)."""
from typing import List, Tuple

import torch.nn as nn

_ACTS = {"ELU": nn.ELU, "ReLU": nn.ReLU, "LeakyReLU": nn.LeakyReLU, "PReLU": nn.PReLU, "GLU": nn.GLU, "GELU": nn.GELU}


def get_activation(activation: str, **kwargs) -> nn.Module:
    if activation not in _ACTS:
        raise ValueError(f"Unsupported activation function: {activation}")
    return _ACTS[activation](**kwargs)


def split_decay_groups(named_params: List[Tuple[str, object]]):
    decay, no_decay = [], []
    for _, p in named_params:
        if not p.requires_grad:
            continue
        (decay if p.ndim >= 2 else no_decay).append(p)
    return decay, no_decay
