"""``ClassifierModel`` plugin base class (mirror of reference models/classifier.py:7-78).

Classifier forwards run on stock PyTorch-ROCm (their kernels are a
"next" row); only the interface is part of the synthesis hot path.
"""
from abc import ABC, abstractmethod
from typing import Dict

import torch
import torch.nn as nn


class ClassifierModel(nn.Module, ABC):
    def __init__(self, n_classes: int):
        super().__init__()
        if n_classes < 2:
            raise ValueError("Number of classes must be at least 2.")
        self.n_classes = n_classes

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(batch, ...) -> (batch, n_classes) scores."""

    def get_layer_nparams(self) -> Dict[str, int]:
        out: Dict[str, int] = {}
        for name, p in self.named_parameters():
            if p.requires_grad:
                key = name.split('.')[0]
                out[key] = out.get(key, 0) + p.numel()
        return out

    def get_nparams(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
