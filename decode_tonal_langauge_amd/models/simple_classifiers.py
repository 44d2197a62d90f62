"""Benchmark classifiers (mirror of reference models/simple_classifiers.py:9-134).

Same constructor arguments, parameter names (``linear``, ``hidden``, ``output``) and error
behaviour, so ``state_dict``s interchange with the reference."""
from typing import Optional

import torch
import torch.nn as nn

from .classifier import ClassifierModel
from .utils import get_activation


def _flatten_checked(x: torch.Tensor, input_dim: int) -> torch.Tensor:
    if x.ndim > 2:
        x = x.reshape(x.size(0), -1)
    if x.shape[1] != input_dim:
        raise ValueError(f"Expected input dimension {input_dim}, got {x.shape[1]}.")
    return x


class LogisticRegressionClassifier(ClassifierModel):
    def __init__(self, input_dim: int, n_classes: int):
        super().__init__(n_classes)
        self.input_dim = input_dim
        self.linear = nn.Linear(input_dim, n_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear(_flatten_checked(x, self.input_dim))


class ShallowNNClassifier(ClassifierModel):
    def __init__(self, input_dim: int, n_classes: int, hidden_dim: Optional[int] = None, activation: str = 'ReLU'):
        super().__init__(n_classes)
        self.input_dim = input_dim
        if hidden_dim is None:
            hidden_dim = input_dim // 2
        self.hidden = nn.Linear(input_dim, hidden_dim)
        self.output = nn.Linear(hidden_dim, n_classes)
        self.activation = get_activation(activation)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.output(self.activation(self.hidden(_flatten_checked(x, self.input_dim))))
