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


def _linear_rows(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """``layer(x)`` for a head with a handful of outputs.  Inference on the GPU (no autograd graph wanted: the label pass of
    the synthesis trainer, reference models/synthesis_trainer.py:207-210) goes through ``tl_linear_rows``; training of the
    classifier itself and CPU tensors (BASELINE config C1) keep ``nn.Linear``."""
    w = layer.weight
    if (x.is_cuda and w.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))
            and w.shape[0] <= 64 and x.shape[1] % 4 == 0 and x.shape[0] > 0):
        from .. import _lib
        from .._lib import check, ptr
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        wd = w.detach()
        wd = wd if wd.is_contiguous() else wd.contiguous()
        b = layer.bias
        check(_lib.load().tl_linear_rows(ptr(x), ptr(wd), ptr(b.detach()) if b is not None else None, ptr(out), x.shape[0],
                                         x.shape[1], w.shape[0], x.stride(0), 0, torch.cuda.current_stream().cuda_stream),
              "tl_linear_rows")
        return out
    return layer(x)


def _hidden_layer(layer: nn.Linear, act: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """``act(layer(x))`` of ShallowNNClassifier's hidden layer (reference :112-134).  Inference on the GPU: one launch of the
    NT MFMA GEMM with bias and ReLU / LeakyReLU in its epilogue (other activations: the GEMM, then the module); with
    autograd or on the CPU: the modules."""
    w = layer.weight
    if (x.is_cuda and w.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))
            and layer.bias is not None and x.shape[1] % 4 == 0 and x.shape[0] > 0):
        from .. import _lib
        from .._classifier_engine import _launch_nt
        from .._lib import EPI_LRELU, EPI_STORE, LOAD_DIRECT, ptr
        x = x.contiguous()
        B, K = x.shape
        N = w.shape[0]
        out = torch.empty(B, N, dtype=torch.float32, device=x.device)
        slope = 0.0 if isinstance(act, nn.ReLU) else (float(act.negative_slope) if isinstance(act, nn.LeakyReLU) else None)
        _launch_nt(_lib.load(), A=ptr(x), Bw=ptr(w.detach().contiguous()), bias=ptr(layer.bias.detach()), out=ptr(out), M=B,
                   A_rows=B, N=N, K=K, lda=K, ldb=K, ldo=N, loader=LOAD_DIRECT,
                   epilogue=EPI_STORE if slope is None else EPI_LRELU, slope=slope or 0.0)
        return out if slope is not None else act(out)
    return act(layer(x))


class LogisticRegressionClassifier(ClassifierModel):
    def __init__(self, input_dim: int, n_classes: int):
        super().__init__(n_classes)
        self.input_dim = input_dim
        self.linear = nn.Linear(input_dim, n_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _linear_rows(self.linear, _flatten_checked(x, self.input_dim))


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
        x = _flatten_checked(x, self.input_dim)
        return _linear_rows(self.output, _hidden_layer(self.hidden, self.activation, x))
