"""Synthesis models on MI355X (mirror of reference models/synthesis_models.py).

``SynthesisModel`` keeps the reference's plugin interface (:7-46): ``forward(inputs_non (B,C,T),
inputs_label (B,2,L)) -> (B, output_dim)`` and ``get_nparams()``.  ``SynthesisModelCNN`` (:49-198)
and ``SynthesisLite`` (:201-296) keep constructor signatures, sub-module / parameter names and
the construction order (hence identical weights for an identical ``torch.manual_seed``), so
``state_dict``s interchange with the reference.  The sub-modules are parameter containers only:
``forward`` runs the hand-written HIP kernels of ``libtonal_hip.so`` through a
``torch.autograd.Function``; calling it on CPU tensors raises (no fallback).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List

import torch
from torch import nn

from .. import _lib
from .._cnn_engine import CnnEngine


class SynthesisModel(nn.Module, ABC):
    """Plugin base class used by ``SynthesisTrainer`` (reference models/synthesis_models.py:7-46)."""

    @abstractmethod
    def forward(self, inputs_non: torch.Tensor, inputs_label: torch.Tensor) -> torch.Tensor:
        """(B, C, T) ECoG windows, (B, 2, L) label dynamics -> (B, output_dim)."""

    def get_nparams(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class _CnnFunction(torch.autograd.Function):
    """Shared by both models: ``model._engine`` does the work on detached tensors."""

    @staticmethod
    def forward(ctx, x, labels, model, need_grad, *params):
        eng = model._engine
        prm = dict(zip(model._pnames, (p.detach() for p in params)))
        prm.update(model._engine_buffers())
        out = eng.forward(prm, x, labels, training=model.training, save=need_grad, seed=model._next_seed())
        ctx.model = model
        ctx.prm = prm
        ctx.generation = eng.generation
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        eng = model._engine
        if ctx.generation != eng.generation:
            raise RuntimeError("SynthesisModelCNN: backward after a newer forward; intermediates were overwritten")
        dout = dout.contiguous().float()
        if eng.ldd != eng.out_dim:
            pad = torch.zeros(dout.shape[0], eng.ldd, dtype=dout.dtype, device=dout.device)
            pad[:, :eng.out_dim] = dout
            dout = pad
        grads = {k: torch.empty_like(ctx.prm[k]) for k in model._pnames}
        eng.backward(ctx.prm, dout, grads)
        return (None, None, None, None) + tuple(grads[k] for k in model._pnames)


class SynthesisModelCNN(SynthesisModel):
    """Paper model: shared-weight (k,1) conv stack per ECoG channel, label LSTM, 1x1 conv stack,
    Linear head (reference models/synthesis_models.py:49-198)."""

    def __init__(self, output_dim: int, n_channels: int, n_timepoints: int = 200, lstm_channels: int = 6,
                 conv_channels: int = 64, dropout: float = 0.5, negative_slope: float = 0.01):
        super().__init__()
        self.n_channels = n_channels
        self.n_timepoints = n_timepoints
        self.conv_channels = conv_channels
        self.lstm_channels = lstm_channels
        # Same modules, same order as the reference (:86-135): identical seeds give identical
        # weights and identical state_dict keys.  Their .forward is never used.
        self.ecog_conv_block = nn.Sequential(
            nn.Conv2d(1, 512, kernel_size=(3, 1)), nn.LeakyReLU(negative_slope), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(512, 512, kernel_size=(3, 1)), nn.LeakyReLU(negative_slope), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(512, 512, kernel_size=(3, 1)), nn.LeakyReLU(negative_slope), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(512, 256, kernel_size=(1, 1)), nn.LeakyReLU(negative_slope), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(256, conv_channels, kernel_size=(1, 1)), nn.LeakyReLU(negative_slope),
        )
        self.ecog_dropout = nn.Dropout(dropout)
        self.latent_len = self._compute_latent_length(n_timepoints)
        lstm_size = self.latent_len * n_channels * lstm_channels
        self.label_lstm = nn.LSTM(input_size=2, hidden_size=lstm_size, batch_first=True)
        total = conv_channels + lstm_channels
        self.concat_conv_block = nn.Sequential(
            nn.Conv2d(total, 128, kernel_size=(1, 1)), nn.LeakyReLU(0.1),
            nn.Conv2d(128, 128, kernel_size=(1, 1)), nn.LeakyReLU(0.1),
            nn.Conv2d(128, 128, kernel_size=(1, 1)), nn.LeakyReLU(0.1),
            nn.Conv2d(128, 128, kernel_size=(1, 1)), nn.LeakyReLU(0.1),
            nn.Conv2d(128, conv_channels, kernel_size=(1, 1)), nn.LeakyReLU(0.1),
        )
        self.flatten = nn.Flatten()
        self.output_layer = nn.Linear(conv_channels * self.latent_len * n_channels, output_dim)

        stage_defs = []
        for layer in self.ecog_conv_block:
            if isinstance(layer, nn.Conv2d):
                stage_defs.append([layer.out_channels, layer.kernel_size[0], False])
            elif isinstance(layer, nn.MaxPool2d):
                stage_defs[-1][2] = True
        concat_widths = [m.out_channels for m in self.concat_conv_block if isinstance(m, nn.Conv2d)]
        if lstm_size % 4 != 0:
            raise ValueError("latent_len * n_channels * lstm_channels must be a multiple of 4 on the MI355X path")
        self._pnames: List[str] = [n for n, _ in self.named_parameters()]
        self._engine = CnnEngine(output_dim, n_channels, n_timepoints, lstm_channels, conv_channels, dropout,
                                 negative_slope, [tuple(s) for s in stage_defs], concat_widths)
        assert self._engine.lat == self.latent_len
        self._drop_calls = 0

    def _next_seed(self) -> int:
        self._drop_calls += 1
        return (torch.initial_seed() * 0x9E3779B1 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF

    def _engine_buffers(self):
        return {}

    def forward(self, inputs_ecog: torch.Tensor, inputs_labels: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(inputs_ecog, "SynthesisModelCNN.forward")
        params = [p for _, p in self.named_parameters()]
        _lib.require_gpu(params[0], "SynthesisModelCNN parameters")
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _CnnFunction.apply(inputs_ecog, inputs_labels.to(inputs_ecog.device), self, need_grad, *params)

    def _compute_latent_length(self, n_timepoints: int) -> int:
        t = n_timepoints
        for layer in self.ecog_conv_block:
            if isinstance(layer, nn.Conv2d):
                t = (t - layer.kernel_size[0] + 2 * layer.padding[0]) // layer.stride[0] + 1
            elif isinstance(layer, nn.MaxPool2d):
                t = (t - layer.kernel_size[0]) // layer.stride[0] + 1
        return t


class SynthesisLite(SynthesisModel):
    """Light model: Conv1d+BatchNorm1d+LeakyReLU+MaxPool1d x2, label LSTM, two Linear layers
    (reference models/synthesis_models.py:201-296)."""

    def __init__(self, output_dim: int, n_channels: int, n_timepoints: int = 200, label_dim: int = 2,
                 conv_channels: int = 32, lstm_hidden: int = 64, dropout: float = 0.3,
                 negative_slope: float = 0.01) -> None:
        super().__init__()
        self.ecog_conv = nn.Sequential(
            nn.Conv1d(n_channels, conv_channels, kernel_size=5, padding=2), nn.BatchNorm1d(conv_channels),
            nn.LeakyReLU(negative_slope), nn.MaxPool1d(kernel_size=2),
            nn.Conv1d(conv_channels, conv_channels, kernel_size=3, padding=1), nn.BatchNorm1d(conv_channels),
            nn.LeakyReLU(negative_slope), nn.MaxPool1d(kernel_size=2),
        )
        self.ecog_out_dim = conv_channels * (n_timepoints // 4)
        self.label_lstm = nn.LSTM(input_size=label_dim, hidden_size=lstm_hidden, batch_first=True,
                                  bidirectional=False)
        self.fc = nn.Sequential(
            nn.Dropout(dropout), nn.Linear(self.ecog_out_dim + lstm_hidden, 512), nn.LeakyReLU(negative_slope),
            nn.Linear(512, output_dim),
        )
        if lstm_hidden % 4 != 0:
            raise ValueError("lstm_hidden must be a multiple of 4 on the MI355X path")
        from .._lite_engine import LiteEngine
        self._pnames: List[str] = [n for n, _ in self.named_parameters()]
        self._engine = LiteEngine(output_dim, n_channels, n_timepoints, label_dim, conv_channels, lstm_hidden,
                                  dropout, negative_slope)
        self._drop_calls = 0

    def _next_seed(self) -> int:
        self._drop_calls += 1
        return (torch.initial_seed() * 0x9E3779B1 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF

    def _engine_buffers(self):
        return {k: v for k, v in self.named_buffers()}

    def forward(self, x_ecog: torch.Tensor, x_label: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu(x_ecog, "SynthesisLite.forward")
        params = [p for _, p in self.named_parameters()]
        _lib.require_gpu(params[0], "SynthesisLite parameters")
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _CnnFunction.apply(x_ecog, x_label.to(x_ecog.device), self, need_grad, *params)
