"""Deep tone / syllable classifiers (API mirror of reference models/deep_classifiers.py:17-343).

They run forward-only inside every synthesis train step (reference models/synthesis_trainer.py:207-210).
Their kernels are a "next" row of the hot-path scope: this module keeps the
reference's constructor arguments, sub-module names (hence ``state_dict`` keys, so pre-trained
``.pt`` files load) and forward semantics - including the raw ``view`` that re-interprets the
(B, 256, t', w) feature map as (B, t', 256*w) in ``CNNRNNClassifier`` (:315).  Called without autograd on CUDA tensors
(how the trainer calls them) they run on the HIP kernels (``_classifier_engine``) - in eval mode and, with their
``nn.Dropout`` drawn from the package's counter-hash stream, in train mode (the reference CLI's default,
train_synthesizer.py:275-284); only a call that needs autograd THROUGH the classifier uses the module graph.
"""
from __future__ import annotations

import torch
from torch import nn

from .classifier import ClassifierModel


def _temporal_length(layers, n: int) -> int:
    for layer in layers:
        if isinstance(layer, nn.Conv2d):
            k, s, p = layer.kernel_size[0], layer.stride[0], layer.padding[0]
        elif isinstance(layer, nn.MaxPool2d):
            k = layer.kernel_size[0] if isinstance(layer.kernel_size, tuple) else layer.kernel_size
            s = layer.stride[0] if isinstance(layer.stride, tuple) else (layer.stride or k)
            p = layer.padding[0] if isinstance(layer.padding, tuple) else layer.padding
        else:
            continue
        n = (n + 2 * p - k) // s + 1
    return n


class CNNClassifier(ClassifierModel):
    """Six (3,1) convolutions over time shared across electrodes, five (2,1) max-pools, two Linear
    layers and a sigmoid (reference :17-155)."""

    def __init__(self, input_channels: int, input_length: int, n_classes: int, dropout_rate: float = 0.5,
                 negative_slope: float = 0.01) -> None:
        super().__init__(n_classes)
        if input_channels <= 0:
            raise ValueError("Input channels must be a positive integer.")
        widths = [(1, 512, True), (512, 512, True), (512, 512, True), (512, 512, True), (512, 512, False),
                  (512, 256, True)]
        layers = []
        for cin, cout, pool in widths:
            layers += [nn.Conv2d(cin, cout, kernel_size=(3, 1)), nn.LeakyReLU(negative_slope=negative_slope)]
            if pool:
                layers.append(nn.MaxPool2d(kernel_size=(2, 1)))
        layers.append(nn.Dropout(dropout_rate))
        self.feature_extractor = nn.Sequential(*layers)
        self.latent_length = _temporal_length(self.feature_extractor, input_length)
        if self.latent_length <= 0:
            raise ValueError("Input length is too small for the convolutional layers. "
                             "Please increase the input length or adjust the model architecture.")
        self.classifier = nn.Sequential(
            nn.Flatten(), nn.Linear(256 * input_channels * self.latent_length, 1024),
            nn.LeakyReLU(negative_slope=negative_slope), nn.Linear(1024, n_classes), nn.Sigmoid())

        self._hip = None
        self._hip_cfg = (input_channels, input_length, negative_slope)
        self._drop_calls = 0

    def _next_seed(self) -> int:
        self._drop_calls += 1
        return (torch.initial_seed() * 0x9E3779B1 + 0x51ED27 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF

    def _hip_engine(self):
        if self._hip is None:
            from .._classifier_engine import CnnClassifierEngine
            stage_defs, convs = [], [m for m in self.feature_extractor if isinstance(m, nn.Conv2d)]
            for m in self.feature_extractor:
                if isinstance(m, nn.Conv2d):
                    stage_defs.append([m.out_channels, m.kernel_size[0], False])
                elif isinstance(m, nn.MaxPool2d):
                    stage_defs[-1][2] = True
            self._hip = CnnClassifierEngine(self._hip_cfg[0], self._hip_cfg[1], [tuple(s) for s in stage_defs],
                                            self.classifier[1].out_features, self.n_classes, self._hip_cfg[2])
            assert self._hip.lat == self.latent_length and len(convs) == len(stage_defs)
        return self._hip

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # Inference on the GPU (how the synthesis trainer calls it): hand-written HIP path.  Anything
        # that needs autograd through the classifier (its own training is outside the hot-path
        # scope) uses the module graph on stock PyTorch-ROCm.
        needs_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        p_drop = float(self.feature_extractor[-1].p) if self.training else 0.0
        if x.is_cuda and not needs_graph and p_drop < 1.0 and self._hip_cfg[2] >= 0:
            convs = [(m.weight.detach(), m.bias.detach()) for m in self.feature_extractor if isinstance(m, nn.Conv2d)]
            fc1, fc2 = self.classifier[1], self.classifier[3]
            self._last_seed = self._next_seed() if p_drop > 0 else 0
            return self._hip_engine().forward_scores(convs, (fc1.weight.detach(), fc1.bias.detach()),
                                                     (fc2.weight.detach(), fc2.bias.detach()), x, p_drop, self._last_seed)
        x = x.unsqueeze(1).permute(0, 1, 3, 2)            # (B, 1, T, C)
        return self.classifier(self.feature_extractor(x))


class CNNRNNClassifier(ClassifierModel):
    """LSTM over electrodes in parallel with a (7,1) conv, concatenation, two more convs and a
    second LSTM (reference :158-343)."""

    def __init__(self, input_channels: int, input_length: int, n_classes: int, lstm_dim: int = 800,
                 dropout: float = 0.5, negative_slope: float = 0.01) -> None:
        super().__init__(n_classes)
        if lstm_dim % input_length != 0:
            raise ValueError(f"lstm_dim ({lstm_dim}) must be divisible by input_length ({input_length}).")
        self.input_channels = input_channels
        self.input_length = input_length
        self.lstm1 = nn.LSTM(input_size=input_channels, hidden_size=lstm_dim, batch_first=True)

        def head():
            return nn.Sequential(nn.Conv2d(1, 1024, kernel_size=(7, 1)), nn.LeakyReLU(negative_slope=negative_slope),
                                 nn.MaxPool2d(kernel_size=(2, 1), stride=(2, 1)))
        self.conv_pool_block1 = head()
        self.conv_pool_block2 = head()
        self.conv_block3 = nn.Sequential(
            nn.Conv2d(1024, 512, kernel_size=(7, 1)), nn.LeakyReLU(negative_slope=negative_slope),
            nn.Conv2d(512, 256, kernel_size=(7, 1)), nn.LeakyReLU(negative_slope=negative_slope),
            nn.MaxPool2d(kernel_size=(3, 1), stride=(3, 1)), nn.Dropout(dropout))
        w = (lstm_dim // input_length) + input_channels
        self.lstm2 = nn.LSTM(input_size=256 * w, hidden_size=512, batch_first=True)
        self.output = nn.Linear(512, n_classes)
        self._hip = None
        self._drop_calls = 0

    def _next_seed(self) -> int:
        self._drop_calls += 1
        return (torch.initial_seed() * 0x9E3779B1 + 0x7A3C11 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, C, T = x.shape
        if C != self.input_channels:
            raise ValueError(f"Expected {self.input_channels} channels, got {C}.")
        if T != self.input_length:
            raise ValueError(f"Expected input length {self.input_length}, got {T}.")
        xt = x.permute(0, 2, 1)                            # (B, T, C)
        needs_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        slope = self.conv_pool_block1[1].negative_slope
        p_drop = float(self.conv_block3[5].p) if self.training else 0.0
        if x.is_cuda and not needs_graph and p_drop < 1.0 and slope >= 0:
            # inference on CUDA (how the synthesis trainer calls the classifiers): everything on the HIP
            # kernels - both LSTMs (LstmInferEngine), the convolutional trunk, the output layer
            if self._hip is None:
                from .._classifier_engine import CnnRnnConvEngine, LstmInferEngine
                self._hip = CnnRnnConvEngine(C, T, self.lstm1.hidden_size, slope)
                self._hip_lstm1 = LstmInferEngine(C, self.lstm1.hidden_size)
                self._hip_lstm2 = LstmInferEngine(self.lstm2.input_size, self.lstm2.hidden_size)
            wb = lambda m: (m.weight.detach(), m.bias.detach())
            lw = lambda m: (m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0)
            h1 = self._hip_lstm1.last_hidden(xt, *lw(self.lstm1))
            self._hip.last_h1 = h1
            self._last_seed = self._next_seed() if p_drop > 0 else 0
            f = self._hip.features(x, h1, wb(self.conv_pool_block1[0]), wb(self.conv_pool_block2[0]),
                                   wb(self.conv_block3[0]), wb(self.conv_block3[2]), p_drop, self._last_seed)
            h2 = self._hip_lstm2.last_hidden(f, *lw(self.lstm2))
            return self._hip.linear(h2, self.output.weight.detach(), self.output.bias.detach(), sigmoid=True)
        h1 = self.lstm1(xt)[0][:, -1, :]                   # (B, lstm_dim)
        a = self.conv_pool_block1(xt.unsqueeze(1))         # (B, 1024, t, C)
        b = self.conv_pool_block2(h1.reshape(B, 1, T, -1))  # (B, 1024, t, lstm_dim // T)
        f = self.conv_block3(torch.cat((b, a), dim=3))     # (B, 256, t', w)
        f = f.contiguous().view(B, f.shape[2], -1)         # raw re-interpretation, as the reference (:315)
        h2 = self.lstm2(f)[0][:, -1, :]
        return torch.sigmoid(self.output(h2))
