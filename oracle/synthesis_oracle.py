"""CPU oracle for the synthesis train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch-CPU fp32 tensor arithmetic) of the
reference's synthesis path.  It is *never* imported by the product package
``decode_tonal_langauge_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference (in the
build container only) and checks every function below against it; the resulting
vectors are committed under ``tests/golden/`` and re-checked by
``tests/test_oracle_golden.py`` on every run (the reference itself has no tests and no
golden vectors: SURVEY.md section 4).

Each function cites the reference file:line it restates.  The arithmetic the
reference delegates to torch (Conv2d, LSTM, NAdam ...) is restated from the published
torch semantics: torch==2.5.1 in the reference's requirements.txt:2, 2.10.0 here.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# SynthesisModelCNN  (reference: models/synthesis_models.py:49-198)
# ----------------------------------------------------------------------------------

#: (out_channels, kernel, pool) of ecog_conv_block, reference models/synthesis_models.py:86-105
ECOG_STAGES = ((512, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (None, 1, False))
#: hidden widths of concat_conv_block, reference models/synthesis_models.py:116-131
CONCAT_WIDTHS = (128, 128, 128, 128, None)


def latent_length(n_timepoints: int) -> int:
    """Floor arithmetic of ``_compute_latent_length`` (models/synthesis_models.py:178-198)."""
    t = n_timepoints
    for _, k, pool in ECOG_STAGES:
        t = (t - k) // 1 + 1
        if pool:
            t = (t - 2) // 2 + 1
    return t


def cnn_param_shapes(output_dim: int, n_channels: int, n_timepoints: int = 200,
                     lstm_channels: int = 6, conv_channels: int = 64) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict names/shapes in registration order (models/synthesis_models.py:86-135)."""
    lat = latent_length(n_timepoints)
    hid = lat * n_channels * lstm_channels
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cin = 1
    for idx, (cout, k, _pool) in zip((0, 3, 6, 9, 12), ECOG_STAGES):
        cout = conv_channels if cout is None else cout
        shapes[f"ecog_conv_block.{idx}.weight"] = (cout, cin, k, 1)
        shapes[f"ecog_conv_block.{idx}.bias"] = (cout,)
        cin = cout
    shapes["label_lstm.weight_ih_l0"] = (4 * hid, 2)
    shapes["label_lstm.weight_hh_l0"] = (4 * hid, hid)
    shapes["label_lstm.bias_ih_l0"] = (4 * hid,)
    shapes["label_lstm.bias_hh_l0"] = (4 * hid,)
    cin = conv_channels + lstm_channels
    for idx, cout in zip((0, 2, 4, 6, 8), CONCAT_WIDTHS):
        cout = conv_channels if cout is None else cout
        shapes[f"concat_conv_block.{idx}.weight"] = (cout, cin, 1, 1)
        shapes[f"concat_conv_block.{idx}.bias"] = (cout,)
        cin = cout
    shapes["output_layer.weight"] = (output_dim, conv_channels * lat * n_channels)
    shapes["output_layer.bias"] = (output_dim,)
    return shapes


def init_cnn_params(output_dim: int, n_channels: int, n_timepoints: int = 200,
                    lstm_channels: int = 6, conv_channels: int = 64) -> "OrderedDict[str, torch.Tensor]":
    """Draw parameters with the same torch calls, in the same order, as the reference
    constructor (Conv2d x5 -> LSTM -> Conv2d x5 -> Linear, models/synthesis_models.py:86-135)
    so that an identical ``torch.manual_seed`` gives identical weights."""
    shapes = cnn_param_shapes(output_dim, n_channels, n_timepoints, lstm_channels, conv_channels)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    names = list(shapes)
    # ecog convs
    for i in range(5):
        w = shapes[names[2 * i]]
        conv = torch.nn.Conv2d(w[1], w[0], kernel_size=(w[2], 1))
        out[names[2 * i]] = conv.weight.detach().clone()
        out[names[2 * i + 1]] = conv.bias.detach().clone()
    hid = shapes["label_lstm.weight_hh_l0"][1]
    lstm = torch.nn.LSTM(input_size=2, hidden_size=hid, batch_first=True)
    for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        out[f"label_lstm.{n}"] = getattr(lstm, n).detach().clone()
    del lstm
    for i in range(5):
        w = shapes[names[14 + 2 * i]]
        conv = torch.nn.Conv2d(w[1], w[0], kernel_size=(1, 1))
        out[names[14 + 2 * i]] = conv.weight.detach().clone()
        out[names[14 + 2 * i + 1]] = conv.bias.detach().clone()
    w = shapes["output_layer.weight"]
    lin = torch.nn.Linear(w[1], w[0])
    out["output_layer.weight"] = lin.weight.detach().clone()
    out["output_layer.bias"] = lin.bias.detach().clone()
    return out


def lstm_last_hidden(x: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor,
                     b_ih: torch.Tensor, b_hh: torch.Tensor) -> torch.Tensor:
    """One-layer LSTM, zero initial state, torch gate order (i, f, g, o); returns h_L.

    Restates ``nn.LSTM(batch_first=True)`` as used at models/synthesis_models.py:112,164-166
    and :249-252,288-289 (for one layer ``output[:, -1]`` equals ``h_n``)."""
    B, L, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    for t in range(L):
        gates = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = gates.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
    return h


class _ActPoolDecided(torch.autograd.Function):
    """LeakyReLU (+ MaxPool2d((2, 1))) with the reference's forward values (models/synthesis_models.py:88-104,
    118-130) and a backward whose two DISCRETE decisions come from outside: ``pos`` - the (pooled) output is positive, so
    LeakyReLU' is 1, else ``slope`` - and ``odd`` - the second row of the pool pair is the arg-max and receives the
    gradient.  With the decisions torch itself takes (``pos = y > 0``, ``odd = z[2t+1] > z[2t]``) the result is
    ``F.leaky_relu`` + ``F.max_pool2d`` bit for bit; with the decisions another implementation took on pre-activations
    that differ from these by rounding, every remaining difference between the two gradients is arithmetic, not a
    flipped branch on a near-tie.  Test infrastructure (tests/test_gpu_north_star.py)."""

    @staticmethod
    def forward(ctx, z, slope, pos, odd):
        y = F.leaky_relu(z, slope)
        if odd is not None:
            y = F.max_pool2d(y, kernel_size=(2, 1), stride=(2, 1))
        ctx.slope, ctx.rows = slope, z.shape[2]
        ctx.save_for_backward(pos, odd if odd is not None else pos)
        ctx.pooled = odd is not None
        return y

    @staticmethod
    def backward(ctx, g):
        pos, odd = ctx.saved_tensors
        gp = g * torch.where(pos, torch.ones((), dtype=g.dtype), torch.full((), ctx.slope, dtype=g.dtype))
        if not ctx.pooled:
            return gp, None, None, None
        gz = g.new_zeros(g.shape[0], g.shape[1], ctx.rows, g.shape[3])      # a trailing odd row gets no gradient
        n = g.shape[2]
        gz[:, :, 0:2 * n:2] = torch.where(odd, torch.zeros((), dtype=g.dtype), gp)
        gz[:, :, 1:2 * n:2] = torch.where(odd, gp, torch.zeros((), dtype=g.dtype))
        return gz, None, None, None


def own_decisions(z: torch.Tensor, pool: bool) -> Dict[str, torch.Tensor]:
    """The decisions torch takes on pre-activations ``z`` (B, ch, t, C): ``pos`` and, for a pooled stage, ``odd``
    (``max_pool2d`` keeps the FIRST maximum of a tie; LeakyReLU' is ``slope`` at exactly 0)."""
    if not pool:
        return {"pos": z > 0}
    n = z.shape[2] // 2
    a, b = z[:, :, 0:2 * n:2], z[:, :, 1:2 * n:2]
    return {"pos": torch.maximum(a, b) > 0, "odd": b > a}


def cnn_forward(p: Dict[str, torch.Tensor], inputs_ecog: torch.Tensor, inputs_labels: torch.Tensor,
                dropout_mask: Optional[torch.Tensor] = None, negative_slope: float = 0.01,
                return_intermediates: bool = False, decisions: Optional[Dict[str, torch.Tensor]] = None,
                own: Optional[Dict[str, torch.Tensor]] = None):
    """``SynthesisModelCNN.forward`` (models/synthesis_models.py:137-176).

    ``dropout_mask`` (B, conv_channels, latent, C) holds the already scaled keep mask
    (0 or 1/(1-p)); ``None`` = eval mode / dropout 0.

    ``decisions`` (test infrastructure): ``{"ecog<i>.pos", "ecog<i>.odd" (i = 1..4), "ecog5.pos", "concat<i>.pos"}`` bool
    tensors in the layout of the activation they belong to - the LeakyReLU' / arg-max branches the BACKWARD pass takes
    (``_ActPoolDecided``); forward values are unchanged.  Layers without an entry decide for themselves.  ``own``: a dict
    that receives the decisions this forward pass would take by itself (for counting how many differ).
    """
    B, C, T = inputs_ecog.shape
    x = inputs_ecog.unsqueeze(1).permute(0, 1, 3, 2)        # (B, 1, T, C)  :157-158
    inter = {}
    dec = decisions or {}

    def act(z, slope, pool, key):
        if own is not None:
            with torch.no_grad():
                for k, v in own_decisions(z, pool).items():
                    own[f"{key}.{k}"] = v
        if f"{key}.pos" in dec:
            return _ActPoolDecided.apply(z, slope, dec[f"{key}.pos"], dec[f"{key}.odd"] if pool else None)
        y = F.leaky_relu(z, slope)
        return F.max_pool2d(y, kernel_size=(2, 1), stride=(2, 1)) if pool else y

    for si, (idx, (_, _k, pool)) in enumerate(zip((0, 3, 6, 9, 12), ECOG_STAGES)):
        x = F.conv2d(x, p[f"ecog_conv_block.{idx}.weight"], p[f"ecog_conv_block.{idx}.bias"])
        x = act(x, negative_slope, pool, f"ecog{si + 1}")
        inter[f"ecog{si + 1}"] = x
    if dropout_mask is not None:                              # :107,160
        x = x * dropout_mask
    x2 = inputs_labels.permute(0, 2, 1)                       # (B, L, 2)  :164
    h = lstm_last_hidden(x2, p["label_lstm.weight_ih_l0"], p["label_lstm.weight_hh_l0"],
                         p["label_lstm.bias_ih_l0"], p["label_lstm.bias_hh_l0"])
    inter["lstm_h"] = h
    lat = x.shape[2]
    x2 = h.view(B, -1, lat, C)                                # :167
    x = torch.cat((x, x2), dim=1)                             # :170
    for si, idx in enumerate((0, 2, 4, 6, 8)):                # :116-131 (slope 0.1)
        x = F.conv2d(x, p[f"concat_conv_block.{idx}.weight"], p[f"concat_conv_block.{idx}.bias"])
        x = act(x, 0.1, False, f"concat{si + 1}")
        inter[f"concat{si + 1}"] = x
    x = x.flatten(1)                                          # :174
    out = x @ p["output_layer.weight"].t() + p["output_layer.bias"]   # :175
    if return_intermediates:
        return out, inter
    return out


# ----------------------------------------------------------------------------------
# SynthesisLite (reference: models/synthesis_models.py:201-296)
# ----------------------------------------------------------------------------------

def lite_param_shapes(output_dim: int, n_channels: int, n_timepoints: int = 200, label_dim: int = 2,
                      conv_channels: int = 32, lstm_hidden: int = 64) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["ecog_conv.0.weight"] = (conv_channels, n_channels, 5)
    s["ecog_conv.0.bias"] = (conv_channels,)
    s["ecog_conv.1.weight"] = (conv_channels,)
    s["ecog_conv.1.bias"] = (conv_channels,)
    s["ecog_conv.4.weight"] = (conv_channels, conv_channels, 3)
    s["ecog_conv.4.bias"] = (conv_channels,)
    s["ecog_conv.5.weight"] = (conv_channels,)
    s["ecog_conv.5.bias"] = (conv_channels,)
    s["label_lstm.weight_ih_l0"] = (4 * lstm_hidden, label_dim)
    s["label_lstm.weight_hh_l0"] = (4 * lstm_hidden, lstm_hidden)
    s["label_lstm.bias_ih_l0"] = (4 * lstm_hidden,)
    s["label_lstm.bias_hh_l0"] = (4 * lstm_hidden,)
    feat = conv_channels * (n_timepoints // 4) + lstm_hidden
    s["fc.1.weight"] = (512, feat)
    s["fc.1.bias"] = (512,)
    s["fc.3.weight"] = (output_dim, 512)
    s["fc.3.bias"] = (output_dim,)
    return s


def init_lite_params(output_dim: int, n_channels: int, n_timepoints: int = 200, label_dim: int = 2,
                     conv_channels: int = 32, lstm_hidden: int = 64):
    """Same constructor order as models/synthesis_models.py:236-263.  Returns (params, buffers)."""
    p: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    b: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    c0 = torch.nn.Conv1d(n_channels, conv_channels, kernel_size=5, padding=2)
    bn0 = torch.nn.BatchNorm1d(conv_channels)
    c1 = torch.nn.Conv1d(conv_channels, conv_channels, kernel_size=3, padding=1)
    bn1 = torch.nn.BatchNorm1d(conv_channels)
    lstm = torch.nn.LSTM(input_size=label_dim, hidden_size=lstm_hidden, batch_first=True)
    feat = conv_channels * (n_timepoints // 4) + lstm_hidden
    fc1 = torch.nn.Linear(feat, 512)
    fc3 = torch.nn.Linear(512, output_dim)
    for name, mod in (("ecog_conv.0", c0), ("ecog_conv.1", bn0), ("ecog_conv.4", c1), ("ecog_conv.5", bn1)):
        p[f"{name}.weight"] = mod.weight.detach().clone()
        p[f"{name}.bias"] = mod.bias.detach().clone()
    for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        p[f"label_lstm.{n}"] = getattr(lstm, n).detach().clone()
    p["fc.1.weight"] = fc1.weight.detach().clone()
    p["fc.1.bias"] = fc1.bias.detach().clone()
    p["fc.3.weight"] = fc3.weight.detach().clone()
    p["fc.3.bias"] = fc3.bias.detach().clone()
    for name in ("ecog_conv.1", "ecog_conv.5"):
        b[f"{name}.running_mean"] = torch.zeros(conv_channels)
        b[f"{name}.running_var"] = torch.ones(conv_channels)
        b[f"{name}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return p, b


def batchnorm1d(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                running_mean: torch.Tensor, running_var: torch.Tensor, training: bool,
                momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """``nn.BatchNorm1d`` on (B, C, T): batch statistics over (B, T) when training
    (biased variance for normalisation, unbiased for the running estimate), running
    statistics in eval.  Updates the running buffers in place like torch does."""
    if training:
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        n = x.shape[0] * x.shape[2]
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(momentum * mean.detach())
            running_var.mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
    else:
        mean, var = running_mean, running_var
    xn = (x - mean[None, :, None]) * torch.rsqrt(var[None, :, None] + eps)
    return xn * gamma[None, :, None] + beta[None, :, None]


def lite_forward(p: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor], x_ecog: torch.Tensor,
                 x_label: torch.Tensor, training: bool = False,
                 dropout_mask: Optional[torch.Tensor] = None, negative_slope: float = 0.01) -> torch.Tensor:
    """``SynthesisLite.forward`` (models/synthesis_models.py:265-296)."""
    x = F.conv1d(x_ecog, p["ecog_conv.0.weight"], p["ecog_conv.0.bias"], padding=2)
    x = batchnorm1d(x, p["ecog_conv.1.weight"], p["ecog_conv.1.bias"],
                    b["ecog_conv.1.running_mean"], b["ecog_conv.1.running_var"], training)
    x = F.max_pool1d(F.leaky_relu(x, negative_slope), 2)
    x = F.conv1d(x, p["ecog_conv.4.weight"], p["ecog_conv.4.bias"], padding=1)
    x = batchnorm1d(x, p["ecog_conv.5.weight"], p["ecog_conv.5.bias"],
                    b["ecog_conv.5.running_mean"], b["ecog_conv.5.running_var"], training)
    x = F.max_pool1d(F.leaky_relu(x, negative_slope), 2)
    feat = x.flatten(1)
    h = lstm_last_hidden(x_label.permute(0, 2, 1), p["label_lstm.weight_ih_l0"],
                         p["label_lstm.weight_hh_l0"], p["label_lstm.bias_ih_l0"],
                         p["label_lstm.bias_hh_l0"])
    z = torch.cat([feat, h], dim=-1)
    if dropout_mask is not None:                              # fc.0 Dropout(0.3)
        z = z * dropout_mask
    z = F.leaky_relu(z @ p["fc.1.weight"].t() + p["fc.1.bias"], negative_slope)
    return z @ p["fc.3.weight"].t() + p["fc.3.bias"]


# ----------------------------------------------------------------------------------
# Loss, metric, optimizer  (reference: models/synthesis_trainer.py:14-43, 131-140)
# ----------------------------------------------------------------------------------

def l1_loss(outputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """``nn.L1Loss()`` = mean |o - t| over all elements (models/synthesis_trainer.py:140,225)."""
    return (outputs - targets.to(outputs.dtype)).abs().mean()


def compute_mcd(true_mcc: torch.Tensor, pred_mcc: torch.Tensor) -> float:
    """models/synthesis_trainer.py:14-43: mean_b(10/ln10 * sqrt(2 * sum_k (t-p)^2))."""
    t = true_mcc.float()
    q = pred_mcc.float()
    sq = ((t - q) ** 2).sum(dim=1)
    return float((10.0 / math.log(10.0) * torch.sqrt(2.0 * sq)).mean())


class NAdamState:
    """Per-parameter state of torch.optim.NAdam (step, mu_product, exp_avg, exp_avg_sq)."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.step = 0
        self.mu_product = 1.0
        self.exp_avg = {k: torch.zeros_like(v) for k, v in params.items()}
        self.exp_avg_sq = {k: torch.zeros_like(v) for k, v in params.items()}


def nadam_scalars(step: int, mu_product: float, lr: float, beta1: float, beta2: float,
                  momentum_decay: float) -> Tuple[float, float, float, float]:
    """Scalar schedule of ``torch.optim.nadam._single_tensor_nadam``: returns
    (coef_grad, coef_mom, bias_correction2, new mu_product) for 1-based ``step``."""
    bc2 = 1.0 - beta2 ** step
    mu = beta1 * (1.0 - 0.5 * (0.96 ** (step * momentum_decay)))
    mu_next = beta1 * (1.0 - 0.5 * (0.96 ** ((step + 1) * momentum_decay)))
    mu_product = mu_product * mu
    coef_grad = lr * (1.0 - mu) / (1.0 - mu_product)
    coef_mom = lr * mu_next / (1.0 - mu_product * mu_next)
    return coef_grad, coef_mom, bc2, mu_product


def nadam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: NAdamState,
               lr: float = 5e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.004, momentum_decay: float = 0.004) -> None:
    """In-place NAdam update as constructed at models/synthesis_trainer.py:131-137:
    coupled L2 ``weight_decay`` (the reference's ``schedule_decay`` lands there) and torch's
    default ``momentum_decay=0.004``; restates ``_single_tensor_nadam``."""
    beta1, beta2 = betas
    state.step += 1
    cg, cm, bc2, state.mu_product = nadam_scalars(state.step, state.mu_product, lr, beta1, beta2,
                                                  momentum_decay)
    with torch.no_grad():
        for k, prm in params.items():
            g = grads[k]
            if weight_decay != 0:
                g = g + weight_decay * prm
            m = state.exp_avg[k]
            v = state.exp_avg_sq[k]
            m.lerp_(g, 1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            denom = (v / bc2).sqrt() + eps
            prm.addcdiv_(g, denom, value=-cg)
            prm.addcdiv_(m, denom, value=-cm)


# ----------------------------------------------------------------------------------
# Host-side pieces of the train loop
# ----------------------------------------------------------------------------------

def prepare_tone_dynamics(tone_dynamic_mapping: Dict[str, List[int]], tone_labels: Sequence[int],
                          syllable_labels: Sequence[int]) -> np.ndarray:
    """data_loading/utils.py:32-79: (B, 2, L) array [[syllable]*L, mapping[str(tone)]]."""
    if len(tone_labels) != len(syllable_labels):
        raise ValueError("Length of tone labels and syllable labels must match.")
    rows = []
    for tone, syl in zip(tone_labels, syllable_labels):
        key = str(int(tone))
        if key not in tone_dynamic_mapping:
            raise ValueError(f"Tone {key} not found in tone_dynamic_mapping.")
        dyn = list(tone_dynamic_mapping[key])
        rows.append([[int(syl)] * len(dyn), dyn])
    return np.array(rows)


def split_indices(n_samples: int, ratios: Sequence[float], seed: int) -> List[List[int]]:
    """Index lists produced by ``split_dataset`` (data_loading/dataloaders.py:43-60):
    ``torch.manual_seed(seed)``; sizes int(n*r) with the remainder to the last split;
    ``random_split`` = one ``randperm(n)`` cut consecutively."""
    sizes: List[int] = []
    for i, r in enumerate(ratios):
        if r <= 0 or r >= 1:
            raise ValueError("All ratios must be between 0 and 1 (exclusive).")
        sizes.append(n_samples - sum(sizes) if i == len(ratios) - 1 else int(n_samples * r))
    torch.manual_seed(seed)
    perm = torch.randperm(n_samples, generator=torch.default_generator).tolist()
    out, ofs = [], 0
    for s in sizes:
        out.append(perm[ofs:ofs + s])
        ofs += s
    return out


def train_step(kind: str, params: Dict[str, torch.Tensor], buffers: Optional[Dict[str, torch.Tensor]],
               state: NAdamState, inputs_non: torch.Tensor, inputs_label: torch.Tensor,
               targets: torch.Tensor, lr: float = 5e-4, weight_decay: float = 0.004,
               dropout_mask: Optional[torch.Tensor] = None, return_grads: bool = False):
    """One body of the batch loop of ``SynthesisTrainer.train`` after the labels are built
    (models/synthesis_trainer.py:220-229): forward, L1 against *integer-truncated* targets
    (:222), backward, NAdam step; returns (loss, mcd[, grads, outputs])."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in params.items()}
    if kind == "cnn":
        out = cnn_forward(leaves, inputs_non, inputs_label, dropout_mask=dropout_mask)
    elif kind == "lite":
        out = lite_forward(leaves, buffers, inputs_non, inputs_label, training=True,
                           dropout_mask=dropout_mask)
    else:
        raise ValueError(kind)
    tgt = targets.long()                                       # :222
    loss = l1_loss(out, tgt)
    grads_list = torch.autograd.grad(loss, list(leaves.values()))
    grads = dict(zip(leaves.keys(), grads_list))
    nadam_step(params, grads, state, lr=lr, weight_decay=weight_decay)
    mcd = compute_mcd(tgt, out.detach())
    if return_grads:
        return float(loss.detach()), mcd, grads, out.detach()
    return float(loss.detach()), mcd
