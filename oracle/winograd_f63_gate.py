"""Numerics gate for a Winograd F(6,3) form of the pooled 3-tap stages (round-4 review item 4) - TEST INFRASTRUCTURE, CPU only.

F(6,3) needs 8 products per 6 outputs (0.444 of the direct convolution's multiplies; the F(4,3) kernels of this package
issue 0.5): about -9 % matrix-pipe work on 80 % of the train step, IF fp32 survives its transforms - their constants
reach 21/4 and 32 and the interpolation points +-2, +-1/2 amplify rounding.  This script measures that before any kernel
is written, in emulation: every transform, the channel contraction and the output transform run in float32 exactly as a
kernel would order them (transform the input tile, multiply by the pre-transformed taps, accumulate over the input
channels, transform the output), and autograd through that graph yields the transposed forms a backward pass would use.

  1. per-stage error: conv2 / conv3 of ``SynthesisModelCNN`` (models/synthesis_models.py:91-97) at their real widths
     (512 -> 512 channels) on activations produced by the real stack from N(0,1) ECoG, seeded default-init weights;
     relative L2 and max-norm error against an fp64 direct convolution, for: direct fp32 (torch), F(4,3) (the product's
     form), F(6,3) with the points (0, +-1, +-2, +-1/2, inf) and with an alternative set (0, +-1, +-1/2, +-3/2?, inf).
  2. trajectory: the 30-step NAdam trajectory of golden G14 (16 x 200) with conv2 / conv3 replaced by the emulated forms,
     deviation of loss / mel MSE per step from the stored reference trajectory.

Gate (review): per-stage relative error <= 2e-5 AND 30-step mel-MSE deviation <= 2e-4 -> build kernels; else rejected.

    python oracle/winograd_f63_gate.py [--skip-traj]
"""
from __future__ import annotations

import argparse
import os
import sys
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def cook_toom(m: int, r: int, points):
    """Exact (Fraction) matrices At (m x n), G (n x r), Bt (n x n), n = m + r - 1, for the finite ``points`` plus infinity:
    y = At [(G g) * (Bt d)] is the valid correlation y[i] = sum_k g[k] d[i + k] (Lavin & Gray's construction)."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    At = [[(pts[j] ** i if j < n - 1 else (1 if i == m - 1 else 0)) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fraction(1)
        for l in range(n - 1):
            if l != j:
                f *= (pts[j] - pts[l])
        G.append([pts[j] ** k / f for k in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])

    def polymul(a, b):
        out = [Fraction(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    Bt = []
    for j in range(n - 1):                      # row j: coefficients of prod_{l != j} (x - p_l), degree n - 2
        poly = [Fraction(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fraction(1)])
        Bt.append(poly + [Fraction(0)] * (n - len(poly)))
    poly = [Fraction(1)]
    for l in range(n - 1):                      # last row: prod_l (x - p_l), degree n - 1
        poly = polymul(poly, [-pts[l], Fraction(1)])
    Bt.append(poly)
    # exactness check in rational arithmetic: sum_j At[i][j] G[j][k] Bt[j][s] == [s == i + k]
    for i in range(m):
        for k in range(r):
            for s in range(n):
                v = sum(At[i][j] * G[j][k] * Bt[j][s] for j in range(n))
                assert v == (1 if s == i + k else 0), (i, k, s, v)
    f32 = lambda M: torch.tensor([[float(x) for x in row] for row in M], dtype=torch.float32)
    return f32(At), f32(G), f32(Bt)


class WinoConv(torch.nn.Module):
    """(k,1) convolution over the time axis of (B, C_in, T, W) as F(m,3) in float32: tiles of n = m + 2 input rows at stride m."""

    def __init__(self, m: int, points):
        super().__init__()
        self.m, self.n = m, m + 2
        At, G, Bt = cook_toom(m, 3, points)
        self.register_buffer("At", At)
        self.register_buffer("G", G)
        self.register_buffer("Bt", Bt)

    def forward(self, x, w, b):
        Bn, Ci, T, W = x.shape
        Co = w.shape[0]
        tout = T - 2
        nt = -(-tout // self.m)
        pad = nt * self.m + 2 - T
        xp = F.pad(x, (0, 0, 0, pad))
        tiles = xp.unfold(2, self.n, self.m)                       # (B, Ci, nt, W, n)
        V = torch.einsum("js,bctws->bctwj", self.Bt, tiles)       # input transform, fp32
        U = torch.einsum("jk,ock->ocj", self.G, w[:, :, :, 0])    # tap transform, fp32
        M = torch.einsum("bctwj,ocj->botwj", V, U)                # channel contraction per transform point, fp32 accumulate
        Y = torch.einsum("ij,botwj->botwi", self.At, M)           # output transform
        y = Y.permute(0, 1, 2, 4, 3).reshape(Bn, Co, nt * self.m, W)[:, :, :tout]
        return y + b.view(1, -1, 1, 1)


POINT_SETS = {
    "F(4,3) 0,+-1,+-2,inf (the product)": (4, (0, 1, -1, 2, -2)),
    "F(6,3) 0,+-1,+-2,+-1/2,inf (Lavin)": (6, (0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2))),
    "F(6,3) 0,+-1,+-1/2,+-3/2,inf": (6, (0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2))),
    "F(6,3) 0,+-1/2,+-1,+-3/4?": (6, (0, Fraction(1, 2), Fraction(-1, 2), 1, -1, Fraction(3, 4), Fraction(-3, 4))),
}


def stage_errors(report):
    torch.manual_seed(0)
    # the real first three stages at reduced extent: 6 windows x 8 ECoG channels x 400 samples, real channel widths
    Bn, C, T = 6, 8, 400
    x = torch.randn(Bn, 1, T, C)
    c1 = torch.nn.Conv2d(1, 512, (3, 1))
    c2 = torch.nn.Conv2d(512, 512, (3, 1))
    c3 = torch.nn.Conv2d(512, 512, (3, 1))
    with torch.no_grad():
        a1 = F.max_pool2d(F.leaky_relu(c1(x), 0.01), (2, 1), (2, 1))
        ref2 = F.conv2d(a1.double(), c2.weight.double(), c2.bias.double())
        a2 = F.max_pool2d(F.leaky_relu(ref2, 0.01), (2, 1), (2, 1)).float()
        ref3 = F.conv2d(a2.double(), c3.weight.double(), c3.bias.double())
        rows = []
        for name, (inp, conv, ref) in (("conv2", (a1, c2, ref2)), ("conv3", (a2, c3, ref3))):
            d = F.conv2d(inp, conv.weight, conv.bias).double()
            rel = lambda y: (float((y - ref).norm() / ref.norm()), float((y - ref).abs().max() / ref.abs().max()))
            rows.append((name, "direct fp32 (torch)", *rel(d)))
            for label, (m, pts) in POINT_SETS.items():
                y = WinoConv(m, pts)(inp, conv.weight, conv.bias).double()
                rows.append((name, label, *rel(y)))
    for r in rows:
        print(f"{r[0]:6s} {r[1]:42s} rel L2 {r[2]:.3e}   max-norm {r[3]:.3e}")
    report["stage"] = rows


def trajectory(report):
    import golden_inputs as gi
    from oracle import synthesis_oracle as so
    g = np.load(os.path.join(REPO, "tests", "golden", "g14_cnn_trajectory.npz"))
    D, C, T, B, steps = (int(v) for v in g["dims"])
    xs, _t, _s, labs, tg = gi.train_batches(steps, B, C, T, seed=int(g["data_seed"]))
    out = {}
    for label in ("direct fp32 (torch)",) + tuple(POINT_SETS):
        wino = None if label.startswith("direct") else WinoConv(*POINT_SETS[label])
        orig = F.conv2d

        def conv2d(x, w, b=None, *a, **k):                       # conv2 / conv3: 512 -> 512, three taps
            if wino is not None and w.shape[1] == 512 and w.shape[2] == 3 and x.dtype == torch.float32:
                return wino(x, w, b)
            return orig(x, w, b, *a, **k)
        torch.manual_seed(int(g["seed"]))
        p = so.init_cnn_params(D, C, T)
        st = so.NAdamState(p)
        dev_loss, dev_mse = 0.0, 0.0
        so.F.conv2d = conv2d
        try:
            for s in range(steps):
                l_, _m, _g, o_ = so.train_step("cnn", p, None, st, xs[s], labs[s], tg[s], return_grads=True)
                mse = float(((o_.double() - tg[s].double()) ** 2).mean())
                dev_loss = max(dev_loss, abs(l_ - float(g["losses"][s])) / float(g["losses"][s]))
                dev_mse = max(dev_mse, abs(mse - float(g["mses"][s])) / float(g["mses"][s]))
        finally:
            so.F.conv2d = orig
        out[label] = (dev_loss, dev_mse)
        print(f"trajectory {label:42s} max loss dev {dev_loss:.3e}   max mel-MSE dev {dev_mse:.3e}", flush=True)
    report["traj"] = out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-traj", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    report = {}
    stage_errors(report)
    if not args.skip_traj:
        trajectory(report)


if __name__ == "__main__":
    main()
