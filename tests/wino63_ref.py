"""Reference forms of the F(6,3) operand layouts for the GPU tests (torch, float64) - TEST INFRASTRUCTURE.

The matrices are the exact Cook-Toom construction of oracle/winograd_f63_gate.py for the points 0, +-1, +-2, +-1/2, inf
(csrc/tonal_wino63.hip uses the same)."""
import torch

BT = torch.tensor([[-1, 0, 5.25, 0, -5.25, 0, 1, 0],
                   [0, 1, 1, -4.25, -4.25, 1, 1, 0],
                   [0, -1, 1, 4.25, -4.25, -1, 1, 0],
                   [0, .5, .25, -2.5, -1.25, 2, 1, 0],
                   [0, -.5, .25, 2.5, -1.25, -2, 1, 0],
                   [0, 2, 4, -2.5, -5, .5, 1, 0],
                   [0, -2, 4, 2.5, -5, -.5, 1, 0],
                   [0, -1, 0, 5.25, 0, -5.25, 0, 1]], dtype=torch.float64)


def hex_transform(P, S, Tp, shift=0):
    """P (S*Tp, C) -> V (S*Tp/6, 8, C), float64, channels last: hex H of a sequence = rows 6H+shift .. 6H+shift+7 (zero
    outside the sequence)."""
    C = P.shape[1]
    x = P.double().view(S, Tp, C)
    pad_l = max(0, -shift)
    x = torch.nn.functional.pad(x, (0, 0, pad_l, 8))
    nh = Tp // 6
    idx = (torch.arange(nh, device=P.device)[:, None] * 6 + torch.arange(8, device=P.device)[None, :]) + shift + pad_l
    return torch.einsum("jk,shkc->shjc", BT.to(P.device), x[:, idx, :]).reshape(S * nh, 8, C)


def logical(V):
    """kernel V / Vd (pair layout [hex / 2][C / 8][8][hex % 2][8], held as a (hexes, 8, C) tensor) -> channels last"""
    nh, _, C = V.shape
    return V.reshape(nh // 2, C // 8, 8, 2, 8).permute(0, 3, 2, 1, 4).reshape(nh, 8, C)


def unpool(G, bits, S, tp, tvalid, C):
    """pooled gradient rows (S*tp, C) + arg-max bits -> dZ (S, 2*tp, C), float64, rows from tvalid on zero"""
    g = G.double().view(S, tp, C)
    w = bits.view(S, tp, C // 32)
    sh = torch.arange(32, device=G.device, dtype=torch.int32)
    odd = ((w[..., None] >> sh) & 1).reshape(S, tp, C).bool()
    dz = torch.zeros(S, 2 * tp, C, dtype=torch.float64, device=G.device)
    dz[:, 0::2] = torch.where(odd, torch.zeros_like(g), g)
    dz[:, 1::2] = torch.where(odd, g, torch.zeros_like(g))
    dz[:, tvalid:] = 0
    return dz


AT = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0],
                   [0, 1, -1, 2, -2, .5, -.5, 0],
                   [0, 1, 1, 4, 4, .25, .25, 0],
                   [0, 1, -1, 8, -8, .125, -.125, 0],
                   [0, 1, 1, 16, 16, .0625, .0625, 0],
                   [0, 1, -1, 32, -32, .03125, -.03125, 1]], dtype=torch.float64)


def y_transform(dz, S, Tp):
    """un-pooled gradient rows dz (S*Tp, C) -> Y (S*Tp/6, 8, C) = A dz per hex (A = the transpose of the output transform),
    float64, channels last: the second operand of the F(6,3) weight gradient"""
    C = dz.shape[1]
    x = dz.double().view(S, Tp // 6, 6, C)
    return torch.einsum("kj,shkc->shjc", AT.to(dz.device), x).reshape(S * (Tp // 6), 8, C)
