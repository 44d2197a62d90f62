"""Bring-up check of the F(6,3) kernels (TONAL_WINO=6) against the direct MFMA kernels (TONAL_WINO=0), stage by stage.

GPU only.  For a few shapes: conv1 -> V1 (hex form) against a torch transform of the stored raw rows; conv2 forward
(POOLV: raw rows, bits, V2); conv3 forward (POOL with out_tp); conv3 / conv2 weight gradient, bias gradient, Vd, input
gradient and the fused conv1 weight gradient.  Prints relative errors; exit code 1 if any exceeds its bound.

    python scripts/check_wino63.py [--shape B,C,T,c1,c2,c3]
"""
import os
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BT = torch.tensor([[-1, 0, 5.25, 0, -5.25, 0, 1, 0],
                   [0, 1, 1, -4.25, -4.25, 1, 1, 0],
                   [0, -1, 1, 4.25, -4.25, -1, 1, 0],
                   [0, .5, .25, -2.5, -1.25, 2, 1, 0],
                   [0, -.5, .25, 2.5, -1.25, -2, 1, 0],
                   [0, 2, 4, -2.5, -5, .5, 1, 0],
                   [0, -2, 4, 2.5, -5, -.5, 1, 0],
                   [0, -1, 0, 5.25, 0, -5.25, 0, 1]], dtype=torch.float64)


def hex_transform(P, S, Tp, shift=0):
    """P (S*Tp, C) -> V (S*Tp/6, 8, C) in float64: hex H of a sequence = rows 6H+shift .. 6H+shift+7 (zero outside)."""
    C = P.shape[1]
    x = P.double().view(S, Tp, C)
    pad_l = max(0, -shift)
    x = torch.nn.functional.pad(x, (0, 0, pad_l, 8))
    nh = Tp // 6
    idx = (torch.arange(nh, device=P.device)[:, None] * 6 + torch.arange(8, device=P.device)[None, :]) + shift + pad_l
    tiles = x[:, idx, :]                                  # (S, nh, 8, C)
    V = torch.einsum("jk,shkc->shjc", BT.to(P.device), tiles)
    return V.reshape(S * nh, 8, C)


def logical(V):
    """kernel V / Vd (pair layout [hex / 2][C / 8][8][hex % 2][8], stored as (hexes, 8, C)) -> (hexes, 8, C) channels last"""
    nh, _, C = V.shape
    return V.reshape(nh // 2, C // 8, 8, 2, 8).permute(0, 3, 2, 1, 4).reshape(nh, 8, C)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(1e-30, float(b.norm())))


def unpool(G, bits, S, tp, tvalid, C):
    """pooled gradient rows (S*tp, C) + arg-max bits -> dZ (S, 2*tp, C) float64 (rows beyond tvalid zero)."""
    g = G.double().view(S, tp, C)
    w = bits.view(S, tp, C // 32)
    sh = torch.arange(32, device=G.device, dtype=torch.int32)
    odd = ((w[..., None] >> sh) & 1).reshape(S, tp, C).bool()
    dz = torch.zeros(S, 2 * tp, C, dtype=torch.float64, device=G.device)
    dz[:, 0::2] = torch.where(odd, torch.zeros_like(g), g)
    dz[:, 1::2] = torch.where(odd, g, torch.zeros_like(g))
    dz[:, tvalid:] = 0
    return dz


def run(shape, dev):
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    B, C, T, c1, c2, c3 = shape
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    worst = {}

    def note(name, val, bound):
        worst[name] = (val, bound)
        print(f"  {name:34s} {val:.3e}   (bound {bound:.0e}) {'' if val <= bound else '  <-- FAIL'}")

    engs = {}
    for mode in ("0", "6"):
        os.environ["TONAL_WINO"] = mode
        eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
        eng.store_p1 = True
        if mode == "0":
            eng.fuse_c1 = False
        eng._alloc(B, dev)
        eng._alloc_bwd()
        engs[mode] = eng
    e0, e6 = engs["0"], engs["6"]
    assert e6.wino63, "shape not covered by the F(6,3) path"
    S = e6.S
    print(f"shape {shape}: S={S} tp1 {e0.tp1}/{e6.tp1}  stage2 tp_in {e6.stages[0].tp_in} tp_out {e6.stages[0].tp_out}  "
          f"stage3 tp_in {e6.stages[1].tp_in} tp_out {e6.stages[1].tp_out} (direct {e0.stages[1].tp_out})")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, C, T, device=dev, generator=g)
    prm = {}
    names = {1: "ecog_conv_block.0", 2: "ecog_conv_block.3", 3: "ecog_conv_block.6"}
    prm[names[1] + ".weight"] = torch.randn(c1, 1, 3, 1, device=dev, generator=g) * 0.5
    prm[names[1] + ".bias"] = torch.randn(c1, device=dev, generator=g) * 0.1
    cin = c1
    for i, co in ((2, c2), (3, c3)):
        prm[names[i] + ".weight"] = torch.randn(co, cin, 3, 1, device=dev, generator=g) * (1.0 / (3 * cin) ** 0.5)
        prm[names[i] + ".bias"] = torch.randn(co, device=dev, generator=g) * 0.1
        cin = co
    st_ = torch.cuda.current_stream().cuda_stream
    from decode_tonal_langauge_amd._lib import check, ptr
    for eng in (e0, e6):
        eng._x = x.contiguous()
        eng.generation += 1
        eng._v_ready = {}
        w1 = prm[names[1] + ".weight"].reshape(c1, 3).contiguous()
        if eng.wino63:
            V1 = eng._v_hex_buffer(eng.V, 1, S * eng.tp1, c1)
            check(eng.lib.tl_conv1_fwd_v6(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), ptr(eng.P[1]), ptr(V1), ptr(eng.bits[1]),
                                          ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "conv1_v6")
            eng._v_ready[1] = V1
        else:
            check(eng.lib.tl_conv1_fwd(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), ptr(eng.P[1]), ptr(eng.bits[1]),
                                       ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "conv1")
    torch.cuda.synchronize()

    def rows(eng, t, tp, n):      # valid rows of a (S*tp, C) tensor
        return t.view(S, tp, -1)[:, :n]

    tin2 = e6.stages[0].tin
    note("conv1 raw rows", rel(rows(e6, e6.P[1], e6.tp1, tin2), rows(e0, e0.P[1], e0.tp1, tin2)), 1e-6)
    assert torch.equal(rows(e6, e6.bits[1], e6.tp1, tin2), rows(e0, e0.bits[1], e0.tp1, tin2))
    V1ref = hex_transform(e6.P[1], S, e6.tp1)
    note("V1 == B^T P1", rel(logical(e6.V[1])[:V1ref.shape[0]], V1ref), 1e-6)
    assert float(e6.V[1][V1ref.shape[0]:].abs().max()) == 0.0 if e6.V[1].shape[0] > V1ref.shape[0] else True

    # ---- forward stages 2, 3 ----
    for si in (2, 3):
        for eng in (e0, e6):
            st = eng.stages[si - 2]
            eng.stage_forward(st, prm[names[si] + ".weight"], prm[names[si] + ".bias"])
        torch.cuda.synchronize()
        s0, s6 = e0.stages[si - 2], e6.stages[si - 2]
        nv = s6.tout
        note(f"conv{si} pooled rows", rel(rows(e6, e6.P[si], s6.tp_out, nv), rows(e0, e0.P[si], s0.tp_out, nv)), 2e-5)
        if s6.tp_out > nv:
            note(f"conv{si} pad rows (abs max)", float(rows(e6, e6.P[si], s6.tp_out, s6.tp_out)[:, nv:].abs().max()), 0.0)
        fl = rows(e6, e6.bits[si], s6.tp_out, nv) ^ rows(e0, e0.bits[si], s0.tp_out, nv)
        note(f"conv{si} arg-max word flips", float((fl != 0).sum()), 4)
        fl = rows(e6, e6.sbits[si], s6.tp_out, nv) ^ rows(e0, e0.sbits[si], s0.tp_out, nv)
        note(f"conv{si} sign word flips", float((fl != 0).sum()), 4)
        if si == 2:
            V2ref = hex_transform(e6.P[2], S, s6.tp_out)
            note("V2 == B^T P2", rel(logical(e6.V[2])[:V2ref.shape[0]], V2ref), 1e-6)
            # the F(6,3) stage 3 must see the same input as the direct one: copy the direct rows where valid
    # ---- backward: random G3 into both engines ----
    s0, s6 = e0.stages[1], e6.stages[1]
    G3 = torch.randn(S, s6.tp_out, c3, device=dev, generator=g)
    G3[:, s6.tout:] = 0
    e6.G[3].copy_(G3.reshape(-1, c3))
    e0.G[3].view(S, s0.tp_out, c3).zero_()
    e0.G[3].view(S, s0.tp_out, c3)[:, :s6.tout] = G3[:, :s6.tout]
    # the direct engine un-pools with ITS arg-max bits: give it the F(6,3) engine's (ties may differ)
    e0.bits[3].view(S, s0.tp_out, -1)[:, :s6.tout] = e6.bits[3].view(S, s6.tp_out, -1)[:, :s6.tout]
    e0.bits[2].view(S, e0.stages[0].tp_out, -1)[:, :e6.stages[0].tout] = e6.bits[2].view(S, e6.stages[0].tp_out, -1)[:, :e6.stages[0].tout]
    e0.sbits[2].view(S, e0.stages[0].tp_out, -1)[:, :e6.stages[0].tout] = e6.sbits[2].view(S, e6.stages[0].tp_out, -1)[:, :e6.stages[0].tout]
    part = {}
    for si in (3, 2):
        res = {}
        for key, eng in (("0", e0), ("6", e6)):
            st = eng.stages[si - 2]
            w = prm[names[si] + ".weight"]
            gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
            eng.stage_wgrad(st, gw, gb)
            p_ = eng.stage_dgrad(st, w)
            res[key] = (gw, gb, p_)
        torch.cuda.synchronize()
        s0, s6 = e0.stages[si - 2], e6.stages[si - 2]
        note(f"conv{si} weight gradient (rel L2)", rel_l2(res["6"][0], res["0"][0]), 1e-5)
        note(f"conv{si} bias gradient (rel L2)", rel_l2(res["6"][1], res["0"][1]), 1e-5)
        if si in e6.G:
            dz = unpool(e6.G[si], e6.bits[si], S, s6.tp_out, 2 * s6.tout, s6.cout)
            if 2 * s6.tp_out < s6.tp_in:
                dz = torch.nn.functional.pad(dz, (0, 0, 0, s6.tp_in - 2 * s6.tp_out))
            Vdref = hex_transform(dz[:, :s6.tp_in].reshape(-1, s6.cout), S, s6.tp_in, shift=-2)
            note(f"Vd{si} == B^T dZ", rel(logical(e6.Vd[si])[:Vdref.shape[0]], Vdref), 1e-6)
        if si == 3 and e6.f63_yprod:
            # the input gradient of stage 3 wrote Y2 = A dz and Vd2 instead of G2: against the direct engine's G2
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from wino63_ref import y_transform
            nin, b2 = s6.tin, e6.stages[0]
            g2 = torch.zeros(S, s6.tp_in, s6.cin, device=dev)
            g2[:, :nin] = e0.G[2].view(S, s0.tp_in, -1)[:, :nin]
            dz2 = unpool(g2.reshape(-1, s6.cin), e6.bits[2], S, s6.tp_in, 2 * b2.tout, s6.cin).reshape(-1, s6.cin)
            Yref = y_transform(dz2, S, b2.tp_in)
            note("Y2 == A dz2 (from the direct G2)", rel(logical(e6.Yt[2])[:Yref.shape[0]], Yref), 1e-5)
            Vd2ref = hex_transform(dz2, S, b2.tp_in, shift=-2)
            note("Vd2 == B^T dz2 (from the direct G2)", rel(logical(e6.Vd[2])[:Vd2ref.shape[0]], Vd2ref), 1e-5)
            assert float(e6.Yt[2][Yref.shape[0]:].abs().max()) == 0.0 and float(e6.Vd[2][Yref.shape[0]:].abs().max()) == 0.0
        elif si == 3:
            nin = s6.tin
            note("conv3 input gradient (rel L2)", rel_l2(rows(e6, e6.G[2], s6.tp_in, nin), rows(e0, e0.G[2], s0.tp_in, nin)), 1e-5)
            pad = rows(e6, e6.G[2], s6.tp_in, s6.tp_in)[:, nin:]
            print(f"  (conv3 input gradient, pad rows abs max {float(pad.abs().max()):.3e})")
            # stage 2 of both engines continues from the direct engine's G2 (valid rows), pad rows zero
            e6.G[2].view(S, s6.tp_in, -1).zero_()
            e6.G[2].view(S, s6.tp_in, -1)[:, :nin] = e0.G[2].view(S, s0.tp_in, -1)[:, :nin]
        else:
            # conv1 weight / bias gradient: the fused partial sums against tl_conv1_wgrad on the direct engine's G1
            p6 = res["6"][2].sum(0)
            nblk = int(min(2048, S))
            p0 = torch.empty(nblk, 4 * c1, device=dev)
            check(e0.lib.tl_conv1_wgrad(ptr(e0._x), ptr(e0.G[1]), ptr(e0.bits[1]), ptr(p0), nblk, S, T, 3, c1, e0.tp1, e0.tout1, st_),
                  "conv1_wgrad")
            note("conv1 weight+bias gradient (rel L2)", rel_l2(p6, p0.sum(0)), 1e-5)
    torch.cuda.synchronize()
    return all(v <= b for v, b in worst.values())


def main():
    dev = torch.device("cuda:0")
    shapes = [(2, 3, 200, 128, 128, 64), (3, 5, 236, 128, 256, 128), (1, 1, 44, 128, 128, 128), (6, 8, 400, 512, 512, 512)]
    for a in sys.argv[1:]:
        if a.startswith("--shape"):
            shapes = [tuple(int(v) for v in a.split("=")[1].split(","))]
    ok = True
    for sh in shapes:
        ok = run(sh, dev) and ok
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
