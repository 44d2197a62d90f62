"""GPU: randomised-shape sweep of the partial-tile stores of the Winograd NT kernels (round 5).

Every epilogue of the V-form kernels drops the stores of rows / columns outside the matrix through the hardware range check
of a buffer resource (per-lane offsets that carry an out-of-range marker: csrc/tonal_wino63_epi.h, ``v6_store_at``).  Round 4
found one way for that to fail silently (marker + instruction immediate) and covered it with five hand-picked shapes; this
sweep draws 200 shapes - sequences from 1 to 42, sequence lengths 44..400 (tiles that end inside a half-wave, a wave, a
sequence; time padding), channel counts with and without a column tail - and holds the F(6,3) (default) and F(4,3) V-form
kernels to the direct MFMA kernels on identical inputs, stage by stage through the C ABI, with every launch repeated into
NaN-filled buffers (bit-identical results required)."""
import os

import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from tests.test_gpu_parity import f63_stage_check, one_tap_gy_check, rel, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


class _Env:
    """os.environ edits undone at the end of an example (hypothesis re-runs the test body: no function-scoped fixture)."""

    def __init__(self):
        self.saved = {}

    def set(self, k, v):
        self.saved.setdefault(k, os.environ.get(k))
        os.environ[k] = v

    def undo(self):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def f43_stage_check(dev, shape):
    """The V-form F(4,3) kernels (tonal_wino43v.hip: forward POOL epilogue, MASK input gradient, weight gradient + Vd) against
    the direct kernels on random stage inputs / gradients / bit words; each Winograd launch twice."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    B, C, T, c1, c2, c3 = shape
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    res = {}
    for mode in ("0", "4"):
        eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
        assert not eng.wino63
        eng.wino43, eng.fuse_c1 = mode == "4", False
        eng.wino_vout = False          # stage kernels one at a time on random inputs: every stage reads P
        eng._alloc(B, dev)
        eng._alloc_bwd()
        g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
        for k in sorted(eng.P):
            eng.P[k].normal_(generator=g)
        for k in sorted(eng.G):
            eng.G[k].normal_(generator=g)
        for k in sorted(eng.bits):
            eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
            eng.sbits[k].random_(-2**31, 2**31 - 1, generator=g)
        out = []
        for si in (2, 3):
            stg = eng.stages[si - 2]
            if mode == "4":
                assert eng._v43(stg), (shape, si)
            # rows past the valid time of a sequence are zero in every real tensor (the epilogues write them so)
            eng.P[si - 1].view(eng.S, stg.tp_in, -1)[:, stg.tin:, :] = 0
            eng.G[si].view(eng.S, stg.tp_out, -1)[:, stg.tout:, :] = 0
            w = torch.randn(stg.cout, stg.cin, 3, 1, device=dev, generator=g) * 0.05
            b = torch.randn(stg.cout, device=dev, generator=g) * 0.1
            keep = (eng.P[si].clone(), eng.bits[si].clone(), eng.sbits[si].clone())
            runs = []
            for rep in range(2 if mode == "4" else 1):
                gw, gb = torch.zeros_like(w), torch.zeros_like(b)
                eng.G[si - 1].fill_(float("nan"))
                eng._v_ready = {}
                eng.stage_wgrad(stg, gw, gb)             # (V-form: writes Vd, the operand of the input gradient below)
                eng.stage_dgrad(stg, w)
                dg = eng.G[si - 1].clone()
                eng.P[si].fill_(float("nan"))
                eng.stage_forward(stg, w, b)
                runs.append((eng.P[si].clone(), eng.bits[si].clone(), dg, gw, gb))
                eng.P[si].copy_(keep[0]); eng.bits[si].copy_(keep[1]); eng.sbits[si].copy_(keep[2])
            if len(runs) == 2:
                for a_, b_ in zip(*runs):
                    assert torch.equal(a_, b_), (shape, si)
            out.append((runs[0], stg))
        res[mode] = out
    for ((p0, b0, d0, w0, g0), s0), ((p1, b1, d1, w1, g1), s1) in zip(res["0"], res["4"]):
        S = B * C
        rows = lambda t, tp, n: t.view(S, tp, -1)[:, :n]
        assert bool(torch.isfinite(p1).all()) and bool(torch.isfinite(d1).all()), shape      # every row of the matrix is written
        assert rel(rows(p1, s1.tp_out, s1.tout).cpu().numpy(), rows(p0, s0.tp_out, s1.tout).cpu().numpy()) < 2e-5, shape
        assert int(((rows(b0, s0.tp_out, s1.tout) ^ rows(b1, s1.tp_out, s1.tout)) != 0).sum()) <= 4, shape
        assert rel_l2(rows(d1, s1.tp_in, s1.tin).cpu().numpy(), rows(d0, s0.tp_in, s1.tin).cpu().numpy()) < 1e-5, shape
        assert rel_l2(w1.cpu().numpy(), w0.cpu().numpy()) < 1e-5 and rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-5, shape


N_CASES = int(os.environ.get("TONAL_SWEEP_CASES", "200"))


@settings(max_examples=N_CASES, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(st.data())
def test_partial_tile_stores_of_the_nt_kernels_on_random_shapes(dev, data):
    fam = data.draw(st.sampled_from(["6", "6", "4"]), label="family")
    B = data.draw(st.integers(1, 6), label="B")
    C = data.draw(st.integers(1, 7), label="C")
    T = data.draw(st.integers(44, 400), label="T")
    if fam == "6":
        c1 = data.draw(st.sampled_from([128, 256]), label="c1")
        c2 = data.draw(st.sampled_from([128, 256]), label="c2")
        c3 = data.draw(st.sampled_from([64, 128, 192]), label="c3")
        yprod = data.draw(st.sampled_from(["1", "1", "0"]), label="yprod")
        env = _Env()
        try:
            f63_stage_check(dev, (B, C, T, c1, c2, c3), yprod, env.set, twice=True, ntail=c3 >= 128)
        finally:
            env.undo()
    else:
        c1 = data.draw(st.sampled_from([64, 128]), label="c1")
        c2 = data.draw(st.sampled_from([64, 128, 192]), label="c2")
        c3 = data.draw(st.sampled_from([32, 64, 96, 160]), label="c3")
        env = _Env()
        try:
            env.set("TONAL_WINO", "4")
            f43_stage_check(dev, (B, C, T, c1, c2, c3))
        finally:
            env.undo()


@settings(max_examples=int(os.environ.get("TONAL_SWEEP_CASES_GY", "200")), deadline=None, derandomize=True, database=None,
          suppress_health_check=list(HealthCheck))
@given(st.data())
def test_one_tap_input_gradient_on_the_nt_kernel_on_random_shapes(dev, data):
    """Round 5: ``tl_conv1_wino63v_dgrad_nt`` (conv4's input gradient as six batched GEMMs of the NT63 kernel, epilogue 7 writing Y / Vd
    of the 3-tap stage below) on drawn geometries - sequence counts whose last hex is half empty, hexes that straddle sequences
    (an odd number of hexes per sequence), valid lengths short of the padding, bit arrays narrower than the hex geometry, column
    tails - against the float64 restatement of ``one_tap_gy_check``; every launch twice, bit-identical."""
    S = data.draw(st.integers(1, 300), label="sequences")
    hexes = data.draw(st.integers(2, 17), label="hexes per sequence of the stage below")
    tp3 = 6 * hexes
    tpg = tp3 // 2
    gtp3 = data.draw(st.integers(max(2, tpg - 2), tpg), label="rows per sequence of the bit arrays")
    tout3 = data.draw(st.integers(2, gtp3), label="valid pooled rows")
    c3 = data.draw(st.sampled_from([64, 96, 128, 256]), label="N")
    c4 = data.draw(st.sampled_from([64, 96, 128]), label="K")
    one_tap_gy_check(dev, (S, tp3, tout3, gtp3, c3, c4))
