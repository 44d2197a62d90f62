"""GPU parity at the NORTH-STAR shape (SynthesisModelCNN 128 ch x 400 samples: H = 18 432,
1 376 768 720 parameters) - the configuration bench.py times.

* golden G11: one train step of the REFERENCE at that shape, B = 2 (oracle/make_golden_r2.py):
  per-stage activations, LSTM state, output, loss, MCD, every parameter gradient and the parameters
  after one NAdam step, against the HIP path with its defaults (Winograd F(6,3) on pre-transformed
  operands for conv2 / conv3, conv4's input gradient on the same kernel, fused conv1 weight gradient,
  low-rank NAdam on W_hh); golden G11b: three steps at the same shape; the F(6,3) kernels against the
  direct MFMA kernels at the timed batch.  Observed deviations go to profiles/parity_observed.json.
* batch 256 (the timed batch; no CPU oracle fits it): the loss is a mean over independent windows, so
  the gradient of the whole batch is the mean of the gradients of its two halves - a size-independent
  property that crosses every batch-dependent code path (6.5 M-row split-K reductions, U = 8 label
  de-duplication, tile tails).
* train-mode dropout: the HIP-generated keep mask is read back, checked (rate, scale) and fed to the
  CPU oracle; forward and every gradient must agree, and the backward mask must be the forward mask.
"""
import os

import ctypes as C_
import numpy as np
import pytest
import torch

from tests import golden_inputs as gi
from tests.branch_planes import count_differing, hip_decisions
from tests.parity_record import record
from tests.test_gpu_parity import rel, rel_l2, _trainer

# one-step NAdam update-vector deviation from the reference golden G11 observed on the MI355X with the default kernels
# (profiles/parity_observed.json, round 5: worst 5.84e-4, ecog_conv_block.3.weight; the two bias vectors whose tiny gradients
# NAdam divides by their own magnitude - 5e-2 allowed in round 2 - sit below 5e-5 at this shape); the test allows twice that
UPD_OBSERVED, UPD_OBSERVED_BIAS = 6e-4, 6e-4

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def c3(dev):
    """The north-star model, seeded like golden G11 (about 40 s of host RNG for 1.38 G weights)."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 128, 400, dropout=0.0)
    assert model.get_nparams() == 1376768720
    chk = float(sum(np.abs(v.detach().numpy().reshape(-1)[::9973].astype(np.float64)).sum()
                    for v in model.state_dict().values()))
    # host copy of the seeded parameters (5.5 GB): tests that train restore them instead of re-drawing 1.38 G weights
    _INIT["state"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = _trainer(model, dev, 400)
    return model, tr, chk


_INIT = {}


def _pristine(model, dev):
    """The fixture's model with its seeded parameters restored and a fresh trainer (NAdam moments / step count reset)."""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(_INIT["state"][k])
    return _trainer(model, dev, 400)


def _sampled(g, prefix, name):
    """(reference sample, stride or None, (sum, abs-sum) or None) of a golden tensor."""
    full = prefix + name
    if full in g.files:
        return g[full], None, None
    key = next(k for k in g.files if k.startswith(full + "@s") and not k.endswith("@sum"))
    return g[key], int(key.rsplit("@s", 1)[1]), g[full + "@sum"]


def _take(t: torch.Tensor, stride):
    t = t.contiguous().reshape(-1)
    return (t if stride is None else t[::stride]).double().cpu().numpy()


def test_c3_shape_train_step_matches_reference_golden(dev, c3):
    model, tr, chk = c3
    g = np.load(os.path.join(GOLD, "g11_c3_step.npz"))
    D, Cn, T, B = (int(v) for v in g["dims"])
    xs, _t, _s, labs, tg = gi.train_batches(1, B, Cn, T, seed=int(g["data_seed"]))
    x, lab, tgt = xs[0], labs[0], tg[0]
    assert abs(gi.checksum(x, lab, tgt) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    assert abs(chk - float(g["param_sample_checksum"])) < 1e-6 * float(g["param_sample_checksum"])
    eng = model._engine
    assert eng.wino63 and eng.fuse_c1 and eng.H == 18432          # the defaults bench.py times (F(6,3) since round 4)
    names = [k for k, _ in model.named_parameters()]
    init = {}
    for k, p in model.named_parameters():
        _ref, stride, _ = _sampled(g, "final.", k)
        init[k] = _take(p.detach(), stride)
    model.train()
    # ---- forward: output and intermediates ----
    with torch.no_grad():
        out = model(x.to(dev), lab.to(dev))
    assert rel(out.cpu().numpy(), g["out"]) < 1e-4
    S = B * Cn
    acts = {}
    # the first stage hands its output to stage 2 as V (the F(4,3) input transform, tl_conv1_fwd_v) and the raw pooled
    # rows are not stored by default: a second forward with store_p1 yields them, and V must be their transform
    assert 1 not in eng.P
    V1 = eng._v_ready[1].clone()
    eng.store_p1 = True
    with torch.no_grad():
        out2 = model(x.to(dev), lab.to(dev))
    eng.store_p1 = False
    assert torch.equal(out2, out)
    from tests.wino63_ref import hex_transform, logical
    Vref = hex_transform(eng.P[1], S, eng.tp1)
    assert float((logical(V1)[:Vref.shape[0]].double() - Vref).abs().max()) <= 1e-6 * float(Vref.abs().max())
    assert float(V1[Vref.shape[0]:].abs().max()) == 0.0
    for i in (1, 2, 3, 4):
        st_tp = eng.tp1 if i == 1 else eng.stages[i - 2].tp_out
        tout = eng.tout1 if i == 1 else eng.stages[i - 2].tout
        ch = eng.P[i].shape[1]
        acts[f"ecog{i}"] = eng.P[i].view(B, Cn, st_tp, ch)[:, :, :tout, :].permute(0, 3, 2, 1)
    acts["ecog5"] = eng.P[5].view(B, Cn, eng.tp5, eng.ld5)[:, :, :eng.lat, :eng.Cc].permute(0, 3, 2, 1)
    acts["lstm_h"] = eng._h[-1][eng._uid.long()]
    acts["concat5"] = eng.Y[-1].view(B, Cn, eng.tp5, eng.ldy5)[:, :, :eng.lat, :eng.Cc].permute(0, 3, 2, 1)
    act_obs = {}
    for k, t in acts.items():
        ref, stride, sums = _sampled(g, "act.", k)
        got = _take(t, stride)
        assert got.shape == ref.reshape(-1).shape, k
        assert rel(got, ref.reshape(-1)) < 1e-4, k
        act_obs[k] = rel(got, ref.reshape(-1))
        if sums is not None:
            assert abs(float(t.double().abs().sum()) - sums[1]) < 1e-4 * sums[1], k
    # ---- one fused train step: loss, MCD, gradients, NAdam update ----
    tr._stats.zero_()
    tr._fused_step(x.to(dev), lab.to(dev), tgt.to(dev))
    st = tr._stats.cpu().numpy()
    assert abs(st[2] - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert abs(st[3] - float(g["mcd"])) < 1e-4 * float(g["mcd"])
    observed = {**{"act." + k: v for k, v in act_obs.items()}, "out": rel(out.cpu().numpy(), g["out"]), "loss": abs(st[2] - float(g["loss"])) / float(g["loss"]),
                "mcd": abs(st[3] - float(g["mcd"])) / float(g["mcd"])}
    grads = dict(tr._grads)
    assert eng.whh_factors is not None, "the W_hh gradient must stay in factored form at this shape"
    fa, fb = eng.whh_factors
    assert set(grads) | {eng.lowrank_param} == set(names)
    for k in names:
        ref, stride, sums = _sampled(g, "grad.", k)
        if k == eng.lowrank_param:                           # gradient = fa^T . fb, sampled without forming 5.4 GB
            n = 4 * eng.H * eng.H
            idx = torch.arange(0, n, stride, device=dev, dtype=torch.int64)
            r, c = idx // eng.H, idx % eng.H
            got = (fa.double()[:, r] * fb.double()[:, c]).sum(0).cpu().numpy()
            absum = 0.0
            for r0 in range(0, 4 * eng.H, 4096):
                absum += float((fa[:, r0:r0 + 4096].t() @ fb).double().abs().sum())
        else:
            got = _take(grads[k], stride)
            absum = float(grads[k].double().abs().sum())
        assert got.shape == ref.reshape(-1).shape, k
        observed["grad." + k] = rel_l2(got, ref.reshape(-1))
        assert observed["grad." + k] < grad_bound(k), (k, observed["grad." + k])
        if sums is not None:
            assert abs(absum - sums[1]) < max(grad_bound(k), 1e-4) * sums[1], k
    # parameters after the step (NAdam, low-rank path for W_hh): the update vector against the reference's
    upd = {}
    for k, p in model.named_parameters():
        ref, stride, _ = _sampled(g, "final.", k)
        fin = _take(p.detach(), stride)
        upd[k] = gi.update_rel_l2(fin, ref.reshape(-1), init[k])
    record("G11 one step at 128x400 (TONAL_WINO=6)", dict(observed, **{"upd." + k: v for k, v in upd.items()}))
    for k, v in upd.items():
        # round 5: 2 x what this test observes on the MI355X with the F(6,3) default (profiles/parity_observed.json: the
        # two bias vectors whose tiny gradients NAdam divides by their own magnitude sit at UPD_OBSERVED_BIAS, every other
        # tensor below UPD_OBSERVED) - round 2's bounds were 5e-2 / 2e-2
        tol = 2 * UPD_OBSERVED_BIAS if k in ("ecog_conv_block.9.bias", "concat_conv_block.4.bias") else 2 * UPD_OBSERVED
        assert v < tol, (k, v)


# gradient deviation from the reference golden G11 observed on the MI355X with the default kernels (profiles/parity_observed.json,
# "G11 one step"): the five conv-stack tensors sit at 1.2e-3 - 2.8e-3 because a handful of LeakyReLU' / arg-max branches on
# pre-activations that agree to 2e-6 fall the other way (test_c3_gradients_match_the_oracle_given_the_same_branches holds
# the arithmetic itself to 2e-5); everything else below 2e-6.  The unconditional test allows twice the observed value
GRAD_OBSERVED = {"ecog_conv_block.0": 2.8e-3, "ecog_conv_block.3": 2.6e-3, "ecog_conv_block.6": 2.4e-3,
                 "ecog_conv_block.9": 1.4e-3}
GRAD_OBSERVED_ELSEWHERE = 2.5e-6


def grad_bound(name: str) -> float:
    return 2 * GRAD_OBSERVED.get(name.rsplit(".", 1)[0], GRAD_OBSERVED_ELSEWHERE)


def test_c3_gradients_match_the_oracle_given_the_same_branches(dev, c3):
    """The 2e-3 deviation of the conv-stack gradients from golden G11 is explained, not tolerated: the HIP path's sign and
    arg-max bit planes are read back and handed to the CPU oracle's backward pass (oracle.synthesis_oracle._ActPoolDecided,
    as the dropout mask already is) on the reference's G11 parameters and inputs.  With the branches shared EVERY parameter
    gradient must agree to 2e-5 relative L2 - a 0.3 % error in any conv kernel cannot hide behind a near-tie any more.  The
    branches that differ from the ones the oracle takes by itself are counted and recorded."""
    from oracle import synthesis_oracle as so
    model, _tr, _chk = c3
    g = np.load(os.path.join(GOLD, "g11_c3_step.npz"))
    D, Cn, T, B = (int(v) for v in g["dims"])
    xs, _t, _s, labs, tg = gi.train_batches(1, B, Cn, T, seed=int(g["data_seed"]))
    x, lab, tgt = xs[0], labs[0], tg[0]
    tr = _pristine(model, dev)
    eng = model._engine
    assert eng.wino63 and eng.gy4 and eng.H == 18432                # the default kernels bench.py times
    model.train()
    tr._stats.zero_()
    tr._fused_step(x.to(dev), lab.to(dev), tgt.to(dev))
    torch.cuda.synchronize()
    grads = dict(tr._grads)
    fa, fb = eng.whh_factors
    dec = hip_decisions(eng, B, Cn)
    # ---- the oracle on the reference's parameters (host copy of the seeded state), its backward on the HIP path's branches ----
    leaves = {k: _INIT["state"][k].clone().requires_grad_(True) for k, _ in model.named_parameters()}
    own = {}
    ref = so.cnn_forward(leaves, x, lab, decisions=dec, own=own)
    ref_loss = so.l1_loss(ref, tgt.long())
    ref_grads = dict(zip(leaves, torch.autograd.grad(ref_loss, list(leaves.values()))))
    obs = {"loss": abs(float(tr._stats[2]) - float(ref_loss)) / float(ref_loss)}
    differ, total = count_differing(dec, own)
    for k, p in model.named_parameters():
        rg = ref_grads[k]
        if k == eng.lowrank_param:                                  # gradient = fa^T . fb: compared in row blocks on the GPU
            num = den = 0.0
            for r0 in range(0, 4 * eng.H, 4096):
                blk = (fa[:, r0:r0 + 4096].t() @ fb).double()
                rb = rg[r0:r0 + 4096].to(dev).double()
                num += float(((blk - rb) ** 2).sum())
                den += float((rb ** 2).sum())
            obs["grad." + k] = (num / den) ** 0.5
        else:
            obs["grad." + k] = rel_l2(grads[k].cpu().numpy(), rg.numpy())
    worst = max(v for k, v in obs.items() if k.startswith("grad."))
    print("gradients against the oracle on shared branches: worst", f"{worst:.2e}", "branches that differ from the oracle's own:",
          {k: v for k, v in differ.items() if v})
    record("G11 gradients, oracle backward on the HIP path's sign / arg-max planes",
           dict(obs, **{"differ." + k: v for k, v in differ.items()}, **{"of." + k: v for k, v in total.items()}))
    for k, v in obs.items():
        assert v < 2e-5, (k, v)
    # near-ties only: a handful of branches out of 10^8
    assert sum(differ.values()) <= 1e-5 * sum(total.values()), differ
    _pristine(model, dev)


def test_c3_shape_three_steps_follow_reference_golden(dev, c3):
    """Golden G11b (oracle/make_golden_r4.py): THREE NAdam steps of the reference at the timed geometry (128 x 400: the
    73 728-row LSTM, the low-rank W_hh update, the split-K Linear), B = 2, three distinct batches - loss, MCD and the mel
    MSE within 1e-3 at every step (north_star's form of parity), the outputs within 1e-3, and the three-step UPDATE
    vector of every parameter against the reference's (sampled).  G11 pins one step at this shape, G14 thirty steps at
    16 x 200; this one carries the timed geometry past step 1."""
    model, _tr, chk = c3
    g = np.load(os.path.join(GOLD, "g11b_c3_trajectory.npz"))
    D, Cn, T, B, steps = (int(v) for v in g["dims"])
    assert (D, Cn, T) == (80, 128, 400)
    xs, _t, _s, labs, tg = gi.train_batches(steps, B, Cn, T, seed=int(g["data_seed"]))
    assert abs(gi.checksum(*xs, *labs, *tg) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    tr = _pristine(model, dev)
    eng = model._engine
    assert eng.wino63 and eng.fuse_c1 and eng.H == 18432          # the defaults bench.py times (F(6,3) since round 4)
    strides = {}
    for k, p in model.named_parameters():                                      # the restored parameters are the reference's
        _ref, strides[k], _ = _sampled(g, "final.", k)
        assert np.array_equal(_take(p.detach(), strides[k]), g["init." + k].reshape(-1).astype(np.float64)), k
    model.train()
    worst = {}
    for s in range(steps):
        x, lab, tgt = xs[s].to(dev), labs[s].to(dev), tg[s].to(dev)
        with torch.no_grad():                                                  # (dropout 0: the step's own forward)
            out = model(x, lab)
        mse = float(((out.double().cpu() - tg[s].double()) ** 2).mean())
        worst[f"mse{s}"] = abs(mse - float(g["mses"][s])) / float(g["mses"][s])
        worst[f"out{s}"] = rel(out.cpu().numpy(), g["outs"][s])
        tr._stats.zero_()
        tr._fused_step(x, lab, tgt)
        st = tr._stats.cpu().numpy()
        worst[f"loss{s}"] = abs(st[2] - float(g["losses"][s])) / float(g["losses"][s])
        worst[f"mcd{s}"] = abs(st[3] - float(g["mcds"][s])) / float(g["mcds"][s])
        for q in ("mse", "out", "loss", "mcd"):
            assert worst[f"{q}{s}"] < 1e-3, (q, s, worst[f"{q}{s}"])
    # three-step update vectors (NAdam turns a numerically-zero gradient into a +-lr step: compare the vector, not elements)
    for k, p in model.named_parameters():
        ref, stride, _ = _sampled(g, "final.", k)
        err = gi.update_rel_l2(_take(p.detach(), stride), ref.reshape(-1), g["init." + k].reshape(-1).astype(np.float64))
        worst["upd." + k] = err
        # observed on MI355X (round 4): the largest three-step update deviation is 8.7e-4 (ecog_conv_block.3.bias), every
        # other tensor below 6e-4 - the bound is ~2.5 x that, not the 2e-2 / 5e-2 the one-step test had to allow in round 2
        assert err < 2.5e-3, (k, err)
    print("G11b observed deviations (largest):", {k: f"{v:.2e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:10]})
    record("G11b three steps at 128x400 (TONAL_WINO=6)", worst)
    _pristine(model, dev)


def test_c3_batch256_gradient_is_mean_of_half_batches(dev, c3):
    """The timed batch (256 windows, U = 8 distinct label rows) against its two halves."""
    model, tr, _ = c3
    eng = model._engine
    gen = torch.Generator(device=dev).manual_seed(99)
    B = 256
    x = torch.randn(B, 128, 400, device=dev, generator=gen)
    tones = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(1))
    syls = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(2))
    lab = gi.tone_dynamics(tones, syls).to(dev)
    tgt = 10 * torch.randn(B, 80, device=dev, generator=gen)
    prm = {k: p.detach() for k, p in model.named_parameters()}
    model.train()

    def grads_of(sl):
        xb, lb, tb = x[sl].contiguous(), lab[sl].contiguous(), tgt[sl].contiguous()
        out = eng.forward(prm, xb, lb, training=True, save=True, seed=0)
        n = xb.shape[0]
        dout = torch.zeros(n, eng.ldd, device=dev)
        stats = torch.zeros(4, device=dev)
        from decode_tonal_langauge_amd._lib import check, ptr
        check(eng.lib.tl_l1_mcd(ptr(out), ptr(tb), ptr(dout), ptr(stats), n, 80, eng.ldd, 1, 1.0,
                                torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
        g = {k: torch.empty_like(v) for k, v in prm.items() if k != eng.lowrank_param}
        eng.backward(prm, dout, g, whh_factors=True)
        fa, fb = eng.whh_factors
        # the W_hh gradient on a fixed 2048 x 2048 patch (forming 5.4 GB three times is not the point)
        g["whh_patch"] = fa[:, 4096:6144].t() @ fb[:, 1024:3072]
        return {k: v.clone() for k, v in g.items()}, out.clone()

    g_all, out_all = grads_of(slice(0, B))
    assert eng._U == 8
    g_a, out_a = grads_of(slice(0, B // 2))
    g_b, out_b = grads_of(slice(B // 2, B))
    assert float((torch.cat([out_a, out_b]) - out_all).abs().max() / out_all.abs().max()) < 1e-5
    for k in g_all:
        mean = 0.5 * (g_a[k].double() + g_b[k].double())
        err = float((g_all[k].double() - mean).norm() / max(float(mean.norm()), 1e-30))
        assert err < 2e-4, (k, err)


def test_c3_batch256_conv4_input_gradient_forms_agree(dev, c3):
    """Round 5, at the TIMED batch: conv4's input gradient on the NT63 kernel (six batched GEMMs, epilogue 7 writing conv3's
    Y3 / Vd3: 17 408 tiles, 68 per persistent workgroup) against the round-4 form (one-tap GEMM writing G3 +
    ``tl_wino63_unpool_yvd``) on the same forward pass: every gradient to 1e-5 (the bit words are shared, so the two forms
    can differ by the rounding of one K = 256 reduction only; observed on the MI355X: identical bit for bit - both forms
    accumulate the same k order on the same MFMA instruction and share the transform arithmetic)."""
    model, tr, _ = c3
    eng = model._engine
    assert eng.gy4
    gen = torch.Generator(device=dev).manual_seed(123)
    B = 256
    x = torch.randn(B, 128, 400, device=dev, generator=gen)
    tones = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(3))
    syls = torch.randint(0, 2, (B,), generator=torch.Generator().manual_seed(4))
    lab = gi.tone_dynamics(tones, syls).to(dev)
    tgt = 10 * torch.randn(B, 80, device=dev, generator=gen)
    prm = {k: p.detach() for k, p in model.named_parameters()}
    model.train()
    out = eng.forward(prm, x, lab, training=True, save=True, seed=0)
    dout = torch.zeros(B, eng.ldd, device=dev)
    stats = torch.zeros(4, device=dev)
    from decode_tonal_langauge_amd._lib import check, ptr
    check(eng.lib.tl_l1_mcd(ptr(out), ptr(tgt), ptr(dout), ptr(stats), B, 80, eng.ldd, 1, 1.0,
                            torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
    res = {}
    try:
        for form in ("nt63", "gemm"):
            eng.gy4 = form == "nt63"
            if not eng.gy4 and 3 not in eng.G:                      # (the default path never allocates conv3's gradient rows)
                st3 = eng.stages[1]
                eng.G[3] = torch.zeros(eng.S * st3.tp_out, st3.cout, device=dev)
            g = {k: torch.empty_like(v) for k, v in prm.items() if k != eng.lowrank_param}
            eng.backward(prm, dout, g, whh_factors=True)
            res[form] = {k: v.clone() for k, v in g.items()}
            # (the other form must write conv3's operands itself: what this one left in the real hexes is poisoned)
            nh3 = eng.S * (eng.stages[1].tp_in // 6)
            assert nh3 % 2 == 0 and bool(torch.isfinite(eng.Yt[3][:nh3]).all())
            eng.Yt[3][:nh3].fill_(float("nan"))
            eng.Vd[3][:nh3].fill_(float("nan"))
    finally:
        eng.gy4 = True
        eng.G.pop(3, None)
    worst = {}
    for k in res["gemm"]:
        a, b = res["nt63"][k].double(), res["gemm"][k].double()
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), k
        worst[k] = float((a - b).norm() / max(float(b.norm()), 1e-30))
        assert worst[k] < 1e-5, (k, worst[k])
    record("conv4 input gradient on the NT63 kernel vs one-tap GEMM + producer, whole model at batch 256 (rel L2 per gradient)", worst)


def test_f63_kernels_at_the_timed_batch_match_direct_kernels(dev, monkeypatch):
    """Round 5: the F(6,3) default against the direct MFMA kernels (TONAL_WINO=0) at the TIMED row count - batch 256,
    128 ch x 400 samples, 512 channels: 6.68 M conv rows per stage-2 pass, 8 704 row tiles, the 4 096-way split-K of the
    weight gradients - the one place the 6.5 M-row reductions meet the F(6,3) constants (32/45, 1/90 ...).  Same inputs
    for both engines, stage by stage through the C ABI: pooled rows 2e-5 (max norm), weight / bias gradients of conv2 and
    conv3 and the fused conv1 gradient 1e-4 relative L2, the input gradient of conv3 (which the default path only ever
    holds as Y2 = A dz: its planes 0 and 7 ARE un-pooled gradient rows) against the direct engine's G2 on the sequences
    at both ends of the batch."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    from decode_tonal_langauge_amd._lib import check, ptr
    from tests.wino63_ref import logical
    B, C, T, c1, c2, c3 = 256, 128, 400, 512, 512, 512
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    engs = {}
    for mode in ("0", "6"):
        monkeypatch.setenv("TONAL_WINO", mode)
        eng = CnnEngine(80, C, T, 1, 8, 0.0, 0.01, defs, [16, 8])
        if mode == "0":
            eng.fuse_c1 = False
        eng._alloc(B, dev)
        eng._alloc_bwd()
        engs[mode] = eng
    e0, e6 = engs["0"], engs["6"]
    assert e6.wino63 and e6.f63_yprod and e6.f63_yprod3 and not e0.wino63 and not e0.wino43
    S = e6.S
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.randn(B, C, T, device=dev, generator=g)
    names = {1: "ecog_conv_block.0", 2: "ecog_conv_block.3", 3: "ecog_conv_block.6"}
    prm = {names[1] + ".weight": torch.randn(c1, 1, 3, 1, device=dev, generator=g) * 0.5,
           names[1] + ".bias": torch.randn(c1, device=dev, generator=g) * 0.1}
    cin = c1
    for i, co in ((2, c2), (3, c3)):
        prm[names[i] + ".weight"] = torch.randn(co, cin, 3, 1, device=dev, generator=g) * (1.0 / (3 * cin) ** 0.5)
        prm[names[i] + ".bias"] = torch.randn(co, device=dev, generator=g) * 0.1
        cin = co
    st_ = torch.cuda.current_stream().cuda_stream
    w1 = prm[names[1] + ".weight"].reshape(c1, 3).contiguous()
    for eng in (e0, e6):
        eng._x = x.contiguous()
        eng.generation += 1
        eng._v_ready = {}
        if eng.wino63:
            V1 = eng._v_hex_buffer(eng.V, 1, S * eng.tp1, c1)
            check(eng.lib.tl_conv1_fwd_v6(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), None, ptr(V1), ptr(eng.bits[1]),
                                          ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "tl_conv1_fwd_v6")
            eng._v_ready[1] = V1
        else:
            check(eng.lib.tl_conv1_fwd(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), ptr(eng.P[1]), ptr(eng.bits[1]),
                                       ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "tl_conv1_fwd")
        for si in (2, 3):
            eng.stage_forward(eng.stages[si - 2], prm[names[si] + ".weight"], prm[names[si] + ".bias"])
    obs = {}
    rows = lambda t, tp, n: t.view(S, tp, -1)[:, :n]
    s0, s6 = e0.stages[1], e6.stages[1]
    a, b = rows(e6.P[3], s6.tp_out, s6.tout), rows(e0.P[3], s0.tp_out, s6.tout)
    obs["pooled_rows_conv3.max_norm"] = float((a - b).abs().max() / b.abs().max())
    assert obs["pooled_rows_conv3.max_norm"] < 2e-5
    for si, name in ((1, "conv1"), (2, "conv2"), (3, "conv3")):
        t6 = e6.tout1 if si == 1 else e6.stages[si - 2].tout
        tp6 = e6.tp1 if si == 1 else e6.stages[si - 2].tp_out
        tp0 = e0.tp1 if si == 1 else e0.stages[si - 2].tp_out
        for kind, b6, b0 in (("argmax", e6.bits, e0.bits), ("sign", e6.sbits, e0.sbits)):
            flips = rows(b6[si], tp6, t6) ^ rows(b0[si], tp0, t6)
            nz = flips[flips != 0]
            cnt = int(sum(bin(int(v) & 0xffffffff).count("1") for v in nz.cpu().tolist())) if nz.numel() < 100000 else -1
            obs[f"{name}.{kind}_bits_that_differ"] = cnt
            assert 0 <= cnt <= (0 if si == 1 else 4096), (name, kind, cnt)       # ties / near-ties only (of 1.7 G / 0.8 G / 0.4 G bits)
    # ---- backward from one random G3; the direct engine un-pools with the F(6,3) engine's bits (ties may differ) ----
    G3 = torch.randn(S, s6.tp_out, c3, device=dev, generator=g)
    G3[:, s6.tout:] = 0
    e6.G[3].copy_(G3.reshape(-1, c3))
    e0.G[3].copy_(G3.reshape(-1, c3))                              # (tp_out of stage 3 is the default geometry's in both)
    del G3
    for idx in (1, 2, 3):
        tpa = e0.tp1 if idx == 1 else e0.stages[idx - 2].tp_out
        tpb = e6.tp1 if idx == 1 else e6.stages[idx - 2].tp_out
        tb = e6.tout1 if idx == 1 else e6.stages[idx - 2].tout
        e0.bits[idx].view(S, tpa, -1)[:, :tb] = e6.bits[idx].view(S, tpb, -1)[:, :tb]
        e0.sbits[idx].view(S, tpa, -1)[:, :tb] = e6.sbits[idx].view(S, tpb, -1)[:, :tb]
    for si in (3, 2):
        res = {}
        for key, eng in (("0", e0), ("6", e6)):
            st = eng.stages[si - 2]
            w = prm[names[si] + ".weight"]
            gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
            eng.stage_wgrad(st, gw, gb)
            res[key] = (gw, gb, eng.stage_dgrad(st, w))
        obs[f"conv{si}.weight_grad.rel_l2"] = rel_l2(res["6"][0].cpu().numpy(), res["0"][0].cpu().numpy())
        obs[f"conv{si}.bias_grad.rel_l2"] = rel_l2(res["6"][1].cpu().numpy(), res["0"][1].cpu().numpy())
        assert obs[f"conv{si}.weight_grad.rel_l2"] < 1e-4 and obs[f"conv{si}.bias_grad.rel_l2"] < 1e-4, (si, obs)
        if si == 3:
            # conv3's input gradient: Y2 plane 0 / 7 of hex h (of stage 2) = un-pooled gradient row 6 h / 6 h + 5, i.e. pooled
            # row 3 h where its arg-max bit is clear / pooled row 3 h + 2 where it is set
            b2 = e6.stages[0]
            nh = b2.tp_in // 6
            Y2 = e6.Yt[2]
            worst = 0.0
            for q0 in (0, S // 2 - 64, S - 128):                  # 128 sequences at the front, the middle and the back
                sl = slice(q0, q0 + 128)
                y = logical(Y2[q0 * nh:(q0 + 128) * nh]).view(128, nh, 8, s6.cin)
                g2 = e0.G[2].view(S, s0.tp_in, -1)[sl]
                wbits = e6.bits[2].view(S, b2.tp_out, -1)[sl]
                sh = torch.arange(32, device=dev, dtype=torch.int32)
                odd = ((wbits[..., None] >> sh) & 1).reshape(128, b2.tp_out, -1).bool()
                nrow = min(3 * nh, g2.shape[1], b2.tout)
                hv = (nrow + 2) // 3                               # hexes with pooled row 3 h inside the valid time
                ref0 = torch.where(odd[:, 0:3 * hv:3], torch.zeros_like(g2[:, 0:3 * hv:3]), g2[:, 0:3 * hv:3])
                worst = max(worst, float((y[:, :hv, 0] - ref0).abs().max() / g2.abs().max()))
                hv7 = max(0, (min(nrow, g2.shape[1]) - 2 + 2) // 3)
                r7 = slice(2, 3 * hv7, 3)
                ref7 = torch.where(odd[:, r7], g2[:, r7], torch.zeros_like(g2[:, r7]))
                worst = max(worst, float((y[:, :ref7.shape[1], 7] - ref7).abs().max() / g2.abs().max()))
            obs["conv3.input_grad_rows_via_Y2.max_norm"] = worst
            assert worst < 2e-5, worst
        else:
            nblk = int(min(2048, S))
            p0 = torch.empty(nblk, 4 * c1, device=dev)
            check(e0.lib.tl_conv1_wgrad(ptr(e0._x), ptr(e0.G[1]), ptr(e0.bits[1]), ptr(p0), nblk, S, T, 3, c1, e0.tp1, e0.tout1, st_),
                  "tl_conv1_wgrad")
            obs["conv1.weight_and_bias_grad_through_conv2_input_grad.rel_l2"] = rel_l2(res["6"][2].double().sum(0).cpu().numpy(),
                                                                                        p0.double().sum(0).cpu().numpy())
            assert obs["conv1.weight_and_bias_grad_through_conv2_input_grad.rel_l2"] < 1e-4, obs
    print("F(6,3) vs direct kernels at batch 256:", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in obs.items()})
    record("F(6,3) vs direct MFMA kernels at batch 256, 128x400, 512 channels", obs)
    del engs, e0, e6, res
    torch.cuda.empty_cache()


def test_train_mode_dropout_mask_forward_and_gradients_against_oracle(dev):
    from decode_tonal_langauge_amd._lib import check, ptr
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from oracle import synthesis_oracle as so
    B, Cn, T = 6, 8, 200
    xs, _t, _s, labs, tg = gi.train_batches(1, B, Cn, T, seed=31)
    torch.manual_seed(2)
    model = SynthesisModelCNN(80, Cn, T, dropout=0.5)
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    model.to(dev).train()
    eng = model._engine
    out = model(xs[0].to(dev), labs[0].to(dev))
    loss = (out - tg[0].to(dev).long()).abs().mean()
    loss.backward()
    assert eng._p_drop_used == 0.5
    seed = eng._seed_used
    dec = hip_decisions(eng, B, Cn)

    def read_mask(seed_, row0=0, nb=B):
        # keep mask * 1/(1-p): the concat kernel applied to an all-ones stage-5 activation
        ones = torch.ones(nb * Cn * eng.tp5, eng.ld5, device=dev)
        xc = torch.empty(nb * Cn * eng.tp5, eng.ldx, device=dev)
        uid = torch.zeros(nb, dtype=torch.int32, device=dev)
        check(eng.lib.tl_concat_pack(ptr(ones), ptr(eng._h[-1]), ptr(uid), ptr(xc), nb, Cn, eng.tp5, eng.lat, eng.Cc,
                                     eng.Lc, eng.ld5, eng.H, eng.ldx, 0.5, seed_, row0 * Cn * eng.tp5,
                                     torch.cuda.current_stream().cuda_stream), "tl_concat_pack")
        return xc.view(nb, Cn, eng.tp5, eng.ldx)[:, :, :eng.lat, :eng.Cc].permute(0, 3, 2, 1).contiguous()

    mask = read_mask(seed)                                   # (B, Cc, lat, C) like the reference's dropout input
    vals = torch.unique(mask).cpu().tolist()
    assert vals == [0.0, 2.0], vals                          # dropped, or kept and scaled by 1 / (1 - p)
    n = mask.numel()
    rate = float((mask > 0).double().mean())
    assert abs(rate - 0.5) < 4 * 0.5 / np.sqrt(n), (rate, n)
    # the hash is indexed by the GLOBAL element: a shard starting at window 2 draws rows 2.. of the same mask
    assert torch.equal(read_mask(seed, row0=2, nb=B - 2), mask[2:])
    # ... and a new forward draws a new mask
    with torch.no_grad():
        model(xs[0].to(dev), labs[0].to(dev))
    assert eng._seed_used != seed and not torch.equal(read_mask(eng._seed_used), mask)
    # oracle with the HIP-generated mask
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = so.cnn_forward(leaves, xs[0], labs[0], dropout_mask=mask.cpu(), decisions=dec)
    ref_loss = so.l1_loss(ref, tg[0].long())
    ref_grads = dict(zip(leaves, torch.autograd.grad(ref_loss, list(leaves.values()))))
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    assert abs(float(loss.detach()) - float(ref_loss)) < 1e-5 * float(ref_loss)
    for k, p in model.named_parameters():              # the oracle's backward takes the HIP path's LeakyReLU' / arg-max branches
        assert rel_l2(p.grad.cpu().numpy(), ref_grads[k].numpy()) < 5e-5, k


def test_dropout_backward_uses_the_forward_mask(dev):
    """G5 (gradient at the stage-5 output) is exactly zero where the forward pass dropped, and carries
    the 1/(1-p) scale elsewhere."""
    from decode_tonal_langauge_amd._lib import check, ptr
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    B, Cn, T = 4, 8, 200
    xs, _t, _s, labs, tg = gi.train_batches(1, B, Cn, T, seed=32)
    torch.manual_seed(3)
    model = SynthesisModelCNN(80, Cn, T, dropout=0.5).to(dev).train()
    eng = model._engine
    out = model(xs[0].to(dev), labs[0].to(dev))
    xc_fwd = eng.Xc.clone()
    p5 = eng.P[5].clone()
    (out - tg[0].to(dev)).abs().mean().backward()
    v = lambda t, ld: t.view(B, Cn, eng.tp5, ld)[:, :, :eng.lat, :eng.Cc]
    kept_fwd = v(xc_fwd, eng.ldx) != 0
    live = v(p5, eng.ld5) != 0                                 # LeakyReLU outputs are non-zero almost surely
    g5 = v(eng.G[5], eng.ld5)
    dxc = v(eng.dXc, eng.ldx)
    assert float(kept_fwd[live].double().mean()) == pytest.approx(0.5, abs=0.02)
    assert bool((g5[live & ~kept_fwd] == 0).all())
    slope = torch.where(v(p5, eng.ld5) > 0, torch.ones_like(g5), torch.full_like(g5, eng.slope))
    sel = live & kept_fwd
    assert torch.allclose(g5[sel], (2.0 * dxc * slope)[sel], rtol=1e-6, atol=0)


def test_tone_dynamics_kernel_matches_reference_golden(dev):
    """tl_tone_dynamics against golden G5 (reference data_loading/utils.py:32-79), directly."""
    from decode_tonal_langauge_amd._lib import check, load, ptr
    g = np.load(os.path.join(GOLD, "g5_tone_dynamics.npz"))
    lib = load()

    def run(mapping, tones, syls):
        keys = sorted(int(k) for k in mapping)
        L = len(next(iter(mapping.values())))
        table = torch.zeros(max(keys) + 1, L)
        for k in keys:
            table[k] = torch.tensor(mapping[str(k)], dtype=torch.float32)
        t = torch.as_tensor(tones, dtype=torch.int64, device=dev)
        s = torch.as_tensor(syls, dtype=torch.int64, device=dev)
        lab = torch.empty(len(tones), 2, L, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.tl_tone_dynamics(ptr(t), ptr(s), ptr(table.to(dev)), ptr(lab), ptr(err), len(tones), table.shape[0], L,
                                   torch.cuda.current_stream().cuda_stream), "tl_tone_dynamics")
        return lab.cpu().numpy(), int(err.item())

    small, e = run({"0": [3, 3, 3], "1": [1, 2, 3]}, [1, 0], [0, 1])
    assert e == 0 and np.array_equal(small, g["out_small"].astype(np.float32))
    big, e = run(gi.TONE_MAP, g["tones"], g["syls"])
    assert e == 0 and np.array_equal(big, g["out"].astype(np.float32))
    _bad, e = run(gi.TONE_MAP, [0, 4, 1], [0, 0, 1])           # tone 4 is not in the table: flagged, not read
    assert e == 1


def test_lowrank_nadam_at_the_rank_limit(dev):
    """kr = 57 and 64 need 64.1 / 72 KB of dynamic LDS (above the default 64 KB per-block limit)."""
    from decode_tonal_langauge_amd.optim import FusedNAdam
    g = torch.Generator(device=dev).manual_seed(5)
    for kr in (57, 64):
        rows, cols = 96, 520
        w0 = torch.randn(rows, cols, device=dev, generator=g)
        pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
        oa, ob = FusedNAdam([pa], lr=5e-3, weight_decay=0.004), FusedNAdam([pb], lr=5e-3, weight_decay=0.004)
        for _ in range(2):
            fa = torch.randn(kr, rows, device=dev, generator=g)
            fb = torch.randn(kr, cols, device=dev, generator=g)
            oa.step(grads={pa: (fa.t() @ fb).contiguous()})
            ob.step(grads={}, lowrank={pb: (fa, fb)})
        assert float((pa - pb).detach().abs().max()) < 1e-5 * max(1.0, float(pa.detach().abs().max())), kr


def test_lowrank_nadam_pass_also_yields_the_last_bptt_product(dev):
    """tl_nadam_lowrank_dh: parameter and moments bit-identical to tl_nadam_lowrank's (same arithmetic, same order), and the
    slab sum equals fa[0:U] . p_OLD - the product with the weight as it was before the update - over ragged rows / columns,
    every U the kernel takes, ranks up to the LDS limit and several row-tile counts."""
    from decode_tonal_langauge_amd.optim import FusedNAdam
    g = torch.Generator(device=dev).manual_seed(9)
    for rows, cols, kr, U, row_tiles in ((1000, 520, 32, 8, 16), (96, 256, 5, 1, 1), (4099, 1028, 64, 8, 3), (33, 4, 8, 5, 2)):
        w0 = torch.randn(rows, cols, device=dev, generator=g)
        pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
        oa, ob = FusedNAdam([pa], lr=5e-3, weight_decay=0.004), FusedNAdam([pb], lr=5e-3, weight_decay=0.004)
        for _ in range(2):
            fa = torch.randn(kr, rows, device=dev, generator=g)
            fb = torch.randn(kr, cols, device=dev, generator=g)
            old = pb.detach().clone()
            nslab = -(-(-(-rows // 32)) // row_tiles)
            slab = torch.full((nslab, U, cols), float("nan"), device=dev)
            oa.step_lowrank({pa: (fa, fb)})
            ob.step_lowrank({pb: (fa, fb)}, dh=(slab, U, row_tiles))
            assert torch.equal(pa.detach(), pb.detach())
            assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"])
            assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
            ref = fa[:U].double() @ old.double()
            got = slab.double().sum(0)
            assert float((got - ref).abs().max()) < 1e-5 * float(ref.abs().max()), (rows, cols, kr, U, row_tiles)
    with pytest.raises(RuntimeError):           # U above the factor rank / a slab of the wrong shape
        ob.step_lowrank({pb: (fa[:2], fb[:2])}, dh=(torch.empty(1, 5, cols, device=dev), 5, 2))


def test_recurrent_gradient_stream_matches_float64(dev):
    """tl_lstm_gw (dgates . W as a stream of W, round 6): ragged row counts (a last block shorter than the others, fewer rows than
    one block), K that is not a multiple of the 1024-column workgroup tile, every U, a row-offset view (the data-parallel shard form),
    against a float64 product."""
    from decode_tonal_langauge_amd._lib import check, load, ptr
    lib = load()
    st = torch.cuda.current_stream().cuda_stream
    g_ = torch.Generator(device=dev).manual_seed(17)
    for N, K, U, rpb in ((4099, 1028, 8, 512), (300, 2052, 1, 512), (2048, 1024, 5, 8), (1536, 516, 8, 1024)):
        W = torch.randn(N, K, device=dev, generator=g_)
        g = torch.randn(U, N, device=dev, generator=g_)
        nb = -(-N // rpb)
        slab = torch.full((nb, U, K), float("nan"), device=dev)
        check(lib.tl_lstm_gw(ptr(g), ptr(W), ptr(slab), U, N, K, N, K, rpb, st), "tl_lstm_gw")
        ref = g.double() @ W.double()
        got = slab.double().sum(0)
        assert float((got - ref).abs().max()) < 2e-6 * float(ref.abs().max()), (N, K, U, rpb)
    # rows [r0, r0 + R) of a wider gradient / a taller weight
    N, K, U, r0, R = 4096, 1024, 8, 1024, 2048
    W = torch.randn(N, K, device=dev, generator=g_)
    g = torch.randn(U, N, device=dev, generator=g_)
    slab = torch.empty(-(-R // 512), U, K, device=dev)
    check(lib.tl_lstm_gw(g.data_ptr() + 4 * r0, W.data_ptr() + 4 * r0 * K, ptr(slab), U, R, K, N, K, 512, st), "tl_lstm_gw")
    ref = g[:, r0:r0 + R].double() @ W[r0:r0 + R].double()
    assert float((slab.double().sum(0) - ref).abs().max()) < 2e-6 * float(ref.abs().max())
