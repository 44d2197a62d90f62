"""GPU parity tests (run with ``-m gpu`` on the MI355X box): the HIP path, called through the
C ABI, against the committed golden vectors of the reference and against the CPU oracle on the
same seeded inputs.  Tolerances: fp32 paths 1e-3 relative (north_star), observed ~1e-6;
fp64 signal paths 1e-9 relative.
"""
import os
from argparse import Namespace

import ctypes as C_
import numpy as np
import pytest
import torch

from tests import golden_inputs as gi
from tests.branch_planes import hip_decisions

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def rel_l2(a, b):
    """Relative L2 distance.  Used for gradients: LeakyReLU makes single gradient entries
    ill-conditioned - a pre-activation that is 1e-9 in one summation order and -1e-9 in another
    changes that entry's slope from 1 to 0.01 (observed: 2 of 1.5 M stage-3 activations of the G4
    case flip between this kernel and the CPU reference) - so an element-wise max-norm bound is
    not meaningful for the layers upstream of such an entry, while the tensor as a whole is."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(float(np.linalg.norm(b)), 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def _trainer(model, dev, T, n_ch=8):
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    tone = LogisticRegressionClassifier(n_ch * T, 4)
    syl = LogisticRegressionClassifier(n_ch * T, 2)
    return SynthesisTrainer(model, tone, syl, gi.TONE_MAP, device=dev, verbose=False)


def test_cnn_forward_matches_reference_golden(dev):
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    g = np.load(os.path.join(GOLD, "g1_cnn_forward.npz"))
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 4, 100).eval()
    x, lab = gi.g1_inputs()
    assert abs(gi.checksum(x, lab) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    assert abs(gi.checksum(*model.state_dict().values()) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    model.to(dev)
    with torch.no_grad():
        out = model(x.to(dev), lab.to(dev))
    assert out.shape == (3, 80)
    assert rel(out.cpu().numpy(), g["out"]) < 1e-4
    # intermediate checks through the engine buffers (layout: [seq*Tp + t][channel])
    eng = model._engine
    e5 = eng.P[5].view(3, 4, eng.tp5, eng.ld5)[:, :, :eng.lat, :64].permute(0, 3, 2, 1).cpu().numpy()
    assert rel(e5, g["ecog5"]) < 1e-4
    h = eng._h[-1][eng._uid.long()].cpu().numpy()
    assert rel(h, g["lstm_h"]) < 1e-4


def test_cnn_train_steps_match_reference_golden(dev):
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    g = np.load(os.path.join(GOLD, "g4_cnn_train.npz"))
    xs, _t, _s, labs, tg = gi.train_batches(3, 8, 16, 200)
    assert abs(gi.checksum(*xs, *labs, *tg) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 16, 200, dropout=0.0)
    assert model.get_nparams() == 7169232
    init = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
    tr = _trainer(model, dev, 200)
    model.train()
    losses, mcds = [], []
    for s in range(3):
        tr._fused_step(xs[s].to(dev), labs[s].to(dev), tg[s].to(dev))
        st = tr._stats.cpu().numpy()
        losses.append(st[2])
        mcds.append(st[3])
        if s == 0:
            grads_now = dict(tr._grads)
            if model._engine.whh_factors is not None:
                fa, fb = model._engine.whh_factors              # the optimiser consumed the factors, not a tensor
                grads_now[model._engine.lowrank_param] = fa.t() @ fb
            assert set(grads_now) == {k for k, _ in model.named_parameters()}
            for k, gr in grads_now.items():
                gr = gr.cpu().numpy()
                if "grad1." + k in g:
                    assert rel_l2(gr, g["grad1." + k]) < 5e-3, k
                else:
                    assert rel_l2(gr.reshape(-1)[::97], g["grad1." + k + "@s97"]) < 5e-3, k
                    sums = g["grad1." + k + "@sum"]
                    assert abs(np.abs(gr).astype(np.float64).sum() - sums[1]) < 5e-3 * sums[1], k
                if not k.startswith("ecog_conv_block"):      # nothing upstream of a flipped LeakyReLU
                    ref_g = g["grad1." + k] if "grad1." + k in g else None
                    if ref_g is not None:
                        assert rel(gr, ref_g) < 1e-4, k
    assert rel(losses, g["losses"]) < 1e-4
    assert rel(mcds, g["mcds"]) < 1e-4
    for k, p in model.named_parameters():
        fin = p.detach().cpu().numpy()
        # NAdam divides by sqrt(v): after 3 steps the update of a bias whose gradient is tiny carries the
        # rounding noise of any independent fp32 summation order at the 2-3e-3 level (direct MFMA form
        # 1.8e-3 / 3.2e-3, F(2,3) 1.8e-3 / 3.2e-3, F(4,3) 2.5e-3 / 1.0e-3 on concat_conv_block.4.bias /
        # ecog_conv_block.9.bias; scripts/update_parity.py).  Only those two bias vectors get the wide
        # bound; every other tensor of the block is held to 5e-3
        tol = 5e-2 if k in ("ecog_conv_block.9.bias", "concat_conv_block.4.bias") else 5e-3
        if "final." + k in g:
            assert gi.update_rel_l2(fin, g["final." + k], init[k]) < tol, k
        else:
            assert gi.update_rel_l2(fin.reshape(-1)[::97], g["final." + k + "@s97"], init[k].reshape(-1)[::97]) < tol, k


def test_cnn_autograd_path_equals_fused_path(dev):
    """model(x, lab) + loss.backward() (torch.autograd.Function) gives the same gradients as the
    trainer's fused step, and they match the oracle on the same inputs."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from oracle import synthesis_oracle as so
    xs, _t, _s, labs, tg = gi.train_batches(1, 5, 4, 100, seed=77)
    torch.manual_seed(3)
    model = SynthesisModelCNN(80, 4, 100, dropout=0.0)
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    model.to(dev).train()
    out = model(xs[0].to(dev), labs[0].to(dev))
    loss = (out - tg[0].to(dev).long()).abs().mean()
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    # (the oracle's backward takes the LeakyReLU' / arg-max branches the HIP path took: tests/branch_planes.py)
    ref = so.cnn_forward(leaves, xs[0], labs[0], decisions=hip_decisions(model._engine, 5, 4))
    ref_loss = so.l1_loss(ref, tg[0].long())
    ref_grads = dict(zip(leaves, torch.autograd.grad(ref_loss, list(leaves.values()))))
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    for k, p in model.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), ref_grads[k].numpy()) < 5e-5, k


def test_cnn_random_labels_no_dedup(dev):
    """Arbitrary float label sequences (every row distinct): the LSTM runs with U = B."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from oracle import synthesis_oracle as so
    torch.manual_seed(5)
    model = SynthesisModelCNN(40, 8, 60, lstm_channels=2, conv_channels=16, dropout=0.0)
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    x = torch.randn(70, 8, 60)
    lab = torch.randn(70, 2, 3)
    model.to(dev).train()
    out = model(x.to(dev), lab.to(dev))
    out.square().mean().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = so.cnn_forward(leaves, x, lab, decisions=hip_decisions(model._engine, 70, 8))
    ref_grads = dict(zip(leaves, torch.autograd.grad(ref.square().mean(), list(leaves.values()))))
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    for k, p in model.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), ref_grads[k].numpy()) < 5e-5, k


def test_signal_filters_match_reference_golden(dev):
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    g = np.load(os.path.join(GOLD, "g6_signal.npz"))
    x, x2 = gi.g6_inputs()
    assert abs(gi.checksum(x, x2) - float(g["in_checksum"])) < 1e-9 * float(g["in_checksum"])
    hil = ff.hilbert_filter(x, 400, freq_ranges=[70., 150.])
    assert hil.dtype == np.float64 and rel(hil, g["hilbert"]) < 1e-9
    assert rel(ff.hilbert_filter(x, 400, freq_ranges=[70., 150.], envelope=False), g["hilbert_real"]) < 1e-9
    # 8 bands with truncated kernels: short recordings take the Hermitian-symmetry kernel (tl_gauss_envelope_sym), recordings
    # of >= 1024 samples overlap-save on the LDS FFT (tl_hilbert_ols); the plain bank (tl_gauss_envelope) must give the same
    # numbers far below the golden tolerance, and so must the three on a longer recording
    os.environ["TONAL_HILBERT"] = "taps"
    try:
        plain = ff.hilbert_filter(x, 400, freq_ranges=[70., 150.])
    finally:
        os.environ.pop("TONAL_HILBERT", None)
    assert rel(plain, g["hilbert"]) < 1e-9 and rel(hil, plain) < 1e-13
    xl = np.random.default_rng(5).standard_normal((3, 5003))
    res = {}
    # (overlap-save comes in two forms: tl_hilbert_ols_bl, hilbert=ols, the default - it drops the bins of a band's kernel
    # spectrum that lie below 1e-12 of its peak, outside a 256-bin window - and tl_hilbert_ols with all 1024 bins,
    # hilbert=ols_full; sym / taps: the time-domain kernels with / without Hermitian symmetry)
    for mode in ("ols", "ols_full", "sym", "taps"):
        os.environ["TONAL_HILBERT"] = mode
        try:
            res[mode] = (ff.hilbert_filter(xl, 400, freq_ranges=[70., 150.]),
                         ff.hilbert_filter(xl, 400, freq_ranges=[70., 150.], envelope=False))
        finally:
            os.environ.pop("TONAL_HILBERT", None)
    for k in ("ols_full", "sym"):
        assert rel(res[k][0], res["taps"][0]) < 1e-13 and rel(res[k][1], res["taps"][1]) < 1e-12
    assert rel(res["ols"][0], res["taps"][0]) < 1e-12 and rel(res["ols"][1], res["taps"][1]) < 1e-12
    assert not np.array_equal(res["ols"][0], res["ols_full"][0])         # (the two forms really are different kernels)
    for mode in ("ols", "ols_full"):                                     # a silent channel stays exactly zero (magnitude guard)
        os.environ["TONAL_HILBERT"] = mode
        try:
            assert not ff.hilbert_filter(np.zeros((2, 2048)), 400, freq_ranges=[70., 150.]).any()
        finally:
            os.environ.pop("TONAL_HILBERT", None)
    assert rel(ff.butter_filter(x, [0.3, 100], 400), g["butter"]) < 1e-9
    assert rel(ff.butter_filter(x, [0.3, 100], 400, causal=True), g["butter_causal"]) < 1e-9
    assert rel(ff.fir_bandpass_filter(x, 400, 390, [100.]), g["fir"]) < 1e-9
    assert rel(ff.fir_bandpass_filter(x, 400, 64, [60., 120.]), g["fir2"]) < 1e-9
    # recordings of >= 1024 samples take the overlap-save form (tl_fir_bank_ols): same numbers as the time-domain kernel
    for order, cfs_ in ((390, [100.]), (64, [60., 120.]), (512, [90.])):
        a_ = ff.fir_bandpass_filter(xl, 400, order, cfs_)
        saved = ff._OLS_N
        ff._OLS_N = 1 << 60                                              # (no recording is that long: the time-domain kernel)
        try:
            b_ = ff.fir_bandpass_filter(xl, 400, order, cfs_)
        finally:
            ff._OLS_N = saved
        assert rel(a_, b_) < 1e-12
    x32 = xl.astype(np.float32)
    a_ = ff.fir_bandpass_filter(x32, 400, 390, [100.])
    assert a_.dtype == np.float32 and rel(a_, ff.fir_bandpass_filter(xl, 400, 390, [100.])) < 1e-5
    # float32 input, odd length, two ranges: the reference itself computes this in complex64
    h2 = ff.hilbert_filter(x2, 400, freq_ranges=[(70., 110.), (110., 150.)])
    assert h2.dtype == np.float64 and rel(h2, g["hilbert2"]) < 1e-5
    prm = Namespace(signal_freq=400, bands=[
        {"method": "hilbert", "params": {"freq_ranges": [70., 150.], "envelope": True}},
        {"method": "butter", "params": {"freqs": [0.3, 100], "filter_type": "bandpass"}},
        {"method": "fir", "params": {"order": 390, "center_frequencies": [100.]}}])
    out = ff.run(x, prm)
    assert out.shape == (6, 1000) and rel(out, g["run"]) < 1e-9


def test_lite_forward_matches_reference_golden(dev):
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    g = np.load(os.path.join(GOLD, "g2_lite_forward.npz"))
    torch.manual_seed(0)
    model = SynthesisLite(80, 32, 200).eval()
    x, lab = gi.g2_inputs()
    assert abs(gi.checksum(x, lab) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    model.to(dev)
    with torch.no_grad():
        out = model(x.to(dev), lab.to(dev))
    assert rel(out.cpu().numpy(), g["out"]) < 1e-4


def test_lite_train_steps_match_reference_golden(dev):
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    g = np.load(os.path.join(GOLD, "g3_lite_train.npz"))
    xs, _t, _s, labs, tg = gi.train_batches(3, 64, 32, 200)
    torch.manual_seed(0)
    model = SynthesisLite(80, 32, 200, dropout=0.0)
    init = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
    tr = _trainer(model, dev, 200)
    model.train()
    losses, mcds = [], []
    for s in range(3):
        tr._fused_step(xs[s].to(dev), labs[s].to(dev), tg[s].to(dev))
        st = tr._stats.cpu().numpy()
        losses.append(st[2])
        mcds.append(st[3])
        if s == 0:
            for k, gr in tr._grads.items():
                gr = gr.cpu().numpy()
                ref = g["grad1." + k] if "grad1." + k in g else None
                if ref is not None and np.abs(ref).max() < 1e-6:
                    assert np.abs(gr).max() < 1e-5, k          # analytically zero (conv bias before BN)
                elif ref is not None:
                    assert rel_l2(gr, ref) < 5e-3, k
                else:
                    assert rel_l2(gr.reshape(-1)[::97], g["grad1." + k + "@s97"]) < 5e-3, k
    assert rel(losses, g["losses"]) < 1e-4 and rel(mcds, g["mcds"]) < 1e-4
    sd = model.state_dict()
    assert rel(sd["ecog_conv.1.running_mean"].cpu().numpy(), g["run_mean0"]) < 1e-4
    assert rel(sd["ecog_conv.1.running_var"].cpu().numpy(), g["run_var0"]) < 1e-4
    assert rel(sd["ecog_conv.5.running_mean"].cpu().numpy(), g["run_mean1"]) < 1e-4
    assert rel(sd["ecog_conv.5.running_var"].cpu().numpy(), g["run_var1"]) < 1e-4
    assert int(sd["ecog_conv.1.num_batches_tracked"]) == 3
    for k, p in model.named_parameters():
        if np.abs(g["grad1." + k]).max() < 1e-6 if "grad1." + k in g else False:
            continue
        fin = p.detach().cpu().numpy()
        if "final." + k in g:
            assert gi.update_rel_l2(fin, g["final." + k], init[k]) < 2e-2, k
        else:
            assert gi.update_rel_l2(fin.reshape(-1)[::97], g["final." + k + "@s97"], init[k].reshape(-1)[::97]) < 2e-2, k


def test_trainer_history_matches_reference_golden(dev):
    """The reference's SynthesisTrainer.train/evaluate on SynthesisLite + LogisticRegression
    classifiers (golden G9) against this build's trainer: same split, same batch order."""
    from decode_tonal_langauge_amd.data_loading.dataloaders import split_dataset
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    g9 = np.load(os.path.join(GOLD, "g9_trainer.npz"))
    N, C, T = 96, 32, 200
    e_non, e_syl, e_tone, tgt = gi.g9_dataset(N, C, T)
    ds = torch.utils.data.TensorDataset(e_non, e_syl, e_tone, tgt)
    torch.manual_seed(7)
    tone_model = LogisticRegressionClassifier(8 * T, 4)
    syl_model = LogisticRegressionClassifier(8 * T, 2)
    loaders = split_dataset(ds, [0.75, 0.25], [True, False], batch_size=16, seed=11)
    torch.manual_seed(0)
    model = SynthesisLite(80, C, T, dropout=0.0)
    trainer = SynthesisTrainer(model, tone_model, syl_model, gi.TONE_MAP, device=dev, verbose=False)
    hist = trainer.train(loaders[0], 2, verbose=False)
    mcd, recon, origin = trainer.evaluate(loaders[1])
    assert rel(np.array(hist), g9["history"]) < 1e-3
    assert abs(mcd - float(g9["eval_mcd"])) < 1e-3 * float(g9["eval_mcd"])
    assert recon.shape == g9["recon"].shape and rel(origin, g9["origin"]) < 1e-6
    assert rel(recon, g9["recon"]) < 2e-2


def test_signal_full_size_properties(dev):
    """BASELINE C5 shape (256 ch x 24 000 samples): size-independent properties of the filters."""
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(256, 24000, device=dev, generator=g, dtype=torch.float32)
    y = torch.randn(256, 24000, device=dev, generator=g, dtype=torch.float32)
    lin = lambda f: float(((f(2.0 * x - 3.0 * y) - (2.0 * f(x) - 3.0 * f(y))).abs().max() / f(x).abs().max()))
    # analytic band signal (real part), zero-phase IIR and FIR are linear maps
    assert lin(lambda v: ff.hilbert_filter(v, 400, [70., 150.], envelope=False)) < 1e-6
    assert lin(lambda v: ff.butter_filter(v, [0.3, 100], 400)) < 1e-5
    assert lin(lambda v: ff.fir_bandpass_filter(v, 400, 390, [100.]).double()) < 1e-5
    # envelope is positively homogeneous and non-negative; per-channel independence
    env = ff.hilbert_filter(x, 400, [70., 150.])
    assert env.dtype == torch.float64 and float(env.min()) >= 0.0
    assert float((ff.hilbert_filter(-2.5 * x, 400, [70., 150.]) - 2.5 * env).abs().max() / env.max()) < 1e-6
    sub = ff.hilbert_filter(x[100:104].contiguous(), 400, [70., 150.])
    assert torch.equal(sub, env[100:104])
    # FIR impulse response = taps (causal, zero initial state)
    imp = torch.zeros(1, 1000, device=dev, dtype=torch.float64)
    imp[0, 0] = 1.0
    from scipy.signal import firwin
    taps = firwin(65, [60 * 0.9 / 200, 60 * 1.1 / 200], pass_zero=False, fs=400)
    h = ff.fir_bandpass_filter(imp, 400, 64, [60.])[0, :65].cpu().numpy()
    assert np.max(np.abs(h - taps)) < 1e-15


def test_hilbert_overlap_save_forms_match_oracle_at_ragged_sizes(dev, monkeypatch):
    """Both overlap-save kernels (tl_hilbert_ols_bl, tl_hilbert_ols) against the CPU oracle (pinned by G6) at the sizes where
    segments wrap, end ragged or are exactly one transform long; fp64 input 1e-9, fp32 input 1e-5 (the reference itself
    computes that case in complex64: frequency_filter.py:167)."""
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    from oracle import signal_oracle as sg
    rng = np.random.default_rng(11)
    for C, T, dt in ((1, 1024, np.float64), (3, 1025, np.float64), (2, 1840, np.float64), (5, 5003, np.float64),
                     (4, 2449, np.float32), (2, 24000, np.float32)):
        x = rng.standard_normal((C, T)).astype(dt)
        x[-1, : T // 3] = 0.0                                          # a stretch of silence in one channel
        tol = 1e-9 if dt == np.float64 else 1e-5
        for env in (True, False):
            ref = sg.hilbert_filter(x.astype(np.float64) if dt == np.float64 else x, 400, [70., 150.], envelope=env)
            for bl in ("ols", "ols_full"):
                monkeypatch.setenv("TONAL_HILBERT", bl)
                out = ff.hilbert_filter(x, 400, freq_ranges=[70., 150.], envelope=env)
                assert out.dtype == np.float64 and out.shape == (C, T)
                assert rel(out, ref) < tol, (C, T, dt, env, bl, rel(out, ref))
                if dt == np.float32 and bl == "ols":
                    # float32 recordings: fp64 math by default (in the reference only the forward FFT stays complex64, the
                    # product / inverse / |.| / mean run in fp64); TONAL_HILBERT_F32=1 opts into fp32 transforms end to end -
                    # both inside the golden's 1e-5, 1e-5 apart at most, and the default is the closer one to the fp64 result
                    monkeypatch.setenv("TONAL_HILBERT_F32", "1")
                    out32 = ff.hilbert_filter(x, 400, freq_ranges=[70., 150.], envelope=env)
                    monkeypatch.delenv("TONAL_HILBERT_F32")
                    assert rel(out32, ref) < tol and rel(out, out32) < tol and not np.array_equal(out, out32)
                    ref64 = sg.hilbert_filter(x.astype(np.float64), 400, [70., 150.], envelope=env)
                    assert rel(out, ref64) < 1e-9 and rel(out, ref64) <= rel(out32, ref64)
    monkeypatch.delenv("TONAL_HILBERT", raising=False)


def test_cnn_full_size_batch_properties(dev):
    """North-star shape (128 ch x 400, batch 256), small LSTM width to keep the test light:
    eval-mode outputs are per-sample (permutation equivariance, sub-batch equality)."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 128, 400, lstm_channels=1, dropout=0.5).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(256, 128, 400, device=dev, generator=g)
    tones = torch.randint(0, 4, (256,), generator=torch.Generator().manual_seed(1))
    syls = torch.randint(0, 2, (256,), generator=torch.Generator().manual_seed(2))
    lab = gi.tone_dynamics(tones, syls).to(dev)
    with torch.no_grad():
        out = model(x, lab)
        perm = torch.randperm(256, generator=torch.Generator().manual_seed(3)).to(dev)
        out_p = model(x[perm].contiguous(), lab[perm].contiguous())
        out_s = model(x[:8].contiguous(), lab[:8].contiguous())
    assert out.shape == (256, 80) and torch.isfinite(out).all()
    assert float((out[perm] - out_p).abs().max() / out.abs().max()) < 1e-5
    assert float((out[:8] - out_s).abs().max() / out.abs().max()) < 1e-5


def test_other_signal_steps_match_reference_golden(dev):
    from decode_tonal_langauge_amd.preprocess.signal import (car_rereference, channel_zscore, rolling_zscore,
                                                             zscore_rereference)
    g = np.load(os.path.join(GOLD, "g7_steps.npz"))
    x = np.random.default_rng(7).standard_normal((5, 900)) * 3.0 + 1.5
    assert abs(gi.checksum(x) - float(g["in_checksum"])) < 1e-9 * float(g["in_checksum"])
    out = channel_zscore.run(x, Namespace())
    assert out.dtype == np.float64 and rel(out, g["channel_zscore"]) < 1e-12
    o32 = channel_zscore.run(x.astype(np.float32), Namespace())
    assert o32.dtype == np.float32 and rel(o32, g["channel_zscore_f32"]) < 1e-5
    assert rel(zscore_rereference.run(x, Namespace(rereference_interval=[0.25, 1.5], signal_freq=200)),
               g["zscore_rereference"]) < 1e-12
    assert rel(car_rereference.run(x, Namespace(exclude_channels=[1, 3])), g["car"]) < 1e-12
    rz = rolling_zscore.run(x, Namespace(window_length=0.25, signal_freq=200))
    assert np.array_equal(np.isnan(rz), np.isnan(g["rolling"]))
    assert rel(np.nan_to_num(rz), np.nan_to_num(g["rolling"])) < 1e-10
    xn = x.copy()
    xn[2, 100:130] = np.nan
    rzn = rolling_zscore.run(xn, Namespace(window_length=0.25, signal_freq=200, preserve_nans=False))
    assert rel(rzn, g["rolling_nan"]) < 1e-10
    with pytest.raises(ValueError, match="out of bounds"):
        zscore_rereference.run(x, Namespace(rereference_interval=[0., 100.], signal_freq=200))
    with pytest.raises(ValueError, match="greater than 1"):
        rolling_zscore.run(x, Namespace(window_length=0.001, signal_freq=200))
    with pytest.raises(ValueError, match="invalid channel"):
        car_rereference.run(x, Namespace(exclude_channels=[9]))


def test_trainer_with_deep_classifiers(dev):
    """Config-5 style wiring: CNN / CNNRNN classifiers (stock PyTorch-ROCm forward) feeding the fused
    SynthesisModelCNN step through SynthesisTrainer."""
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier, SynthesisModelCNN, SynthesisTrainer
    torch.manual_seed(0)
    T = 160
    syl = CNNClassifier(input_channels=4, input_length=T, n_classes=2)
    tone = CNNRNNClassifier(input_channels=4, input_length=T, n_classes=4, lstm_dim=320)
    model = SynthesisModelCNN(80, 8, T)
    tr = SynthesisTrainer(model, tone, syl, gi.TONE_MAP, device=dev, verbose=False)
    g = torch.Generator().manual_seed(0)
    ds = torch.utils.data.TensorDataset(torch.randn(12, 8, T, generator=g), torch.randn(12, 4, T, generator=g),
                                        torch.randn(12, 4, T, generator=g), 10 * torch.randn(12, 80, generator=g))
    hist = tr.train(torch.utils.data.DataLoader(ds, batch_size=6), 2, verbose=False)
    assert len(hist) == 2 and all(np.isfinite(v) for h in hist for v in h)
    mcd, recon, origin = tr.evaluate(torch.utils.data.DataLoader(ds, batch_size=6))
    assert np.isfinite(mcd) and recon.shape == (12, 80) and origin.shape == (12, 80)
    # the reference CLI's default: train_classifiers=True puts the classifiers in train mode (dropout active) - they
    # must still run on the HIP engines, with a fresh dropout seed per call
    tr2 = SynthesisTrainer(model, tone, syl, gi.TONE_MAP, device=dev, verbose=False, train_classifiers=True)
    tone._hip = syl._hip = None
    s0, t0 = syl._drop_calls, tone._drop_calls
    hist = tr2.train(torch.utils.data.DataLoader(ds, batch_size=6), 1, verbose=False)
    assert tone.training and syl.training and tone._hip is not None and syl._hip is not None
    assert syl._drop_calls == s0 + 2 and tone._drop_calls == t0 + 2 and all(np.isfinite(v) for h in hist for v in h)


@pytest.mark.parametrize("cfg", [
    # (B, C, T, out_dim, lstm_channels, conv_channels, L)
    (1, 4, 100, 80, 6, 64, 5),        # batch of one
    (3, 2, 61, 12, 2, 8, 1),          # L = 1 (no recurrence), odd T, tiny widths
    (5, 6, 77, 7, 2, 10, 4),          # output_dim and conv_channels not multiples of 4
    (9, 3, 130, 33, 4, 20, 3),        # odd channel count, odd output_dim
    (130, 1, 64, 8, 4, 4, 2),         # one ECoG channel, batch > one row tile
])
def test_cnn_edge_shapes_against_oracle(dev, cfg):
    """Ragged / minimal shapes: forward and every parameter gradient against the CPU oracle."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from oracle import synthesis_oracle as so
    B, C, T, D, lc, cc, L = cfg
    torch.manual_seed(B * 1000 + T)
    model = SynthesisModelCNN(D, C, T, lstm_channels=lc, conv_channels=cc, dropout=0.0)
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, C, T, generator=g)
    lab = torch.randint(0, 4, (B, 2, L), generator=g).float()
    tgt = torch.randn(B, D, generator=g)
    model.to(dev).train()
    out = model(x.to(dev), lab.to(dev))
    ((out - tgt.to(dev)) ** 2).mean().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    # oracle with the model's widths (its helper assumes the reference's names/shapes only)
    ref = so.cnn_forward(leaves, x, lab, decisions=hip_decisions(model._engine, B, C))
    ref_grads = dict(zip(leaves, torch.autograd.grad(((ref - tgt) ** 2).mean(), list(leaves.values()))))
    assert rel(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    for k, p in model.named_parameters():
        gref = ref_grads[k].numpy()
        if np.abs(gref).max() < 1e-12:
            assert float(p.grad.abs().max()) < 1e-9, k
        else:                                         # (shared LeakyReLU' / arg-max branches: tests/branch_planes.py)
            assert rel_l2(p.grad.cpu().numpy(), gref) < 5e-5, k


def test_cnn_rejects_bad_inputs(dev):
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    model = SynthesisModelCNN(80, 4, 100).to(dev)
    with pytest.raises(ValueError, match="expected ECoG input"):
        model(torch.randn(2, 5, 100, device=dev), torch.randn(2, 2, 5, device=dev))
    with pytest.raises(ValueError, match="expected labels"):
        model(torch.randn(2, 4, 100, device=dev), torch.randn(2, 3, 5, device=dev))
    with pytest.raises((ValueError, RuntimeError)):
        SynthesisModelCNN(80, 4, 20)           # the conv stack consumes more samples than there are
    with pytest.raises(ValueError, match="multiple of 4"):
        SynthesisModelCNN(80, 3, 100, lstm_channels=1)
    with pytest.raises(ValueError, match="negative_slope"):
        SynthesisModelCNN(80, 4, 100, negative_slope=-0.1)


def test_glds_variant_matches_default_kernel(dev, monkeypatch):
    """The direct-to-LDS NT kernel (default since round 4) and the register-staged one (TONAL_GLDS=0) give the same forward and
    gradients."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    xs, _t, _s, labs, tg = gi.train_batches(1, 6, 8, 200, seed=5)
    outs, grads = [], []
    for flag in ("0", "1"):
        monkeypatch.setenv("TONAL_GLDS", flag)
        torch.manual_seed(1)
        model = SynthesisModelCNN(80, 8, 200, dropout=0.0).to(dev).train()
        out = model(xs[0].to(dev), labs[0].to(dev))
        (out - tg[0].to(dev)).abs().mean().backward()
        outs.append(out.detach().cpu().numpy())
        grads.append({k: p.grad.cpu().numpy() for k, p in model.named_parameters()})
    assert rel(outs[1], outs[0]) < 1e-5
    for k in grads[0]:
        assert rel_l2(grads[1][k], grads[0][k]) < 5e-3, k


@pytest.mark.parametrize("shape", [(1, 64, 16), (130, 72, 48), (1000, 200, 128), (517, 128, 272), (256, 64, 64), (70, 20, 32), (300, 36, 80)])
def test_one_tap_nt_gemm_forms_match_matmul(dev, shape, monkeypatch):
    """``tl_gemm_nt_window`` with one tap (conv4 / conv5 / the 1x1 stack / Linear: reference models/synthesis_models.py:99-131)
    on ragged shapes - partial row and column tiles, 1 .. 17 K-stages, split-K - for the register-staged kernel and the
    direct-to-LDS kernel (default): STORE (+ split-K slabs), bias + LeakyReLU, and the input-gradient MASK from the stage input
    (whose values a tile requests before its first store), against float64 matmul."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._classifier_engine import _launch_nt
    from decode_tonal_langauge_amd._lib import LOAD_DIRECT, EPI_STORE, EPI_LRELU, EPI_MASK, ptr
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M + 3, K + 4, device=dev, generator=g)           # lda > K, rows behind M never read as outputs
    W = torch.randn(N, K + 8, device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g)
    aux = torch.randn(M, N + 4, device=dev, generator=g)
    ref = A[:M, :K].double() @ W[:, :K].double().t()
    scale = float(ref.abs().max())
    for glds in ("0", "1"):
        monkeypatch.setenv("TONAL_GLDS", glds)
        kw = dict(A=ptr(A), Bw=ptr(W), M=M, A_rows=M + 3, N=N, K=K, lda=K + 4, ldb=K + 8, loader=LOAD_DIRECT)
        out = torch.full((M, N + 4), float("nan"), device=dev)
        _launch_nt(lib, out=ptr(out), ldo=N + 4, epilogue=EPI_STORE, **kw)
        assert float((out[:, :N].double() - ref).abs().max()) < 2e-6 * scale and bool(torch.isnan(out[:, N:]).all())
        out.fill_(float("nan"))
        _launch_nt(lib, out=ptr(out), ldo=N + 4, epilogue=EPI_LRELU, bias=ptr(bias), slope=0.1, **kw)
        y = ref + bias.double()
        assert float((out[:, :N].double() - torch.where(y > 0, y, 0.1 * y)).abs().max()) < 2e-6 * scale
        out.fill_(float("nan"))
        _launch_nt(lib, out=ptr(out), ldo=N + 4, epilogue=EPI_MASK, aux=ptr(aux), ldaux=N + 4, slope=0.1, **kw)
        assert float((out[:, :N].double() - torch.where(aux[:, :N] > 0, ref, 0.1 * ref)).abs().max()) < 2e-6 * scale
        assert bool(torch.isnan(out[:, N:]).all())
        nk = K // 16
        for sk in sorted({1, min(2, nk), min(3, nk), nk}):
            slab = torch.full((sk, M, N), float("nan"), device=dev)
            _launch_nt(lib, out=ptr(slab), ldo=N, epilogue=EPI_STORE, splitk=sk, slab_stride=M * N, **kw)
            assert float((slab.double().sum(0) - ref).abs().max()) < 2e-6 * scale, (glds, sk)


def test_winograd_kernels_match_direct_kernels(dev, monkeypatch):
    """The Winograd conv kernels (F(4,3) and the default F(6,3), both on pre-transformed operands) against the direct MFMA
    kernels (TONAL_WINO=0): same forward, same gradients, on a ragged shape (row tiles, time padding and the last
    reduction chunk are all partial)."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from decode_tonal_langauge_amd import _lib
    lib = _lib.load()
    for (B, C, T) in ((6, 8, 200), (5, 4, 131)):
        g = torch.Generator().manual_seed(T)
        x = torch.randn(B, C, T, generator=g)
        lab = torch.randint(0, 4, (B, 2, 5), generator=g).float()
        tgt = torch.randn(B, 80, generator=g)
        outs, grads = [], []
        for flag in ("0", "4", "6"):              # direct, F(4,3), default (F(6,3)), the last two on pre-transformed operands
            monkeypatch.setenv("TONAL_WINO", flag)
            torch.manual_seed(1)
            model = SynthesisModelCNN(80, C, T, dropout=0.0).to(dev).train()
            assert model._engine.wino43 == (flag in "46") and model._engine.wino63 == (flag == "6")
            assert all(model._engine._v43(st) == (flag == "4") or model._engine.wino63 for st in model._engine.stages[:2])
            out = model(x.to(dev), lab.to(dev))
            (out - tgt.to(dev)).abs().mean().backward()
            outs.append(out.detach().cpu().numpy())
            grads.append({k: p.grad.cpu().numpy() for k, p in model.named_parameters()})
        for v in (1, 2):
            assert rel(outs[v], outs[0]) < 1e-5
            for k in grads[0]:
                # (the gradients of the ecog stages at this loss are sums over every row that cancel to 1e-9 .. 1e-7: a handful of
                # arg-max / sign ties decided the other way by a different rounding moves them by several parts per
                # thousand; the stage tests hold the kernels to 1e-5 on identical inputs, the reference goldens the model)
                r = rel_l2(grads[v][k], grads[0][k])
                if r > 2e-3:
                    print(f"(B, C, T) = {(B, C, T)}, form {('0', '4', '6')[v]}: {k} {r:.2e} from the direct kernels")
                assert r < (2e-2 if k.startswith("ecog_conv_block.") else 5e-3), k
    # the C ABI refuses shapes the Winograd form does not cover instead of computing garbage
    p = _lib.NtParams()
    dummy = torch.zeros(64, device=dev)
    for k in ("A", "Bw", "out"):
        setattr(p, k, dummy.data_ptr())
    p.M, p.N, p.K, p.lda, p.ldb, p.J, p.Tp, p.loader = 4, 32, 24, 24, 24, 3, 4, 2          # K % 16 != 0
    assert lib.tl_conv3_wino43v_nt(C_.byref(p), None) != 0
    assert b"wino43v_nt" in lib.tl_last_error()


def test_conv4_input_gradient_forms_agree_on_the_whole_model(dev, monkeypatch):
    """Round 5: conv4's input gradient on the NT63 kernel (the default: its epilogue writes conv3's backward operands) against
    the one-tap GEMM + ``tl_wino63_unpool_yvd`` form of round 4, whole model, identical forward (the bit words are the same, so
    no tie can be decided differently): every gradient to 1e-5.  15 sequences: the last hex of the NT63 form is half empty."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    for (B, C, T) in ((3, 5, 400), (2, 4, 200)):
        g = torch.Generator().manual_seed(T + 1)
        x = torch.randn(B, C, T, generator=g)
        lab = torch.randint(0, 4, (B, 2, 5), generator=g).float()
        tgt = torch.randn(B, 80, generator=g)
        grads = {}
        for form in ("gemm", "nt63"):
            monkeypatch.setenv("TONAL_CONV4_DGRAD", form)
            torch.manual_seed(1)
            model = SynthesisModelCNN(80, C, T, dropout=0.0).to(dev).train()
            eng = model._engine
            assert eng.wino63 and eng.gy4 == (form == "nt63")
            out = model(x.to(dev), lab.to(dev))
            (out - tgt.to(dev)).abs().mean().backward()
            assert (3 in eng.G) == (form == "gemm")                     # the NT63 form never stores conv3's gradient rows
            grads[form] = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
        for k in grads["gemm"]:
            assert rel_l2(grads["nt63"][k], grads["gemm"][k]) < 1e-5, (B, C, T, k)


@pytest.mark.parametrize("wino", ["6", "4", "0"])
def test_cnn_classifier_hip_forward_matches_module_graph(dev, wino, monkeypatch):
    """CNNClassifier inference on the HIP conv kernels vs the same module's stock PyTorch graph: with its leading pooled 3-tap
    stages on the F(6,3) V-form kernels (round 5, the default), on the F(4,3) V form (one transform pass per stage) and on the
    direct MFMA kernels."""
    from decode_tonal_langauge_amd.models import CNNClassifier
    monkeypatch.setenv("TONAL_WINO", wino)
    torch.manual_seed(0)
    for (C, T, B) in ((4, 160, 5), (3, 233, 9), (8, 400, 7)):
        clf = CNNClassifier(input_channels=C, input_length=T, n_classes=3).to(dev).eval()
        x = torch.randn(B, C, T, device=dev)
        with torch.no_grad():
            hip = clf(x)
            hip2 = clf(x)                                   # (a second pass through the cached buffers / packed weights)
        assert clf._hip is not None, "HIP path was not taken"
        assert clf._hip.n63 == (3 if wino == "6" else 0)
        ref = clf.classifier(clf.feature_extractor(x.unsqueeze(1).permute(0, 1, 3, 2)))
        assert hip.shape == ref.shape == (B, 3)
        assert float((hip - ref.detach()).abs().max()) < 1e-5 and torch.equal(hip, hip2)


def test_downsample_matches_reference_golden(dev):
    from decode_tonal_langauge_amd.preprocess.signal import downsample
    g = np.load(os.path.join(GOLD, "g7_steps.npz"))
    x = np.random.default_rng(7).standard_normal((5, 900)) * 3.0 + 1.5
    prm = Namespace(signal_freq=1000, downsample_freq=400)
    out = downsample.run(x, prm)
    assert prm.signal_freq == 400 and out.shape == (5, 360) and out.dtype == np.float64
    assert rel(out, g["downsample"]) < 1e-11
    o32 = downsample.run(x.astype(np.float32), Namespace(signal_freq=1000))
    assert o32.dtype == np.float32 and rel(o32, g["downsample_f32"]) < 1e-5
    up = downsample.run(x[:, :601], Namespace(signal_freq=300, downsample_freq=400))
    assert up.shape == g["downsample_up"].shape and rel(up, g["downsample_up"]) < 1e-11
    # a long recording with an awkward (prime) length, device resident: the resampled band-limited
    # signal interpolates the original (round trip up then down returns the input)
    t = torch.randn(3, 10007, device=dev, dtype=torch.float64)
    up2 = downsample.resample(t, 20014)
    back = downsample.resample(up2, 10007)
    assert float((back - t).abs().max()) < 1e-9


def test_cnnrnn_classifier_hip_trunk_matches_module_graph(dev):
    """CNNRNNClassifier inference: convolutional trunk on the HIP kernels (7-tap convolutions) vs the same
    module's stock PyTorch graph."""
    from decode_tonal_langauge_amd.models import CNNRNNClassifier
    torch.manual_seed(0)
    for (C, T, B, lstm_dim) in ((4, 100, 3, 200), (3, 131, 5, 131)):
        clf = CNNRNNClassifier(input_channels=C, input_length=T, n_classes=4, lstm_dim=lstm_dim).to(dev).eval()
        x = torch.randn(B, C, T, device=dev)
        with torch.no_grad():
            hip = clf(x)
        assert clf._hip is not None, "HIP path was not taken"
        xt = x.permute(0, 2, 1)
        h1 = clf.lstm1(xt)[0][:, -1, :]
        a = clf.conv_pool_block1(xt.unsqueeze(1))
        b = clf.conv_pool_block2(h1.reshape(B, 1, T, -1))
        f = clf.conv_block3(torch.cat((b, a), dim=3))
        with torch.no_grad():
            feat = clf._hip.features(x, h1.detach(), *[(m.weight, m.bias) for m in (clf.conv_pool_block1[0],
                                     clf.conv_pool_block2[0], clf.conv_block3[0], clf.conv_block3[2])])
        fr = f.detach().contiguous().view(B, f.shape[2], -1)
        assert feat.shape == fr.shape
        assert float((feat - fr).abs().max()) < 1e-4 * max(1.0, float(fr.abs().max()))
        ref = torch.sigmoid(clf.output(clf.lstm2(fr)[0][:, -1, :]))
        assert hip.shape == ref.shape == (B, 4)
        assert float((hip - ref.detach()).abs().max()) < 1e-4


@pytest.mark.parametrize("widths", [(64, 96, 32), (32, 160, 64), (96, 32, 96), (128, 64, 64)])
def test_winograd_stage_kernels_on_ragged_widths(dev, widths):
    """Stage kernels at the C-ABI level on channel counts that are not multiples of the column tiles
    (tile tails in N, one- and two-chunk K loops): the F(4,3) V form (where a stage's widths allow it; the direct kernels
    elsewhere) against the direct kernels."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    c1, c2, c3 = widths
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    B, C, T = 3, 5, 236
    g = torch.Generator(device=dev).manual_seed(sum(widths))
    res = {}
    for mode in ("0", "4"):
        eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
        assert not eng.wino63                      # (widths the F(6,3) kernels do not cover: the F(4,3) V form / direct kernels)
        eng.wino43, eng.fuse_c1 = mode == "4", False
        eng.wino_vout = False          # stage kernels one at a time on random inputs: every stage reads P
        eng._alloc(B, dev)
        eng._alloc_bwd()
        g.manual_seed(sum(widths))
        for k in sorted(eng.P):
            eng.P[k].normal_(generator=g)
        for k in sorted(eng.G):
            eng.G[k].normal_(generator=g)
        for k in sorted(eng.bits):
            eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
            eng.sbits[k].random_(-2**31, 2**31 - 1, generator=g)
        out = []
        for si in (2, 3):
            st = eng.stages[si - 2]
            w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.05
            b = torch.randn(st.cout, device=dev, generator=g) * 0.1
            gw, gb = torch.zeros_like(w), torch.zeros_like(b)
            keep = (eng.P[si].clone(), eng.bits[si].clone(), eng.sbits[si].clone())
            eng._v_ready = {}
            eng.stage_wgrad(st, gw, gb)            # (the V form's input gradient reads Vd, which the weight gradient writes)
            eng.stage_dgrad(st, w)
            dg = eng.G[si - 1].clone()
            eng._v_ready = {}
            eng.stage_forward(st, w, b)
            out.append((eng.P[si].clone(), eng.bits[si].clone(), dg, gw, gb))
            eng.P[si].copy_(keep[0]); eng.bits[si].copy_(keep[1]); eng.sbits[si].copy_(keep[2])
        res[mode] = out
    for mode in ("4",):
        for (p0, b0, d0, w0, g0), (p1, b1, d1, w1, g1) in zip(res["0"], res[mode]):
            assert rel(p1.cpu().numpy(), p0.cpu().numpy()) < 2e-5
            flips = (b0 ^ b1)
            assert int((flips != 0).sum()) <= 2                       # arg-max ties only
            assert rel_l2(d1.cpu().numpy(), d0.cpu().numpy()) < 1e-5
            assert rel_l2(w1.cpu().numpy(), w0.cpu().numpy()) < 1e-5
            assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("shape", [(3, 5, 236, (128, 128, 64)), (2, 3, 400, (128, 256, 128)), (1, 1, 44, (128, 128, 128)),
                                   (7, 3, 100, (64, 192, 64))])
def test_forward_epilogue_writes_the_next_stage_operand(dev, shape, monkeypatch):
    """Round 4: the forward kernel of a pooled 3-tap stage writes V = the F(4,3) input transform of its own pooled output
    (epilogue 5 + tl_wino43_v_fixup) instead of the raw rows.  With the raw rows stored as well (store_p1) the V it wrote
    must be the transform of exactly those rows - including the quads that take rows from the next half-wave, the next
    wave, the next tile (fix-up pass) and the quads that end a sequence - and rows / bits must equal the plain POOL launch."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    from decode_tonal_langauge_amd._lib import check, ptr
    monkeypatch.setenv("TONAL_WINO", "4")              # (the F(4,3) kernels; test_f63_* below hold the default form)
    B, C, T, (c1, c2, c3) = shape
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    res = {}
    for vout in (False, True):
        eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
        eng.fuse_c1, eng.wino_vout, eng.store_p1 = False, vout, True
        eng._alloc(B, dev)
        st = eng.stages[0]
        assert eng._writes_v(st) == vout and st.tp_in % 8 == 0
        g = torch.Generator(device=dev).manual_seed(11)
        eng.P[1].normal_(generator=g)
        eng.P[1].view(eng.S, st.tp_in, -1)[:, st.tin:, :] = 0
        w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.05
        b = torch.randn(st.cout, device=dev, generator=g) * 0.1
        eng._v_ready = {}
        eng.stage_forward(st, w, b)
        res[vout] = (eng.P[2].clone(), eng.bits[2].clone(), eng.sbits[2].clone())
        if vout:
            V = eng._v_ready[2]
            nq = eng.S * st.tp_out // 4
            Vref = torch.full_like(V, float("nan"))
            check(eng.lib.tl_wino43_input_transform(ptr(eng.P[2]), ptr(Vref), eng.P[2].shape[0], st.tp_out, st.cout, st.cout,
                                                    st.cout, torch.cuda.current_stream().cuda_stream), "tl_wino43_input_transform")
            assert bool(torch.isfinite(V).all())
            assert torch.allclose(V[:nq], Vref[:nq], rtol=1e-6, atol=1e-6), float((V[:nq] - Vref[:nq]).abs().max())
            assert float(V[nq:].abs().max()) == 0.0 if V.shape[0] > nq else True
            # a second launch into the same buffers: the raw rows the fix-up pass consumed are rewritten, not re-transformed
            eng._v_ready = {}
            eng.stage_forward(st, w, b)
            assert torch.equal(eng._v_ready[2], V)
    for a, b_ in zip(res[False], res[True]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("shape", [(3, 5, 236, (128, 128, 64)), (2, 3, 400, (128, 256, 128)), (1, 1, 44, (128, 128, 128))])
def test_weight_gradient_tilings_agree_bit_for_bit(dev, shape, monkeypatch):
    """The three V-form F(4,3) weight-gradient kernels (64-wide C_in tile; 128-wide with Y staged through registers; 128-wide
    with the Y side by LDS-DMA and the workgroups taking turns at Vd - the product path where C_in % 128 == 0 and C_out % 64
    == 0) keep the same k order per accumulator: weight gradient, bias gradient and the Vd they write must be identical,
    on shapes with ragged last K-steps, sequences shorter than a K-step and empty reduction splits."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    monkeypatch.setenv("TONAL_WINO", "4")
    B, C, T, (c1, c2, c3) = shape
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
    eng.fuse_c1 = False
    eng.wino_vout = False
    eng._alloc(B, dev)
    eng._alloc_bwd()
    g = torch.Generator(device=dev).manual_seed(7)
    for k in sorted(eng.P):
        eng.P[k].normal_(generator=g)
    for k in sorted(eng.G):
        eng.G[k].normal_(generator=g)
    for k in sorted(eng.bits):
        eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
    for si in (2, 3):
        st = eng.stages[si - 2]
        assert eng._v43(st)
        eng.P[si - 1].view(eng.S, st.tp_in, -1)[:, st.tin:, :] = 0
        res = {}
        for bm in (64, 127, 128):
            eng.tn_bm = bm
            eng._v_ready = {}
            if si in eng.Vd:
                eng.Vd[si].fill_(float("nan"))
            gw = torch.zeros(st.cout, st.cin, 3, 1, device=dev)
            gb = torch.zeros(st.cout, device=dev)
            eng.stage_wgrad(st, gw, gb)
            nq = eng.S * st.tp_in // 4
            res[bm] = (gw.clone(), gb.clone(), eng.Vd[si][:nq].clone())
        assert bool(torch.isfinite(res[128][2]).all()) and float(res[128][0].abs().max()) > 0
        for bm in (127, 128):
            for a, b in zip(res[64], res[bm]):
                assert torch.equal(a, b), (si, bm)


def test_multi_tensor_nadam_equals_per_tensor_launches_and_torch(dev):
    """tl_nadam_multi (one launch for a parameter list) is bit-identical to one tl_nadam launch per tensor and
    follows torch.optim.NAdam (reference models/synthesis_trainer.py:131-137): sizes below / above a chunk,
    not multiples of 4, a parameter that skips a step (its schedule then differs from the others')."""
    from decode_tonal_langauge_amd.optim import FusedNAdam
    sizes = [(80,), (3, 5), (4096,), (4097,), (129, 67), (1, 1), (70000,)]
    gen = torch.Generator().manual_seed(5)
    init = [torch.randn(*sz, generator=gen) for sz in sizes]
    pm = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pt = [torch.nn.Parameter(t.clone().double()) for t in init]
    om = FusedNAdam(pm, lr=5e-3, weight_decay=0.004)
    os_ = FusedNAdam(ps, lr=5e-3, weight_decay=0.004)
    os_.multi_tensor = False
    ot = torch.optim.NAdam(pt, lr=5e-3, weight_decay=0.004)
    assert om.multi_tensor
    for step in range(4):
        grads = [torch.randn(*sz, generator=gen) for sz in sizes]
        skip = 2 if step == 1 else -1                      # parameter 2 has no gradient in step 1
        for plist in (pm, ps):
            for i, (p, g) in enumerate(zip(plist, grads)):
                p.grad = None if i == skip else g.to(dev)
        for i, (p, g) in enumerate(zip(pt, grads)):
            p.grad = None if i == skip else g.double()
        om.step(); os_.step(); ot.step()
    for a, b, c in zip(pm, ps, pt):
        assert torch.equal(a.detach(), b.detach())
        assert float((a.detach().cpu().double() - c.detach()).abs().max()) < 2e-6
    assert len(om._tables) >= 2                            # full list, list without the skipped tensor, ...


def test_lowrank_nadam_equals_dense_nadam(dev):
    """FusedNAdam with gradient factors (tl_nadam_lowrank) == FusedNAdam on the materialised gradient."""
    from decode_tonal_langauge_amd.optim import FusedNAdam
    g = torch.Generator(device=dev).manual_seed(3)
    for rows, cols, kr in ((70, 260, 33), (64, 512, 5), (33, 4, 0)):
        w0 = torch.randn(rows, cols, device=dev, generator=g)
        pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w0.clone())
        oa = FusedNAdam([pa], lr=5e-3, weight_decay=0.004)
        ob = FusedNAdam([pb], lr=5e-3, weight_decay=0.004)
        for step in range(3):
            if kr:
                fa = torch.randn(kr, rows, device=dev, generator=g)
                fb = torch.randn(kr, cols, device=dev, generator=g)
                dense = fa.t() @ fb
            else:
                fa = fb = None
                dense = torch.zeros(rows, cols, device=dev)
            oa.step(grads={pa: dense.contiguous()}, grad_scale=0.5)
            ob.step(grads={}, lowrank={pb: (fa, fb)}, grad_scale=0.5)
        assert float((pa - pb).detach().abs().max()) < 2e-6 * max(1.0, float(pa.detach().abs().max()))
        sa, sb = oa.state[pa], ob.state[pb]
        assert sa["step"] == sb["step"] == 3
        assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) < 1e-5
        assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) < 1e-4 * float(sa["exp_avg_sq"].abs().max())
    with pytest.raises(RuntimeError, match="rank"):
        ob.step(grads={}, lowrank={pb: (torch.zeros(65, 33, device=dev), torch.zeros(65, 4, device=dev))})


def test_sparse_tone_mapping_keeps_the_label_lstm_finite(dev):
    """A tone mapping with a key gap above the classifier's classes (keys 0..3 and "5" beside a 4-class tone model): the
    (tone, syllable) pair table must hold the predictable classes only.  Rows for the missing key 4 would be NaN; no
    batch element gathers them, but the label LSTM unrolls over every table row and 0 * NaN in its backward pass would
    turn W_hh, W_ih and the biases NaN after one step.  The update must equal the one under the dense mapping."""
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    B, C, T = 6, 4, 100
    xs, _t, _s, _labs, tg = gi.train_batches(2, B, C, T, seed=3)
    gen = torch.Generator().manual_seed(9)
    xt = [torch.randn(B, 8, T, generator=gen) for _ in range(2)]
    xsyl = [torch.randn(B, 8, T, generator=gen) for _ in range(2)]
    finals = []
    for mapping in (gi.TONE_MAP, dict(gi.TONE_MAP, **{"5": [2, 2, 2, 2, 2]})):
        torch.manual_seed(0)
        model = SynthesisModelCNN(80, C, T, dropout=0.0)
        torch.manual_seed(1)
        tone, syl = LogisticRegressionClassifier(8 * T, 4), LogisticRegressionClassifier(8 * T, 2)
        tr = SynthesisTrainer(model, tone, syl, mapping, device=dev, verbose=False)
        assert tr._pair_table is not None and tr._pair_table.shape[0] == 8 and not torch.isnan(tr._pair_table).any()
        for i in range(2):
            tr.train_step(xs[i], xsyl[i], xt[i], tg[i])
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k, v in finals[1].items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, finals[0][k]), k


@pytest.mark.parametrize("wino", ["6", "4", "0"])
def test_cnn_training_trajectory_matches_reference_golden(dev, monkeypatch, wino):
    """G14: 30 NAdam steps of the reference's SynthesisModelCNN(80, 16, 200, dropout=0) on 30 seeded batches - the HIP
    path's L1 loss, MCD and mel MSE mean((out - target)^2) stay within 1e-3 of the reference at EVERY step (the bound
    BASELINE.json's north_star states), for the default kernels (F(4,3) on V), F(2,3) and the direct MFMA form."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    g = np.load(os.path.join(GOLD, "g14_cnn_trajectory.npz"))
    D, C, T, B, N = (int(v) for v in g["dims"])
    xs, _t, _s, labs, tg = gi.train_batches(N, B, C, T, seed=int(g["data_seed"]))
    assert abs(gi.checksum(*xs, *labs, *tg) - float(g["in_checksum"])) < 1e-6 * float(g["in_checksum"])
    monkeypatch.setenv("TONAL_WINO", wino)
    torch.manual_seed(int(g["seed"]))
    model = SynthesisModelCNN(D, C, T, dropout=0.0)
    tr = _trainer(model, dev, T)
    assert model._engine.wino43 == (wino in "46") and model._engine.wino63 == (wino == "6")
    model.train()
    worst = 0.0
    obs = {"loss": 0.0, "mcd": 0.0, "mse": 0.0, "out": 0.0}
    for s in range(N):
        tr._fused_step(xs[s].to(dev), labs[s].to(dev), tg[s].to(dev))
        st = tr._stats.cpu().numpy()
        out = tr._last_out.double().cpu() if hasattr(tr, "_last_out") else None
        assert abs(st[2] - g["losses"][s]) < 1e-3 * g["losses"][s], (s, st[2], g["losses"][s])
        assert abs(st[3] - g["mcds"][s]) < 1e-3 * g["mcds"][s], (s, st[3], g["mcds"][s])
        assert out is not None
        mse = float(((out - tg[s].double()) ** 2).mean())
        worst = max(worst, abs(mse - g["mses"][s]) / g["mses"][s])
        assert abs(mse - g["mses"][s]) < 1e-3 * g["mses"][s], (s, mse, g["mses"][s])
        assert rel(out.numpy(), g["outs"][s]) < 5e-3, s          # element-wise, late in the trajectory
        obs = {"loss": max(obs["loss"], abs(st[2] - g["losses"][s]) / g["losses"][s]),
               "mcd": max(obs["mcd"], abs(st[3] - g["mcds"][s]) / g["mcds"][s]), "mse": worst,
               "out": max(obs["out"], rel(out.numpy(), g["outs"][s]))}
    print(f"TONAL_WINO={wino}: worst relative mel-MSE deviation over {N} steps {worst:.2e}")
    from tests.parity_record import record
    record(f"G14 thirty steps at 16x200 (TONAL_WINO={wino})", {"worst_over_steps." + k: float(v) for k, v in obs.items()})


def test_hilbert_low_band_at_raw_rate_matches_reference_golden(dev, monkeypatch):
    """G15: the reference's hilbert_filter for a 1-4 Hz band at 3 kHz - Gaussian kernels thousands of samples long, more
    than the time-domain kernel's LDS window: the DFT-domain path (tl_hilbert_fft, Bluestein over Stockham passes) -
    and the high-gamma band at that rate on both paths; G6 (400 Hz) through the DFT-domain path as well."""
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    g = np.load(os.path.join(GOLD, "g15_hilbert_low_band.npz"))
    fs = int(g["fs"])
    x = np.random.default_rng(15).standard_normal((2, 9000))
    assert abs(float(np.abs(x).sum()) - float(g["x_checksum"])) < 1e-9
    assert rel(ff.hilbert_filter(x, fs, [1.0, 4.0]), g["low_env"]) < 1e-9
    assert rel(ff.hilbert_filter(x, fs, [1.0, 4.0], envelope=False), g["low_real"]) < 1e-9
    assert rel(ff.hilbert_filter(x.astype(np.float32), fs, [1.0, 4.0]), g["low_env_f32"]) < 1e-5     # reference: complex64
    assert rel(ff.hilbert_filter(x, fs, [70.0, 150.0]), g["hg_env"]) < 1e-9                          # time-domain taps
    with monkeypatch.context() as m:
        m.setenv("TONAL_HILBERT", "taps")
        with pytest.raises(ValueError, match="taps"):
            ff.hilbert_filter(x, fs, [1.0, 4.0])
    monkeypatch.setenv("TONAL_HILBERT", "fft")
    assert rel(ff.hilbert_filter(x, fs, [70.0, 150.0]), g["hg_env"]) < 1e-9                          # DFT domain
    g6 = np.load(os.path.join(GOLD, "g6_signal.npz"))
    x6, _x2 = gi.g6_inputs()
    assert rel(ff.hilbert_filter(x6, 400, freq_ranges=[70., 150.]), g6["hilbert"]) < 1e-9
    assert rel(ff.hilbert_filter(x6, 400, freq_ranges=[70., 150.], envelope=False), g6["hilbert_real"]) < 1e-9
    xt = torch.from_numpy(x).to(dev)                                                                 # resident input
    assert rel(ff.hilbert_filter(xt, fs, [1.0, 4.0]).cpu().numpy(), g["low_env"]) < 1e-9


def _hip_dropout_mask(dev, numel, p, seed):
    """The keep mask tl_dropout_scale draws for a buffer of ``numel`` elements: the kernel applied to ones."""
    from decode_tonal_langauge_amd import _lib
    m = torch.ones(numel, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().tl_dropout_scale(m.data_ptr(), numel, float(p), int(seed), torch.cuda.current_stream().cuda_stream),
               "tl_dropout_scale")
    return m


def test_deep_classifiers_train_mode_dropout_stays_on_hip(dev):
    """The reference CLI trains with ``train_classifiers=True`` by default (train_synthesizer.py:275-284), i.e. the deep
    classifiers run in train mode with their Dropout active (models/deep_classifiers.py:81,258; synthesis_trainer.py:
    193-195).  They must stay on the HIP kernels then.  torch's Philox draws cannot be replayed, so - as for the
    synthesis model's dropout - the HIP keep mask is read back (values {0, 1/(1-p)}, keep rate, a fresh mask per call)
    and fed to the module's own graph with nn.Dropout replaced by that mask: outputs must agree."""
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier
    torch.manual_seed(0)
    # ---- CNNClassifier: Dropout on the (B, 256, lat, C) feature map ----
    C, T, B, p = 4, 160, 6, 0.5
    clf = CNNClassifier(input_channels=C, input_length=T, n_classes=3, dropout_rate=p).to(dev).train()
    x = torch.randn(B, C, T, device=dev)
    with torch.no_grad():
        hip = clf(x)
        hip2 = clf(x)
    eng = clf._hip
    assert eng is not None and clf._last_seed != 0
    assert float((hip - hip2).abs().max()) > 0                       # a new mask per forward
    feat = eng.P[eng.stages[-1].idx]
    m = _hip_dropout_mask(dev, feat.numel(), p, clf._last_seed).view(B, C, eng.tp_last, eng.ld_last)
    vals = torch.unique(m)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
    keep = float((m > 0).float().mean())
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / m.numel()) ** 0.5
    mask_t = m[:, :, :eng.lat, :256].permute(0, 3, 2, 1)            # -> (B, 256, lat, C)
    with torch.no_grad():
        f = clf.feature_extractor[:-1](x.unsqueeze(1).permute(0, 1, 3, 2))
        ref = clf.classifier(f * mask_t)
    assert float((hip2 - ref).abs().max()) < 1e-5
    # ---- CNNRNNClassifier: Dropout on the (B, 256, t', W) map behind the (3,1) pool ----
    C, T, B, lstm_dim = 4, 100, 3, 200
    rnn = CNNRNNClassifier(input_channels=C, input_length=T, n_classes=4, lstm_dim=lstm_dim, dropout=p).to(dev).train()
    x = torch.randn(B, C, T, device=dev)
    with torch.no_grad():
        hip = rnn(x)
    e2 = rnn._hip
    assert e2 is not None and rnn._last_seed != 0
    W, w1, tq = e2.W, e2.w1, e2.tq
    m = _hip_dropout_mask(dev, B * W * tq * 256, p, rnn._last_seed).view(B * W, tq, 256)
    nb = B * w1
    mask_t = torch.cat((m[:nb].view(B, w1, tq, 256), m[nb:].view(B, C, tq, 256)), dim=1).permute(0, 3, 2, 1)   # (B, 256, t', W)
    with torch.no_grad():
        xt = x.permute(0, 2, 1)
        h1 = rnn.lstm1(xt)[0][:, -1, :]
        a = rnn.conv_pool_block1(xt.unsqueeze(1))
        b = rnn.conv_pool_block2(h1.reshape(B, 1, T, -1))
        f = rnn.conv_block3[:-1](torch.cat((b, a), dim=3)) * mask_t
        f = f.contiguous().view(B, f.shape[2], -1)
        ref = torch.sigmoid(rnn.output(rnn.lstm2(f)[0][:, -1, :]))
    assert float((hip - ref).abs().max()) < 1e-4


def test_lite_hip_graph_replay_equals_eager_steps(dev, monkeypatch):
    """SynthesisLite's launch-bound train step is captured into a HIP graph after three eager steps (NAdam coefficients and
    the dropout seed live in device memory and are refreshed per replay).  Ten steps with dropout 0.3 through the graph
    path must leave exactly the parameters, BatchNorm statistics and loss statistics of ten eager steps (TONAL_GRAPH=0)."""
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    B, C, T = 16, 32, 200
    gen = torch.Generator().manual_seed(21)
    data = [(torch.randn(B, C, T, generator=gen), torch.randn(B, 8, T, generator=gen), torch.randn(B, 8, T, generator=gen),
             10 * torch.randn(B, 80, generator=gen)) for _ in range(10)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TONAL_GRAPH", mode)
        torch.manual_seed(0)
        model = SynthesisLite(80, C, T, dropout=0.3)
        torch.manual_seed(1)
        tr = SynthesisTrainer(model, LogisticRegressionClassifier(8 * T, 4), LogisticRegressionClassifier(8 * T, 2), gi.TONE_MAP,
                              device=dev, verbose=False)
        model.train()
        for b in data:
            tr.train_step(*b)
        torch.cuda.synchronize()
        captured = sum(1 for v in tr._graphs.values() if v["graph"] is not None)
        assert captured == (1 if mode == "1" else 0)
        res[mode] = ({k: v.detach().clone() for k, v in model.state_dict().items()}, tr._stats.clone(), tr._last_out.clone())
    for k, v in res["1"][0].items():
        assert torch.equal(v, res["0"][0][k]), k
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])


def test_lite_hip_graph_is_dropped_when_what_it_froze_is_replaced(dev, monkeypatch):
    """A captured train step holds raw pointers (NAdam moments, the optimiser's entry table, packed classifier weights) and
    scalar launch arguments (betas, eps, weight decay).  ``optimizer.load_state_dict`` after the capture (new moment tensors),
    an edited hyper-parameter and a re-loaded classifier must each invalidate the graph: the run must keep matching an eager
    run (TONAL_GRAPH=0) that goes through the same events, bit for bit."""
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisLite
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    B, C, T = 16, 32, 200
    gen = torch.Generator().manual_seed(23)
    data = [(torch.randn(B, C, T, generator=gen), torch.randn(B, 8, T, generator=gen), torch.randn(B, 8, T, generator=gen),
             10 * torch.randn(B, 80, generator=gen)) for _ in range(20)]
    torch.manual_seed(5)
    other_tone = LogisticRegressionClassifier(8 * T, 4).state_dict()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TONAL_GRAPH", mode)
        torch.manual_seed(0)
        model = SynthesisLite(80, C, T, dropout=0.3)
        torch.manual_seed(1)
        tr = SynthesisTrainer(model, LogisticRegressionClassifier(8 * T, 4), LogisticRegressionClassifier(8 * T, 2), gi.TONE_MAP,
                              device=dev, verbose=False)
        model.train()
        captures = []
        for i, b in enumerate(data):
            if i == 6:                                            # moments move to freshly allocated tensors (a checkpoint
                import copy                                       # that went through torch.save / torch.load)
                tr.optimizer.load_state_dict(copy.deepcopy(tr.optimizer.state_dict()))
            if i == 11:                                           # a launch scalar the capture passed by value
                for gparam in tr.optimizer.param_groups:
                    gparam["weight_decay"] = 0.01
            if i == 16:                                           # classifier weights re-loaded in place (version bump)
                tr.tone_model.load_state_dict({k: v.to(dev) for k, v in other_tone.items()})
            tr.train_step(*b)
            captures.append(sum(1 for v in tr._graphs.values() if v["graph"] is not None))
        torch.cuda.synchronize()
        if mode == "1":
            assert captures[5] == 1 and captures[6] == 0 and captures[10] == 1 and captures[11] == 0 and captures[16] == 0 \
                and captures[-1] == 1, captures
        res[mode] = ({k: v.detach().clone() for k, v in model.state_dict().items()}, tr._stats.clone(), tr._last_out.clone())
    for k, v in res["1"][0].items():
        assert torch.equal(v, res["0"][0][k]), k
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])


def test_small_classifier_heads_run_on_hip_for_inference(dev):
    """LogisticRegressionClassifier / the output layer of ShallowNNClassifier (reference models/simple_classifiers.py:34-60,
    107-134): under no_grad on the GPU the head is tl_linear_rows, with autograd it stays nn.Linear - same numbers."""
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier, ShallowNNClassifier
    torch.manual_seed(3)
    for C_, T_, ncls, B in ((8, 400, 4, 256), (8, 200, 2, 64), (4, 100, 5, 7), (3, 33, 4, 5)):
        m = LogisticRegressionClassifier(C_ * T_, ncls).to(dev)
        x = torch.randn(B, C_, T_, device=dev)
        ref = torch.nn.functional.linear(x.reshape(B, -1).double(), m.linear.weight.double(), m.linear.bias.double()).detach()
        with torch.no_grad():
            out = m(x)
            sub = m(x[1:])                                      # an offset view: rows stay 16-byte aligned or are copied
        assert out.shape == (B, ncls) and out.dtype == torch.float32
        assert float((out.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
        assert torch.equal(sub, out[1:])
        g = m(x)                                                # autograd path: trainable as before
        assert g.requires_grad and float((g.detach().double() - ref).abs().max()) < 1e-4
        g.square().mean().backward()
        assert m.linear.weight.grad is not None
    for act in ("ReLU", "LeakyReLU", "ELU"):                    # hidden layer: NT GEMM + fused (Leaky)ReLU, or + the module
        s = ShallowNNClassifier(8 * 100, 4, activation=act).to(dev)
        x = torch.randn(16, 8, 100, device=dev)
        with torch.no_grad():
            out = s(x)
            ref = s.output(s.activation(s.hidden(x.reshape(16, -1))))
        assert float((out - ref).abs().max()) < 1e-5, act
        assert s(x).requires_grad                               # autograd path untouched
    with pytest.raises(ValueError, match="Expected input dimension"):
        LogisticRegressionClassifier(10, 2).to(dev)(torch.randn(2, 11, device=dev))


def test_stage_step_and_slab_sums_small_entry_points(dev):
    """tl_stage_step (scalars + up to four tensor copies in one launch: 16-byte, 4-byte and byte paths) and tl_sum_slabs2
    (two slab reductions in one launch) straight through the C ABI."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import check, ptr
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(7)
    srcs = [torch.randn(64, 32, 200, device=dev, generator=g),                       # 16-byte units
            torch.randn(1001, device=dev, generator=g)[1:],                          # 4-byte aligned only (offset view)
            torch.randint(0, 255, (1003,), device=dev, dtype=torch.uint8, generator=g)[1:],   # bytes
            torch.randint(0, 9, (64, 80), device=dev, generator=g)]                  # int64
    dsts = [torch.zeros_like(s.contiguous()) if s.is_contiguous() else None for s in srcs]
    assert all(d is not None for d in dsts)
    scal = torch.zeros(4, device=dev)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    S = (C_.c_void_p * 4)(*[s.data_ptr() for s in srcs])
    D = (C_.c_void_p * 4)(*[d.data_ptr() for d in dsts])
    N = (C_.c_int64 * 4)(*[s.numel() * s.element_size() for s in srcs])
    check(lib.tl_stage_step(ptr(scal), ptr(seed), 0.25, -1.5, 3.0, 123456789012345, S, D, N, 4, st), "tl_stage_step")
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    assert scal.tolist() == [0.25, -1.5, 3.0, 0.0] and int(seed.item()) == 123456789012345
    a = torch.randn(17, 300, device=dev, generator=g)
    b = torch.randn(17, 5, device=dev, generator=g)
    ra, rb = torch.empty(300, device=dev), torch.empty(5, device=dev)
    check(lib.tl_sum_slabs2(ptr(a), ptr(ra), 300, ptr(b), ptr(rb), 5, 17, st), "tl_sum_slabs2")
    assert float((ra - a.double().sum(0).float()).abs().max()) < 1e-5 and float((rb - b.double().sum(0).float()).abs().max()) < 1e-5
    check(lib.tl_sum_slabs2(ptr(a), ptr(ra), 300, None, None, 0, 17, st), "tl_sum_slabs2")      # second tensor optional
    assert float((ra - a.double().sum(0).float()).abs().max()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 3, 200, 128, 128, 64), (3, 5, 236, 128, 256, 128), (1, 1, 44, 128, 128, 128),
                                   (7, 3, 100, 256, 128, 64), (6, 8, 400, 512, 512, 512),
                                   # 544 / 272 tiles for the 256 persistent workgroups: the tile walk, the next tile's first two
                                   # stages issued in front of an epilogue and the rolling stage index are only exercised when a
                                   # workgroup processes more than one tile
                                   (8, 32, 400, 512, 512, 512)])
@pytest.mark.parametrize("yprod", ["1", "0"])
def test_f63_stage_kernels_match_direct_kernels(dev, shape, yprod, monkeypatch):
    if yprod == "0" and shape[3] % 256 != 0 and shape[4] % 256 != 0:
        pytest.skip("the producer paths need C_in of stage 2 / stage 3 % 256 == 0: nothing to switch off")
    f63_stage_check(dev, shape, yprod, monkeypatch.setenv, twice=True, ntail=shape[5] >= 128)
    # ---- the C ABI refuses what the kernels do not cover ----
    from decode_tonal_langauge_amd import _lib
    lib = _lib.load()
    p = _lib_nt()
    assert lib.tl_conv3_wino63v_nt(None, None) != 0 and lib.tl_conv3_wino63v_tn(None, None) != 0
    dummy = torch.zeros(64, device=dev)
    for k in ("A", "Bw", "out"):
        setattr(p, k, dummy.data_ptr())
    p.loader, p.J, p.M, p.N, p.K, p.lda, p.ldb, p.Tp, p.A_rows = 2, 3, 12, 32, 16, 16, 16, 12, 128      # K < 40
    assert lib.tl_conv3_wino63v_nt(C_.byref(p), None) != 0 and b"wino63v_nt" in lib.tl_last_error()
    p.K, p.lda, p.ldb, p.Tp = 48, 48, 48, 8                                                            # Tp % 6
    assert lib.tl_conv3_wino63v_nt(C_.byref(p), None) != 0 and b"Tp" in lib.tl_last_error()


def f63_stage_check(dev, shape, yprod, setenv, twice=False, ntail=False):
    """Round 4: the Winograd F(6,3) kernels (default where the stack allows them) against the direct MFMA kernels, stage by
    stage through the C ABI: conv1 writing V1 in hex form (== B^T of the raw rows it stores on request), conv2 forward
    writing V2 (epilogue 5 + fix-up: hexes that take rows from the next half-wave, wave, tile; hexes that end a sequence),
    conv3 forward with the decoupled output row stride, both weight gradients incl. the Vd they write (== B^T of the
    un-pooled gradient), both input gradients and the fused conv1 weight gradient.  Shapes: sequences shorter than a
    half-wave's rows, ragged last tiles, a pooled row count that is odd (400 -> 51 rows per sequence behind conv3's hexes)."""
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    from decode_tonal_langauge_amd._lib import check, ptr
    from tests.wino63_ref import hex_transform, logical, unpool, y_transform
    B, C, T, c1, c2, c3 = shape
    setenv("TONAL_F63_YPROD", yprod)
    defs = [(c1, 3, True), (c2, 3, True), (c3, 3, True), (32, 1, True), (8, 1, False)]
    engs = {}
    for mode in ("0", "6"):
        setenv("TONAL_WINO", mode)
        eng = CnnEngine(80, C, T, 4, 8, 0.0, 0.01, defs, [16, 8])
        eng.store_p1 = True
        if mode == "0":
            eng.fuse_c1 = False
        eng._alloc(B, dev)
        eng._alloc_bwd()
        engs[mode] = eng
    e0, e6 = engs["0"], engs["6"]
    assert e6.wino63 and not e0.wino63 and e6.tp1 % 12 == 0
    S = e6.S
    g = torch.Generator(device=dev).manual_seed(5)
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
            check(eng.lib.tl_conv1_fwd_v6(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), ptr(eng.P[1]), ptr(V1), ptr(eng.bits[1]),
                                          ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "tl_conv1_fwd_v6")
            eng._v_ready[1] = V1
        else:
            check(eng.lib.tl_conv1_fwd(ptr(x), ptr(w1), ptr(prm[names[1] + ".bias"]), ptr(eng.P[1]), ptr(eng.bits[1]),
                                       ptr(eng.sbits[1]), S, T, 3, c1, eng.tp1, eng.tout1, eng.slope, st_), "tl_conv1_fwd")

    def rows(t, tp, n):          # the first n rows of every sequence of a (S * tp, C) tensor
        return t.view(S, tp, -1)[:, :n]

    tin2 = e6.stages[0].tin
    assert torch.equal(rows(e6.P[1], e6.tp1, tin2), rows(e0.P[1], e0.tp1, tin2))
    assert torch.equal(rows(e6.bits[1], e6.tp1, tin2), rows(e0.bits[1], e0.tp1, tin2))
    Vref = hex_transform(e6.P[1], S, e6.tp1)
    close = lambda a, b: float((a.double() - b).abs().max()) <= 1e-6 * float(b.abs().max())     # (max norm: the transform cancels)
    assert close(logical(e6.V[1])[:Vref.shape[0]], Vref)
    assert float(e6.V[1][Vref.shape[0]:].abs().max()) == 0.0
    for si in (2, 3):
        for eng in (e0, e6):
            eng.stage_forward(eng.stages[si - 2], prm[names[si] + ".weight"], prm[names[si] + ".bias"])
        s0, s6 = e0.stages[si - 2], e6.stages[si - 2]
        nv = s6.tout
        assert rel(rows(e6.P[si], s6.tp_out, nv).cpu().numpy(), rows(e0.P[si], s0.tp_out, nv).cpu().numpy()) < 2e-5, si
        assert float(rows(e6.P[si], s6.tp_out, s6.tp_out)[:, nv:].abs().max()) == 0.0 if s6.tp_out > nv else True
        for b6, b0 in ((e6.bits, e0.bits), (e6.sbits, e0.sbits)):
            flips = rows(b6[si], s6.tp_out, nv) ^ rows(b0[si], s0.tp_out, nv)
            # arg-max / sign ties and near-ties only: 6e-7 of the bits at the timed batch (profiles/parity_observed.json)
            assert int((flips != 0).sum()) <= max(4, int(2e-6 * S * nv * s6.cout)), si
        if twice:
            # a second launch into the same buffers: every partial-tile store lands (or is dropped) the same way again
            kept = (e6.P[si].clone(), e6.bits[si].clone(), e6.sbits[si].clone())
            e6.P[si].fill_(float("nan")); e6.bits[si].fill_(0x5a5a5a5a); e6.sbits[si].fill_(0x5a5a5a5a)
            e6.stage_forward(s6, prm[names[si] + ".weight"], prm[names[si] + ".bias"])
            for a_, b_ in zip(kept, (e6.P[si], e6.bits[si], e6.sbits[si])):
                assert torch.equal(rows(a_, s6.tp_out, nv), rows(b_, s6.tp_out, nv)), si
                b_.copy_(a_)
        if si == 2:
            V2ref = hex_transform(e6.P[2], S, s6.tp_out)
            V2 = e6._v_ready[2]
            assert close(logical(V2)[:V2ref.shape[0]], V2ref)
            keep = V2.clone()
            e6.stage_forward(s6, prm[names[2] + ".weight"], prm[names[2] + ".bias"])      # (raw hexes of the fix-up pass rewritten)
            assert torch.equal(e6._v_ready[2], keep)
        if si == 3 and ntail and s6.cout % 64 == 0:
            # column tail: the same launch with N = C_out - 32 (the last 64-column tile of every row tile is half empty) must
            # write the first N columns exactly as before and nothing to their right (rows, arg-max and sign words)
            from decode_tonal_langauge_amd._lib import EPI_POOL, LOAD_V
            nn_ = s6.cout - 32
            wt = e6._pack_wino63(prm[names[3] + ".weight"][:nn_].contiguous(), True)
            V = e6._v_ready[2]
            outp = torch.full_like(e6.P[3], -777.0)
            ob, osg = torch.full_like(e6.bits[3], 0x13572468), torch.full_like(e6.sbits[3], 0x13572468)
            e6._nt(fn="tl_conv3_wino63v_nt", A=ptr(V), A_rows=V.shape[0], lda=V.shape[2], loader=LOAD_V, Bw=ptr(wt),
                   bias=ptr(prm[names[3] + ".bias"]), out=ptr(outp), M=S * s6.tp_in, N=nn_, K=s6.cin, ldb=s6.cin, ldo=s6.cout, J=3,
                   row_shift=0, Tp=s6.tp_in, slope=e6.slope, obits=ptr(ob), osign=ptr(osg), ld_obits=s6.cout // 32,
                   Tvalid=2 * s6.tout, epilogue=EPI_POOL, out_tp=s6.tp_out)
            assert torch.equal(rows(outp, s6.tp_out, nv)[..., :nn_], rows(e6.P[3], s6.tp_out, nv)[..., :nn_])
            assert bool((outp[:, nn_:] == -777.0).all())
            nw = nn_ // 32
            assert torch.equal(rows(ob, s6.tp_out, nv)[..., :nw], rows(e6.bits[3], s6.tp_out, nv)[..., :nw])
            assert torch.equal(rows(osg, s6.tp_out, nv)[..., :nw], rows(e6.sbits[3], s6.tp_out, nv)[..., :nw])
            assert bool((ob[:, nw:] == 0x13572468).all()) and bool((osg[:, nw:] == 0x13572468).all())
    # ---- backward from a random G3; the direct engine un-pools with the F(6,3) engine's bits (ties may differ) ----
    s0, s6 = e0.stages[1], e6.stages[1]
    G3 = torch.randn(S, s6.tp_out, c3, device=dev, generator=g)
    G3[:, s6.tout:] = 0
    e6.G[3].copy_(G3.reshape(-1, c3))
    e0.G[3].view(S, s0.tp_out, c3).zero_()
    e0.G[3].view(S, s0.tp_out, c3)[:, :s6.tout] = G3[:, :s6.tout]
    for idx in (2, 3):
        a, b = e0.stages[idx - 2], e6.stages[idx - 2]
        e0.bits[idx].view(S, a.tp_out, -1)[:, :b.tout] = e6.bits[idx].view(S, b.tp_out, -1)[:, :b.tout]
        e0.sbits[idx].view(S, a.tp_out, -1)[:, :b.tout] = e6.sbits[idx].view(S, b.tp_out, -1)[:, :b.tout]
    for si in (3, 2):
        res = {}
        for key, eng in (("0", e0), ("6", e6)):
            st = eng.stages[si - 2]
            w = prm[names[si] + ".weight"]
            gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
            eng.stage_wgrad(st, gw, gb)
            res[key] = (gw, gb, eng.stage_dgrad(st, w))
            if twice and key == "6":
                # the whole backward of the stage a second time into the same buffers: bit-identical outputs
                snap = {k: v.clone() for k, v in (("gw", gw), ("gb", gb), ("part", res[key][2])) if v is not None}
                for name_, store in (("Vd", eng.Vd), ("Yt", eng.Yt), ("G", eng.G)):
                    for idx_ in (si, si - 1):
                        if idx_ in store:
                            snap[f"{name_}{idx_}"] = store[idx_].clone()
                eng._y_ready[si] = eng._vd_ready[si] = -1
                if si == 3 and eng.f63_yprod3:
                    pass                                   # (the stand-alone producer re-runs inside stage_wgrad)
                elif eng.f63_yprod and si == 2:
                    eng._y_ready[2] = eng._vd_ready[2] = eng.generation          # Y2 / Vd2 from stage 3's epilogue are still there
                gw2, gb2 = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
                eng.stage_wgrad(st, gw2, gb2)
                part2 = eng.stage_dgrad(st, w)
                again = {"gw": gw2, "gb": gb2, "part": part2}
                for name_, store in (("Vd", eng.Vd), ("Yt", eng.Yt), ("G", eng.G)):
                    for idx_ in (si, si - 1):
                        if idx_ in store:
                            again[f"{name_}{idx_}"] = store[idx_]
                for k_, v_ in snap.items():
                    assert torch.equal(v_, again[k_]), (si, k_)
        s0, s6 = e0.stages[si - 2], e6.stages[si - 2]
        assert rel_l2(res["6"][0].cpu().numpy(), res["0"][0].cpu().numpy()) < 1e-5, si
        assert rel_l2(res["6"][1].cpu().numpy(), res["0"][1].cpu().numpy()) < 1e-5, si
        if si in e6.G:
            dz = unpool(e6.G[si], e6.bits[si], S, s6.tp_out, 2 * s6.tout, s6.cout)
            if 2 * s6.tp_out < s6.tp_in:
                dz = torch.nn.functional.pad(dz, (0, 0, 0, s6.tp_in - 2 * s6.tp_out))
            Vdref = hex_transform(dz[:, :s6.tp_in].reshape(-1, s6.cout), S, s6.tp_in, shift=-2)
            assert close(logical(e6.Vd[si])[:Vdref.shape[0]], Vdref), si
        if si == 3 and e6.f63_yprod3:
            # stage 3's own operands from the stand-alone producer: Y3 = A dz3 of the same un-pooled gradient
            assert c2 % 256 == 0
            Y3ref = y_transform(dz[:, :s6.tp_in].reshape(-1, s6.cout), S, s6.tp_in)
            assert close(logical(e6.Yt[3])[:Y3ref.shape[0]], Y3ref) and float(e6.Yt[3][Y3ref.shape[0]:].abs().max()) == 0.0
        if si == 3 and e6.f63_yprod:
            # the input gradient of stage 3 wrote the operands of stage 2's backward - Y2 = A dz2 and Vd2 = B^T dz2 - instead
            # of G2 (epilogue 6 + tl_wino63_vd_fixup): against the direct engine's G2, un-pooled with the same bits
            assert 2 not in e6.G and c1 % 256 == 0
            nin, b2 = s6.tin, e6.stages[0]
            g2 = torch.zeros(S, s6.tp_in, s6.cin, device=dev)
            g2[:, :nin] = e0.G[2].view(S, s0.tp_in, -1)[:, :nin]
            dz2 = unpool(g2.reshape(-1, s6.cin), e6.bits[2], S, s6.tp_in, 2 * b2.tout, s6.cin).reshape(-1, s6.cin)
            Yref, Vd2ref = y_transform(dz2, S, b2.tp_in), hex_transform(dz2, S, b2.tp_in, shift=-2)
            near = lambda a, b: float((a.double() - b).abs().max()) <= 1e-5 * float(b.abs().max())   # (two fp32 gradients apart)
            assert near(logical(e6.Yt[2])[:Yref.shape[0]], Yref) and near(logical(e6.Vd[2])[:Yref.shape[0]], Vd2ref)
            assert float(e6.Yt[2][Yref.shape[0]:].abs().max()) == 0.0 and float(e6.Vd[2][Yref.shape[0]:].abs().max()) == 0.0
        elif si == 3:
            nin = s6.tin
            assert rel_l2(rows(e6.G[2], s6.tp_in, nin).cpu().numpy(), rows(e0.G[2], s0.tp_in, nin).cpu().numpy()) < 1e-5
            e6.G[2].view(S, s6.tp_in, -1).zero_()
            e6.G[2].view(S, s6.tp_in, -1)[:, :nin] = e0.G[2].view(S, s0.tp_in, -1)[:, :nin]
        else:
            nblk = int(min(2048, S))
            p0 = torch.empty(nblk, 4 * c1, device=dev)
            check(e0.lib.tl_conv1_wgrad(ptr(e0._x), ptr(e0.G[1]), ptr(e0.bits[1]), ptr(p0), nblk, S, T, 3, c1, e0.tp1, e0.tout1, st_),
                  "tl_conv1_wgrad")
            assert rel_l2(res["6"][2].sum(0).cpu().numpy(), p0.sum(0).cpu().numpy()) < 1e-5


def _lib_nt():
    from decode_tonal_langauge_amd import _lib
    return _lib.NtParams()


@pytest.mark.parametrize("kt", [1, 2, 3, 4, 5, 6, 7, 8])
def test_conv1_kernels_for_every_tap_count(dev, kt):
    """The first stage (reference models/synthesis_models.py:85-90, deep_classifiers.py:230-247: Conv2d(1, C1, (k, 1)) +
    LeakyReLU + MaxPool (2, 1)) for every tap count the entry points take (the tap count is a template parameter of the
    kernels): ``tl_conv1_fwd`` (rows, both thread mappings) against a float64 restatement, ``tl_conv1_fwd_v`` (quads) and
    ``tl_conv1_fwd_v6`` (hexes, pair layout) against the transforms of those rows."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import check, ptr
    from tests import wino63_ref as w6
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(100 + kt)
    S, T, slope = 5, 131, 0.01
    tout = (T - kt + 1) // 2
    tp = (tout + 11) // 12 * 12                                  # whole quads and whole hexes
    bt43 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                         [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64, device=dev)
    for c1 in (128, 192):                                       # 192: the one-channel-per-thread kernel
        x = torch.randn(S, T, device=dev, generator=g)
        w = torch.randn(c1, kt, device=dev, generator=g) * 0.5
        b = torch.randn(c1, device=dev, generator=g) * 0.1
        xd = x.double()
        z = sum(w.double()[None, None, :, j] * xd[:, j:j + 2 * tout, None] for j in range(kt)) + b.double()
        y = torch.where(z > 0, z, slope * z).view(S, tout, 2, c1)
        odd = y[:, :, 1] > y[:, :, 0]
        pooled = torch.where(odd, y[:, :, 1], y[:, :, 0])
        ref = torch.zeros(S, tp, c1, dtype=torch.float64, device=dev)
        ref[:, :tout] = pooled
        P = torch.full((S * tp, c1), float("nan"), device=dev)
        bits = torch.full((S * tp, c1 // 32), -1, dtype=torch.int32, device=dev)
        sign = torch.full_like(bits, -1)
        check(lib.tl_conv1_fwd(ptr(x), ptr(w), ptr(b), ptr(P), ptr(bits), ptr(sign), S, T, kt, c1, tp, tout, slope, st),
              "tl_conv1_fwd")
        torch.cuda.synchronize()
        assert float((P.double().view(S, tp, c1) - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
        sh = torch.arange(32, device=dev, dtype=torch.int32)
        got_odd = ((bits.view(S, tp, c1 // 32, 1) >> sh) & 1).reshape(S, tp, c1).bool()
        got_pos = ((sign.view(S, tp, c1 // 32, 1) >> sh) & 1).reshape(S, tp, c1).bool()
        # a bit may differ from the float64 restatement only where the two candidates (or the value and zero) tie in fp32
        gap = (y[:, :, 1] - y[:, :, 0]).abs()
        assert bool(((got_odd[:, :tout] == odd) | (gap < 1e-5)).all())
        assert bool(((got_pos[:, :tout] == (pooled > 0)) | (pooled.abs() < 1e-5)).all())
        assert not bool(got_odd[:, tout:].any()) and not bool(got_pos[:, tout:].any())
        if c1 != 128:
            continue
        # quads: V[quad][6][C1] = B^T (rows 4q .. 4q+5), raw rows, bits and sign words identical to the row kernel's
        P4, b4, s4 = torch.full_like(P, float("nan")), torch.full_like(bits, -1), torch.full_like(bits, -1)
        V4 = torch.full((S * tp // 4, 6, c1), float("nan"), device=dev)
        check(lib.tl_conv1_fwd_v(ptr(x), ptr(w), ptr(b), ptr(P4), ptr(V4), ptr(b4), ptr(s4), S, T, kt, c1, tp, tout, slope, st),
              "tl_conv1_fwd_v")
        rows = torch.nn.functional.pad(P.double().view(S, tp, c1), (0, 0, 0, 4))
        idx = torch.arange(tp // 4, device=dev)[:, None] * 4 + torch.arange(6, device=dev)[None, :]
        v4ref = torch.einsum("jk,sqkc->sqjc", bt43, rows[:, idx, :]).reshape(S * tp // 4, 6, c1)
        torch.cuda.synchronize()
        assert torch.equal(P4, P) and torch.equal(b4, bits) and torch.equal(s4, sign)
        assert float((V4.double() - v4ref).abs().max()) < 2e-6 * max(1.0, float(v4ref.abs().max()))
        # hexes, pair layout
        P6, b6, s6 = torch.full_like(P, float("nan")), torch.full_like(bits, -1), torch.full_like(bits, -1)
        nh = S * tp // 6
        V6 = torch.zeros((nh + 1) // 2 * 2, 8, c1, device=dev)
        check(lib.tl_conv1_fwd_v6(ptr(x), ptr(w), ptr(b), ptr(P6), ptr(V6), ptr(b6), ptr(s6), S, T, kt, c1, tp, tout, slope, st),
              "tl_conv1_fwd_v6")
        torch.cuda.synchronize()
        assert torch.equal(P6, P) and torch.equal(b6, bits) and torch.equal(s6, sign)
        v6ref = w6.hex_transform(P, S, tp)
        assert float((w6.logical(V6)[:nh].double() - v6ref).abs().max()) < 2e-6 * max(1.0, float(v6ref.abs().max()))


def test_permute_reduce_forms(dev):
    """``tl_permute_reduce`` (the per-step weight packs and split-K slab reductions of the engine; reference side: the layouts
    torch keeps its Conv2d / Linear weights in, models/synthesis_models.py:86-135): the element-per-thread kernel, the slab-parallel
    one and the tiled transpose (innermost destination dimension strided in the source, another one contiguous) against torch
    indexing, with limits that cut every dimension."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import check, ptr
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(77)

    def run(src, dims, strides, lims=None, nz=1, zs=0, bias=None):
        d = (C_.c_int64 * 4)(*dims)
        s = (C_.c_int64 * 4)(*strides)
        l = (C_.c_int64 * 4)(*(lims if lims is not None else dims))
        out = torch.full(tuple(dims), float("nan"), device=dev)
        check(lib.tl_permute_reduce(ptr(src), ptr(out), d, s, l, nz, zs, ptr(bias) if bias is not None else None, st),
              "tl_permute_reduce")
        torch.cuda.synchronize()
        return out

    def ref(src, dims, strides, lims=None, nz=1, zs=0, bias=None):
        lims = lims if lims is not None else dims
        idx = [torch.arange(n, device=dev) for n in dims]
        off = sum(i.view([-1 if k == j else 1 for k in range(4)]) * strides[j] for j, i in enumerate(idx))
        ok = torch.ones(tuple(dims), dtype=torch.bool, device=dev)
        for j, i in enumerate(idx):
            ok = ok & (i.view([-1 if k == j else 1 for k in range(4)]) < lims[j])
        flat = src.reshape(-1).double()
        acc = sum(flat[(off + z * zs).clamp(max=flat.numel() - 1)] for z in range(nz))
        if bias is not None:
            acc = acc + bias.double().view(1, 1, 1, -1)
        return torch.where(ok, acc, torch.zeros_like(acc)).float()

    # the Linear layer's packs at a reduced size: (C, tp, ld, out) from torch's (out, Cc, lat, C) and back
    Cn, tp, lat, Cc, ld, nout, ldd = 40, 7, 6, 12, 16, 10, 12
    w = torch.randn(nout, Cc * lat * Cn, device=dev, generator=g)
    cases = [
        (w, (Cn, tp, ld, ldd), (1, Cn, lat * Cn, Cc * lat * Cn), (Cn, lat, Cc, nout)),            # dgrad pack: K = 0 contiguous
        (torch.randn(ldd, Cn * tp * ld, device=dev, generator=g), (nout, Cc, lat, Cn), (Cn * tp * ld, 1, ld, tp * ld), None),   # K = 1... short
        (torch.randn(5, 3, 70, 50, device=dev, generator=g), (5, 3, 50, 70), (3 * 70 * 50, 70 * 50, 1, 50), (5, 2, 45, 70)),     # K = 2, ragged tiles
        (torch.randn(64, 3, 2, 33, device=dev, generator=g), (33, 3, 2, 64), (1, 2 * 33, 33, 3 * 2 * 33), None),                  # K = 0, 33 x 64
        (torch.randn(9, 11, 13, 17, device=dev, generator=g), (9, 13, 11, 17), (11 * 13 * 17, 17, 13 * 17, 1), None),             # contiguous innermost: element kernel
    ]
    for src, dims, strides, lims in cases:
        got, exp = run(src, dims, strides, lims), ref(src, dims, strides, lims)
        assert torch.equal(got, exp), (dims, strides)
    # slab reductions (nz > 1) and the bias of the last dimension
    slabs = torch.randn(70, 6, 8, 4, 5, device=dev, generator=g)
    b = torch.randn(5, device=dev, generator=g)
    got = run(slabs, (6, 8, 4, 5), (8 * 4 * 5, 4 * 5, 5, 1), nz=70, zs=6 * 8 * 4 * 5, bias=b)
    assert float((got - ref(slabs, (6, 8, 4, 5), (8 * 4 * 5, 4 * 5, 5, 1), nz=70, zs=6 * 8 * 4 * 5, bias=b)).abs().max()) < 1e-4
    big = torch.randn(3, 40, 30, 20, 16, device=dev, generator=g)
    got = run(big, (40, 20, 30, 16), (30 * 20 * 16, 16, 20 * 16, 1), nz=3, zs=40 * 30 * 20 * 16)
    assert float((got - ref(big, (40, 20, 30, 16), (30 * 20 * 16, 16, 20 * 16, 1), nz=3, zs=40 * 30 * 20 * 16)).abs().max()) < 1e-5


@pytest.mark.parametrize("shape", [(3000, 128, 128, 25, 24), (1037, 72, 64, 12, 11), (5000, 256, 200, 1, 1)])
def test_one_tap_weight_gradient_carries_the_bias_gradient(dev, shape):
    """``tl_gemm_tn_window`` on its one-tap direct kernel: slab[z] = A^T B over split z and, with ``colsum``, the column sums of
    B over the same rows - rows whose time index (row % Tp) is past Tvalid excluded from both (the weight and bias gradients of
    a 1x1 convolution, reference models/synthesis_models.py:103-131, from one pass over the output gradient)."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import TnParams, LOAD_DIRECT, check, ptr
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    K, M, N, Tp, Tv = shape
    g = torch.Generator(device=dev).manual_seed(K + M)
    A = torch.randn(K, M + 4, device=dev, generator=g)
    B = torch.randn(K, N + 8, device=dev, generator=g)
    valid = (torch.arange(K, device=dev) % Tp) < Tv
    Bd = torch.where(valid[:, None], B[:, :N].double(), torch.zeros((), dtype=torch.float64, device=dev))
    ref_w, ref_b = A[:, :M].double().t() @ Bd, Bd.sum(0)
    for sk in (1, 3, 16):
        slab = torch.full((sk, M, N), float("nan"), device=dev)
        cs = torch.full((sk, N), float("nan"), device=dev)
        p = TnParams()
        p.J, p.Tp, p.Tvalid, p.splitk, p.slab_stride = 1, Tp, Tv, sk, M * N
        p.A, p.B, p.slab, p.colsum = ptr(A), ptr(B), ptr(slab), ptr(cs)
        p.Krows, p.A_rows, p.B_rows, p.Mdim, p.Ndim, p.lda, p.ldb, p.ldc, p.loader = K, K, K, M, N, M + 4, N + 8, N, LOAD_DIRECT
        check(lib.tl_gemm_tn_window(C_.byref(p), st), "tl_gemm_tn_window")
        torch.cuda.synchronize()
        assert float((slab.double().sum(0) - ref_w).abs().max()) < 2e-6 * float(ref_w.abs().max()) * max(1.0, K / 1000)
        assert float((cs.double().sum(0) - ref_b).abs().max()) < 2e-6 * float(ref_b.abs().max() + K ** 0.5), sk


@pytest.mark.parametrize("nblk", [40, 300, 1111])
def test_first_stage_gradient_partials_reduce_to_torch_layout(dev, nblk):
    """The per-tile partial sums of conv1's weight / bias gradient (``tl_conv1_wgrad`` / the fused epilogue of conv2's input
    gradient; reference: the gradient of ``ecog_conv_block[0]``, models/synthesis_models.py:87) summed and permuted into
    torch's layouts - the slab-parallel permute (few tiles) and the column-sum form (many tiles) against torch."""
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    model = SynthesisModelCNN(80, 8, 200, dropout=0.0).to(dev)
    eng = model._engine
    g = torch.Generator(device=dev).manual_seed(nblk)
    part = torch.randn(nblk, (eng.k1 + 1) * eng.c1, device=dev, generator=g)
    gw = torch.full((eng.c1, 1, eng.k1, 1), float("nan"), device=dev)
    gb = torch.full((eng.c1,), float("nan"), device=dev)
    eng._reduce_c1_partials(part, gw, gb)
    torch.cuda.synchronize()
    tot = part.double().sum(0)
    ref_w = tot[:eng.k1 * eng.c1].view(eng.k1, eng.c1).t()
    assert float((gw.view(eng.c1, eng.k1).double() - ref_w).abs().max()) < 1e-5 * float(ref_w.abs().max())
    assert float((gb.double() - tot[eng.k1 * eng.c1:]).abs().max()) < 1e-5 * float(tot.abs().max())


@pytest.mark.parametrize("shape", [
    # (sequences, conv rows per sequence of the 3-tap stage below (hexes of 6), its valid pooled rows, rows per sequence of its
    #  bit arrays, its channels = N, channels of the one-tap stage = K)
    (256, 102, 48, 50, 512, 256),          # the timed geometry per 2 windows: 17 hexes per sequence (odd: hexes here straddle sequences)
    (5, 102, 48, 50, 128, 64),             # odd sequence count: the last hex is half empty (M % 6 == 3)
    (3, 12, 5, 6, 64, 64),                 # two hexes per sequence, an odd number of valid pooled rows
    (37, 30, 14, 15, 96, 96),              # K / N that are not powers of two; bit arrays as wide as the hex geometry
    (700, 54, 26, 27, 64, 64),             # more than one row tile per workgroup column: the tile walk
])
def test_one_tap_input_gradient_on_the_f63_nt_kernel_writes_y_and_vd(dev, shape):
    one_tap_gy_check(dev, shape, refusals=True)


def one_tap_gy_check(dev, shape, refusals=False):
    """Round 5: conv4's input gradient (reference models/synthesis_models.py:99-101, backward of loss.backward(),
    synthesis_trainer.py:226) as ``tl_wino63_unpool_rows6`` + ``tl_wino63_weights1`` + ``tl_conv1_wino63v_dgrad_nt`` +
    ``tl_wino63_vd_fixup``: Y3 = A dz3 and Vd3 = B^T dz3 of the 3-tap stage below, against a float64 restatement
    (un-pool the one-tap stage's pooled gradient, multiply by W, LeakyReLU' from the sign words, un-pool by the stage's own
    arg-max words, transform).  Pad hexes stay zero; a second launch into the same buffers is bit-identical."""
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import EPI_GY, LOAD_V, check, ptr
    from tests.wino63_ref import hex_transform, logical, unpool, y_transform
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    S, tp3, tout3, gtp3, c3, c4 = shape
    slope = 0.01
    tpg = tp3 // 2                                     # gradient rows per sequence in the hex geometry (3 per hex)
    tin4 = tout3                                       # the one-tap stage: rows in = conv rows
    tout4, gtp4 = tin4 // 2, (gtp3 + 1) // 2
    g = torch.Generator(device=dev).manual_seed(77 + S)
    G4 = torch.randn(S * gtp4, c4, device=dev, generator=g)
    G4.view(S, gtp4, c4)[:, tout4:] = 0
    rnd = lambda *s: torch.randint(-2 ** 31, 2 ** 31 - 1, s, device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    bits4, bits3, sbits3 = rnd(S * gtp4, c4 // 32), rnd(S * gtp3, c3 // 32), rnd(S * gtp3, c3 // 32)
    W = torch.randn(c4, c3, device=dev, generator=g) / c4 ** 0.5
    # ---- float64 restatement
    dz4 = unpool(G4, bits4, S, gtp4, 2 * tout4, c4)[:, :gtp3]                         # (S, gtp3, c4): rows of the stage's input
    if dz4.shape[1] < gtp3:
        dz4 = torch.nn.functional.pad(dz4, (0, 0, 0, gtp3 - dz4.shape[1]))
    g3 = dz4 @ W.double()                                                              # (S, gtp3, c3)
    sh = torch.arange(32, device=dev, dtype=torch.int32)
    pos = ((sbits3.view(S, gtp3, c3 // 32, 1) >> sh) & 1).reshape(S, gtp3, c3).bool()
    g3 = g3 * torch.where(pos, 1.0, slope)
    dz3 = unpool(g3.reshape(-1, c3), bits3, S, gtp3, 2 * tout3, c3)                    # (S, 2 gtp3, c3)
    if 2 * gtp3 < tp3:
        dz3 = torch.nn.functional.pad(dz3, (0, 0, 0, tp3 - 2 * gtp3))
    dz3 = dz3[:, :tp3].reshape(-1, c3)
    Yref, Vdref = y_transform(dz3, S, tp3), hex_transform(dz3, S, tp3, shift=-2)
    # ---- the HIP path
    rows = S * tpg
    nh = -(-rows // 6)
    nh_pad = (nh + 127) // 128 * 128
    A = torch.zeros(nh_pad, 8, c4, device=dev)
    nh3 = S * (tp3 // 6)
    nh3_pad = (nh3 + 24 + 127) // 128 * 128
    tile_rows = lib.tl_wino63_nt_tile_rows()
    ntm = -(-rows // tile_rows)
    taps = torch.empty(c4 // 8, 8, c3, 8, device=dev)
    outs = []
    for _ in range(2):
        Y, Vd = torch.zeros(nh3_pad, 8, c3, device=dev), torch.zeros(nh3_pad, 8, c3, device=dev)
        halo = torch.zeros(ntm, 2, c3, device=dev)
        check(lib.tl_wino63_unpool_rows6(ptr(G4), ptr(bits4), ptr(A), rows, G4.shape[0], tpg, gtp4, 2 * tout4, c4, c4, c4 // 32, c4, 0,
                                         st), "tl_wino63_unpool_rows6")
        check(lib.tl_wino63_weights1(ptr(W), ptr(taps), c4, c3, c4, st), "tl_wino63_weights1")
        p = _lib.NtParams()
        p.splitk, p.bm = 1, 128
        for k, v in dict(A=ptr(A), A_rows=nh_pad, lda=c4, loader=LOAD_V, Bw=ptr(taps), M=rows, N=c3, K=c4, ldb=c4, ldo=c3, J=1,
                         row_shift=0, Tp=tpg, slope=slope, auxbits=ptr(sbits3), ld_auxbits=c3 // 32, abits=ptr(bits3),
                         ld_abits=c3 // 32, out_tp=gtp3, Tvalid_in=2 * tout3, epilogue=EPI_GY, vout=ptr(Y), vout2=ptr(Vd),
                         vhalo=ptr(halo), vout_quads=nh3_pad, ld_vout=c3).items():
            setattr(p, k, v)
        check(lib.tl_conv1_wino63v_dgrad_nt(C_.byref(p), st), "tl_conv1_wino63v_dgrad_nt")
        check(lib.tl_wino63_vd_fixup(ptr(Vd), ptr(halo), rows // 3, ntm, tp3 // 6, c3, c3, st), "tl_wino63_vd_fixup")
        torch.cuda.synchronize()
        outs.append((Y, Vd))
    (Y, Vd), (Y2, Vd2) = outs
    assert torch.equal(Y, Y2) and torch.equal(Vd, Vd2)
    # slots 0..5 of A are the un-pooled rows in hex order, slots 6, 7 zero
    Al = logical(A)[:nh]
    want = torch.nn.functional.pad(dz4[:, :tpg] if dz4.shape[1] >= tpg else torch.nn.functional.pad(dz4, (0, 0, 0, tpg - dz4.shape[1])),
                                   (0, 0, 0, 0)).reshape(-1, c4)
    want = torch.nn.functional.pad(want, (0, 0, 0, 6 * nh - want.shape[0])).view(nh, 6, c4)
    assert torch.equal(Al[:, :6].double(), want) and float(Al[:, 6:].abs().max()) == 0.0
    near = lambda a, b: float((a.double() - b).abs().max()) <= 1e-5 * float(b.abs().max())
    assert near(logical(Y)[:nh3], Yref), float((logical(Y)[:nh3].double() - Yref).abs().max())
    assert near(logical(Vd)[:nh3], Vdref), float((logical(Vd)[:nh3].double() - Vdref).abs().max())
    # (an odd hex count ends inside a pair of the pair layout: slice the logical view, not the memory order)
    assert float(logical(Y)[nh3:].abs().max()) == 0.0 and float(logical(Vd)[nh3:].abs().max()) == 0.0
    if not refusals:
        return
    # ---- what the entry refuses
    p.K = 32
    assert lib.tl_conv1_wino63v_dgrad_nt(C_.byref(p), st) != 0 and b"K" in lib.tl_last_error()
    p.K, p.Tp = c4, tpg + 1
    assert lib.tl_conv1_wino63v_dgrad_nt(C_.byref(p), st) != 0 and b"Tp" in lib.tl_last_error()
    p.Tp, p.epilogue = tpg, 6
    assert lib.tl_conv1_wino63v_dgrad_nt(C_.byref(p), st) != 0
