"""GPU tests of the callers either side of the kernels:

* golden G12: the deep classifiers' HIP inference path against the REFERENCE's eval forward
  (models/deep_classifiers.py:17-155, 158-343) for seeded weights - not against this package's own graph;
* golden G13: the step dispatcher (reference preprocess/preprocessor.py:39-70) over four steps on one
  shared, mutated Namespace;
* BASELINE config C5 assembled end to end at reduced width / batch: raw ECoG -> Hilbert high-gamma ->
  400-sample epochs -> CNNClassifier / CNNRNNClassifier labels -> SynthesisModelCNN train step, against
  the CPU oracle chain on the same inputs.
"""
import os
from argparse import Namespace
from copy import deepcopy

import numpy as np
import pytest
import torch

from tests import golden_inputs as gi
from tests.test_gpu_parity import rel, rel_l2

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PKG = "decode_tonal_langauge_amd"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def test_deep_classifiers_hip_forward_matches_reference_golden(dev):
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier
    g = np.load(os.path.join(GOLD, "g12_deep_classifiers.npz"))
    for i in range(2):
        Cn, T, ncls, B, seed = (int(v) for v in g[f"cnn{i}.cfg"])
        torch.manual_seed(seed)
        clf = CNNClassifier(input_channels=Cn, input_length=T, n_classes=ncls).eval()
        x = torch.randn(B, Cn, T)
        assert list(clf.state_dict().keys()) == list(g[f"cnn{i}.keys"])
        assert clf.get_nparams() == int(g[f"cnn{i}.nparams"]) and clf.latent_length == int(g[f"cnn{i}.latent"])
        clf.to(dev)
        with torch.no_grad():
            out = clf(x.to(dev))
        assert clf._hip is not None, "HIP path was not taken"
        assert out.shape == g[f"cnn{i}.out"].shape
        assert float(np.abs(out.cpu().numpy() - g[f"cnn{i}.out"]).max()) < 2e-5
        # feature map of the conv trunk (reference layout (B, 256, t, C)) from the engine buffer
        eng = clf._hip
        last = eng.stages[-1]
        feat = eng.P[last.idx].view(B, Cn, eng.tp_last, eng.ld_last)[:, :, :eng.lat, :eng.c_last].permute(0, 3, 2, 1)
        assert rel(feat.cpu().numpy(), g[f"cnn{i}.feat"]) < 1e-4
    for i in range(2):
        Cn, T, ncls, B, ld, seed = (int(v) for v in g[f"cnnrnn{i}.cfg"])
        torch.manual_seed(seed)
        clf = CNNRNNClassifier(input_channels=Cn, input_length=T, n_classes=ncls, lstm_dim=ld).eval()
        x = torch.randn(B, Cn, T)
        assert list(clf.state_dict().keys()) == list(g[f"cnnrnn{i}.keys"])
        assert clf.get_nparams() == int(g[f"cnnrnn{i}.nparams"])
        clf.to(dev)
        with torch.no_grad():
            out = clf(x.to(dev))
        assert clf._hip is not None, "HIP path was not taken"
        assert float(np.abs(out.cpu().numpy() - g[f"cnnrnn{i}.out"]).max()) < 5e-5
        h1 = getattr(clf._hip, "last_h1", None)
        if h1 is not None:                                    # first LSTM's final state from the HIP kernel
            assert rel(h1.cpu().numpy(), g[f"cnnrnn{i}.h1"]) < 1e-4


@pytest.mark.parametrize("B,T,D,H", [(5, 7, 12, 20), (64, 9, 33, 136), (70, 4, 8, 800), (33, 1, 16, 24)])
def test_lstm_inference_paths_against_torch_lstm(dev, B, T, D, H, monkeypatch):
    """The HIP inference form of nn.LSTM(batch_first=True)(x)[0][:, -1] (reference models/deep_classifiers.py:294-296,
    316-318) - one fused launch per step: 32-row tiles, hidden widths padded to 8, ragged row tiles, T = 1 and odd / even T
    for the ping-pong state - against torch on the CPU."""
    from decode_tonal_langauge_amd._classifier_engine import LstmInferEngine
    torch.manual_seed(B * 1000 + H)
    lstm = torch.nn.LSTM(D, H, batch_first=True)
    x = torch.randn(B, T, D)
    with torch.no_grad():
        ref = lstm(x)[0][:, -1, :].double()
    w = [getattr(lstm, n).detach().to(dev) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    eng = LstmInferEngine(D, H)
    got = eng.last_hidden(x.to(dev), *w).cpu().double()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("nseq,Tp,tvalid,cin,cout,taps", [(3, 24, 21, 64, 96, 7), (2, 516, 516, 32, 64, 7), (700, 198, 197, 64, 64, 7),
                                                          (5, 12, 12, 16, 32, 9), (4, 18, 13, 48, 160, 8)])
def test_conv7_on_the_f63_nt_kernel_against_float64(dev, nseq, Tp, tvalid, cin, cout, taps):
    """Round 5: tl_wino63_xform2 + tl_wino63_weights7 + tl_conv7_wino63v_nt (a 7..9-tap convolution + LeakyReLU as three
    F(6,3) segments summed in the transform domain by ONE launch of the synthesis stack's V-form NT kernel; the third segment
    re-reads V0 one hex on) against a float64 sliding-window convolution.  Rows of a sequence from ``tvalid`` on enter as
    zeros and a window stops at its sequence's end (the operand's hexes are per sequence): that is what the reference
    computes for every row the classifier reads.  Ragged column tile, partial row tiles, a shape with several tiles per
    persistent workgroup, 7 / 8 / 9 taps; a second launch is bit-identical."""
    import ctypes as C
    from decode_tonal_langauge_amd import _lib
    from decode_tonal_langauge_amd._lib import NtParams, LOAD_V, EPI_LRELU, check, ptr
    lib = _lib.load()
    gen = torch.Generator().manual_seed(100 * nseq + taps)
    rows = nseq * Tp
    x = torch.randn(rows, cin, generator=gen)
    w = torch.randn(cout, cin, taps, generator=gen) / np.sqrt(cin * taps)
    b = torch.randn(cout, generator=gen)
    xd, wd, bd = x.to(dev), w.to(dev).contiguous(), b.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    nh = rows // 6
    nh_pad = (nh + 2 + 127) // 128 * 128
    V0 = torch.zeros(nh_pad, 8, cin, device=dev)
    V1 = torch.zeros(nh_pad, 8, cin, device=dev)
    check(lib.tl_wino63_xform2(ptr(xd), ptr(V0), ptr(V1), rows, Tp, tvalid, cin, cin, cin, st), "tl_wino63_xform2")
    wp = torch.empty(3 * cin // 8, 8, cout, 8, device=dev)
    check(lib.tl_wino63_weights7(ptr(wd), ptr(wp), cout, cin, taps, st), "tl_wino63_weights7")
    outs = []
    for _ in range(2):
        out = torch.full((rows, cout + 4), float("nan"), device=dev)
        p = NtParams()
        p.A, p.aux, p.Bw, p.bias, p.out = ptr(V0), ptr(V1), ptr(wp), ptr(bd), ptr(out)
        p.M, p.A_rows, p.N, p.K, p.lda, p.ldb, p.ldo = rows, nh_pad, cout, cin, cin, 3 * cin, cout + 4
        p.J, p.row_shift, p.Tp, p.Tvalid, p.slope = taps, 0, Tp, Tp, 0.3
        p.loader, p.epilogue, p.splitk, p.bm = LOAD_V, EPI_LRELU, 1, 128
        check(lib.tl_conv7_wino63v_nt(C.byref(p), st), "tl_conv7_wino63v_nt")
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][:, :cout], outs[1][:, :cout]) and bool(torch.isnan(outs[0][:, cout:]).all())
    xs = x.double().view(nseq, Tp, cin).clone()
    xs[:, tvalid:] = 0
    xpad = torch.nn.functional.pad(xs, (0, 0, 0, taps))               # a window does not cross into the next sequence
    ref = torch.zeros(nseq, Tp, cout, dtype=torch.float64)
    for j in range(taps):
        ref += xpad[:, j:j + Tp] @ w[:, :, j].double().T
    ref = torch.nn.functional.leaky_relu(ref + b.double(), 0.3).view(rows, cout)
    got = outs[0][:, :cout].cpu().double()
    assert torch.isfinite(got).all()
    # (the LAST hex of a sequence is outside the contract: its third segment is the first hex of the next sequence - no row of
    # it can be a valid output of a 7-tap convolution, whose valid rows end at tvalid - taps + 1 <= Tp - 6)
    keep = (torch.arange(rows) % Tp) < Tp - 6
    assert float((got - ref)[keep].abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    # bad arguments are refused
    p.ldb = 2 * cin
    assert lib.tl_conv7_wino63v_nt(C.byref(p), st) != 0 and b"leading" in lib.tl_last_error()
    p.ldb, p.J = 3 * cin, 3
    assert lib.tl_conv7_wino63v_nt(C.byref(p), st) != 0
    p.J, p.aux = taps, None
    assert lib.tl_conv7_wino63v_nt(C.byref(p), st) != 0


def _chain_steps():
    return deepcopy(gi.CHAIN_STEPS)


def test_preprocess_dispatch_chain_matches_reference_golden(dev):
    """downsample -> car_rereference -> frequency_filter -> channel_zscore, looked up by module name and
    run on ONE shared Namespace exactly like the reference's preprocess_signal."""
    from decode_tonal_langauge_amd.preprocess import preprocessor
    g = np.load(os.path.join(GOLD, "g13_preprocess_chain.npz"))
    x = gi.chain_input()
    assert abs(gi.checksum(x) - float(g["in_checksum"])) < 1e-9 * float(g["in_checksum"])
    # (a) the reference's module names resolve to this package's kernels
    prm = Namespace(signal_freq=1000)
    out, freq = preprocessor.preprocess_signal(x.copy(), _chain_steps(), prm)
    assert freq == 400 and prm.signal_freq == 400 and isinstance(out, np.ndarray)
    assert out.shape == g["out"].shape == (12, 1200)
    # the Hilbert rows agree to 1e-12; the order-4 0.3-100 Hz Butterworth rows amplify the 1e-15 difference of
    # the resampled input (Bluestein vs pocketfft) to ~2e-8 (poles at |z| = 0.998)
    assert rel(out[:6], g["out"][:6]) < 1e-10
    assert rel(out, g["out"]) < 2e-7
    # every step's keys were merged onto the shared Namespace
    assert prm.downsample_freq == 400 and prm.exclude_channels == [2] and len(prm.bands) == 2
    # (b) fully qualified module names + device-resident chain (one upload, one download)
    steps = _chain_steps()
    for s in steps:
        s["module"] = f"{PKG}." + s["module"]
    out2, _ = preprocessor.preprocess_signal(x.copy(), steps, Namespace(signal_freq=1000), resident=True)
    assert np.array_equal(out2, out)
    # (c) a CUDA tensor in gives a CUDA tensor out
    out3, _ = preprocessor.preprocess_signal(torch.from_numpy(x).to(dev), _chain_steps(), Namespace(signal_freq=1000))
    assert isinstance(out3, torch.Tensor) and out3.is_cuda and np.array_equal(out3.cpu().numpy(), out)
    # (d) a step parameter that is already on the Namespace is refused (reference :46-52)
    with pytest.raises(ValueError, match="already exists"):
        preprocessor.preprocess_signal(x.copy(), _chain_steps(), Namespace(signal_freq=1000, downsample_freq=200))
    dup = _chain_steps() + [{"module": "preprocess.signal.car_rereference", "params": {"exclude_channels": []}}]
    with pytest.raises(ValueError, match="exclude_channels"):
        preprocessor.preprocess_signal(x.copy(), dup, Namespace(signal_freq=1000))
    # (e) preprocess_modalities: sampling rate taken from / written back to <modality>_sf
    dd = {"ecog": x.copy(), "ecog_sf": 1000}
    cfg = {"ecog": {"type": "signal", "preprocessing": {"steps": _chain_steps()}}}
    dd = preprocessor.preprocess_modalities(dd, cfg, Namespace())
    assert dd["ecog_sf"] == 400 and np.array_equal(dd["ecog"], out)
    with pytest.raises(KeyError, match="type"):
        preprocessor.preprocess_modalities({"a": x, "a_sf": 1}, {"a": {}}, Namespace())


def _c5_chain(dev, Craw, n_non, n_cls, NB, seed):
    """raw ECoG (Craw, 400 * NB) N(0,1) @ 400 Hz -> frequency_filter.run (Hilbert 70-150 Hz envelope) -> 400-sample epochs ->
    n_non channels to the synthesiser, n_cls + n_cls to the CNN (syllable) / CNN-RNN (tone) classifiers -> one SynthesisTrainer
    step; against the CPU oracle chain: signal oracle, the classifier modules' own CPU graph (bit-identical to the reference,
    golden G12), prepare_tone_dynamics, synthesis oracle."""
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier, SynthesisModelCNN, SynthesisTrainer
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    from oracle import signal_oracle as sg
    from oracle import synthesis_oracle as so
    T = 400
    raw = np.random.default_rng(seed).standard_normal((Craw, T * NB)).astype(np.float32)
    bands = [{"method": "hilbert", "params": {"freq_ranges": [70., 150.], "envelope": True}}]
    hg = ff.run(torch.from_numpy(raw).to(dev), Namespace(signal_freq=400, bands=bands))        # on device
    hg_ref = sg.run(raw, Namespace(signal_freq=400, bands=bands))
    assert hg.shape == (Craw, T * NB) and rel(hg.cpu().numpy(), hg_ref) < 1e-5                # float32 input
    # epochs (B, C, T); z-score per channel so the classifiers see O(1) inputs
    hgz = (hg - hg.mean(dim=1, keepdim=True)) / hg.std(dim=1, keepdim=True)
    ep = hgz.view(Craw, NB, T).permute(1, 0, 2).float().contiguous()
    e_non = ep[:, :n_non].contiguous()
    e_syl = ep[:, n_non:n_non + n_cls].contiguous()
    e_tone = ep[:, n_non + n_cls:n_non + 2 * n_cls].contiguous()
    tgt = 10 * torch.randn(NB, 80, generator=torch.Generator().manual_seed(4))
    torch.manual_seed(9)
    syl = CNNClassifier(input_channels=n_cls, input_length=T, n_classes=2)
    tone = CNNRNNClassifier(input_channels=n_cls, input_length=T, n_classes=4, lstm_dim=800)
    with torch.no_grad():        # freshly initialised heads score every class ~0.5: widen the margins so that arg-max
        syl.classifier[3].weight.mul_(40.0)                  # is a property of the model, not of fp32 rounding
        tone.output.weight.mul_(40.0)
    model = SynthesisModelCNN(80, n_non, T, dropout=0.0)
    params = {k: v.detach().clone() for k, v in model.named_parameters()}
    # ---- CPU oracle chain on the same epochs ----
    with torch.no_grad():
        p_tone = tone.eval()(e_tone.cpu())
        p_syl = syl.eval()(e_syl.cpu())
    top2 = lambda p: torch.topk(p, 2, dim=1).values
    margin = min(float((top2(p_tone)[:, 0] - top2(p_tone)[:, 1]).min()), float((top2(p_syl)[:, 0] - top2(p_syl)[:, 1]).min()))
    assert margin > 1e-3, f"arg-max margin {margin:.2e} too small for a label comparison: change the seed"
    lab_ref = torch.tensor(so.prepare_tone_dynamics(gi.TONE_MAP, p_tone.argmax(1).numpy(), p_syl.argmax(1).numpy()),
                           dtype=torch.float32)
    st = so.NAdamState(params)
    loss_ref, mcd_ref, g_ref, _o = so.train_step("cnn", params, None, st, e_non.cpu(), lab_ref, tgt, return_grads=True)
    # ---- the MI355X trainer ----
    tr = SynthesisTrainer(model, tone, syl, gi.TONE_MAP, device=dev, verbose=False)
    with torch.no_grad():
        s_tone, s_syl = tr.tone_model(e_tone), tr.syllable_model(e_syl)
    assert tone._hip is not None and syl._hip is not None, "classifier HIP paths were not taken"
    assert float((s_tone.cpu() - p_tone).abs().max()) < 2e-4 and float((s_syl.cpu() - p_syl).abs().max()) < 2e-4
    lab = tr._labels(e_tone, e_syl)
    assert torch.equal(lab.cpu(), lab_ref)
    model.train()
    tr._stats.zero_()
    tr.train_step(e_non, e_syl, e_tone, tgt)
    stats = tr._stats.cpu().numpy()
    assert abs(stats[2] - loss_ref) < 1e-4 * abs(loss_ref)
    assert abs(stats[3] - mcd_ref) < 1e-4 * abs(mcd_ref)
    eng = model._engine
    for k, g_o in g_ref.items():
        if k == eng.lowrank_param:                            # never materialised on the trainer path: compare the factors' product
            fa, fb = eng.whh_factors[0], eng.whh_factors[1]
            blk = slice(0, min(2048, fa.shape[1]))
            got = (fa[:, blk].t() @ fb).cpu().numpy()
            assert rel_l2(got, g_o[blk].numpy()) < 5e-3, k
        else:
            assert rel_l2(tr._grads[k].cpu().numpy(), g_o.numpy()) < 5e-3, k


def test_c5_end_to_end_against_oracle_chain(dev):
    """Config C5 at reduced width and batch: 64 raw channels -> 32 to the synthesiser, 16 + 16 to the classifiers, 6 windows."""
    _c5_chain(dev, Craw=64, n_non=32, n_cls=16, NB=6, seed=21)


def test_c5_end_to_end_full_width(dev):
    """Config C5 at its real widths: raw (256, 400 * B) -> Hilbert -> 128 channels to SynthesisModelCNN (the north-star
    model: 1.38 G parameters), 64 + 64 to CNNClassifier / CNNRNNClassifier(lstm_dim 800); batch 4 to bound the CPU oracle."""
    _c5_chain(dev, Craw=256, n_non=128, n_cls=64, NB=4, seed=22)
