"""CPU: the C-ABI library loads and exports every symbol include/tonal_hip.h declares; argument
validation (no kernel is launched without a GPU); host-side logic of the Python mirror."""
import ctypes as C
import os
import re
from argparse import Namespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from decode_tonal_langauge_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tonal_hip.h")).read()
    declared = set(re.findall(r"\b(tl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tl_version() >= 100
    assert C.sizeof(_lib.NtParams) == lib.tl_sizeof_nt_params() and C.sizeof(_lib.TnParams) == lib.tl_sizeof_tn_params()


def test_abi_argument_validation_without_gpu():
    from decode_tonal_langauge_amd import _lib
    lib = _lib.load()
    p = _lib.NtParams()
    assert lib.tl_gemm_nt_window(C.byref(p), None) == -1
    assert b"null" in lib.tl_last_error()
    p.A = p.Bw = p.out = 16
    p.M, p.N, p.K, p.lda, p.ldb, p.J, p.Tp = 128, 128, 30, 32, 32, 1, 1
    assert lib.tl_gemm_nt_window(C.byref(p), None) == -1 and b"multiples of 4" in lib.tl_last_error()
    p.K, p.J = 32, 8
    assert lib.tl_gemm_nt_window(C.byref(p), None) == -1 and b"J must be" in lib.tl_last_error()
    t = _lib.TnParams()
    assert lib.tl_gemm_tn_window(C.byref(t), None) == -1
    assert lib.tl_nadam(None, None, None, None, 10, 0., 0., .9, .999, 1., 1e-8, 0., 1., None) == -1
    assert lib.tl_nadam_multi(None, 3, 3, 0., 0., .9, .999, 1., 1e-8, 0., 1., None) == -1 and lib.tl_nadam_multi_chunk() > 0
    assert lib.tl_nadam_lowrank(16, 16, 16, 16, 16, 65, 8, 8, 8, 8, 0., 0., .9, .999, 1., 1e-8, 0., 1., None) == -1
    assert b"rank" in lib.tl_last_error()
    assert lib.tl_conv3_wino43v_nt(None, None) == -1 and lib.tl_conv3_wino43v_tn(None, None) == -1
    assert lib.tl_wino43_weights(None, None, None, 8, 8, 8, 8, None) == -1 and b"null" in lib.tl_last_error()
    assert lib.tl_nadam_lowrank_dh(16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 0., 0., .9, .999, 1., 1e-8, 0., 1., None, 4, 2, None) == -1
    assert b"dh_slab" in lib.tl_last_error()
    assert lib.tl_nadam_lowrank_dh(16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 0., 0., .9, .999, 1., 1e-8, 0., 1., 16, 9, 2, None) == -1
    assert b"U <=" in lib.tl_last_error()
    assert lib.tl_lstm_gw(16, 16, 16, 9, 64, 64, 64, 64, 512, None) == -1 and b"U <=" in lib.tl_last_error()
    assert lib.tl_lstm_gw(16, 16, 16, 8, 64, 64, 64, 64, 2048, None) == -1 and b"rows_per_block" in lib.tl_last_error()
    assert lib.tl_filtfilt_scan_f64(16, 1, 16, 16, 16, 16, 7, 16, 16, 16, 2, 5000, 10, 128, None) == -1 and b"ntaps" in lib.tl_last_error()
    assert lib.tl_filtfilt_scan_f64(16, 1, 16, 16, 16, 16, 3, 16, 16, 16, 2, 5000, 9, 128, None) == -1 and b"levels" in lib.tl_last_error()
    assert lib.tl_filtfilt_f64(16, 1, 16, 16, 16, 16, 16, 2, 20, 9, None) == -1
    assert b"padlen" in lib.tl_last_error()
    assert lib.tl_gauss_envelope(16, 1, 16, 16, 2, 100, 8, 200, 0, 1, None) == -1
    # F(4,3) weight-gradient epilogue, inference LSTM
    assert lib.tl_wino43_wgrad_finalize(None, None, 4, 4, 4, None) == -1
    # round-4 entry points: Winograd F(6,3) on pre-transformed operands (argument checks only: no GPU call is reached)
    assert lib.tl_conv3_wino63v_nt(None, None) == -1 and lib.tl_conv3_wino63v_tn(None, None) == -1
    assert lib.tl_wino63_weights(16, 16, None, 8, 12, 12, 8, None) == -1 and b"multiples of 8" in lib.tl_last_error()
    assert lib.tl_wino63_wgrad_finalize(None, None, 4, 4, 4, None) == -1
    assert lib.tl_wino63_vd_fixup(None, None, 4, 1, 2, 8, 8, None) == -1
    assert lib.tl_wino63_unpool_yvd(16, 16, 16, 16, 24, 12, 12, 6, 8, 12, 12, 1, 16, None) == -1 and b"C %% 8" not in lib.tl_last_error()
    assert b"unpool_yvd" in lib.tl_last_error()
    assert lib.tl_wino63_v_fixup(16, 16, 4, 1, 8, 8, 8, None) == -1 and b"Tq" in lib.tl_last_error()           # Tq % 6
    assert lib.tl_conv1_fwd_v6(16, 16, 16, None, 16, 16, None, 4, 100, 3, 96, 48, 49, 0.01, None) == -1 and b"C1" in lib.tl_last_error()
    assert lib.tl_conv1_fwd_v6(16, 16, 16, None, 16, 16, None, 4, 100, 3, 128, 50, 49, 0.01, None) == -1 and b"multiple of 6" in lib.tl_last_error()
    t6 = _lib.TnParams()
    t6.A = t6.B = t6.slab = t6.bbits = 16
    t6.J, t6.loader, t6.Krows, t6.Mdim, t6.Ndim, t6.lda, t6.ldb, t6.ldc, t6.Tp, t6.Tvalid = 3, 1, 24, 128, 64, 128, 64, 64, 8, 4
    t6.A_rows, t6.B_rows, t6.ld_bbits = 128, 12, 2
    assert lib.tl_conv3_wino63v_tn(C.byref(t6), None) == -1 and b"Tp" in lib.tl_last_error()                   # Tp % 6
    t6.Tp, t6.Mdim = 12, 64
    assert lib.tl_conv3_wino63v_tn(C.byref(t6), None) == -1 and b"Mdim" in lib.tl_last_error()                 # C_in % 128
    import ctypes as C2
    flag = C2.c_int(0)
    assert lib.tl_lstm_infer_seq_fused(16, 100, 16, 16, 16, 16, 4, 12, 3, C2.byref(flag), None) == -1
    assert b"multiple of 8" in lib.tl_last_error()
    assert lib.tl_lstm_infer_seq_fused(16, 10, 16, 16, 16, 16, 4, 8, 3, C2.byref(flag), None) == -1
    assert b"row stride" in lib.tl_last_error()
    # round-3 signal entry points: overlap-save banks
    assert lib.tl_hilbert_ols(None, 1, 16, 16, 16, 2, 4096, 8, 104, 1024, 1, None) == -1 and b"null" in lib.tl_last_error()
    assert lib.tl_hilbert_ols(16, 1, 16, 16, 16, 2, 4096, 8, 104, 512, 1, None) == -1 and b"nfft" in lib.tl_last_error()
    assert lib.tl_hilbert_ols(16, 1, 16, 16, 16, 2, 4096, 8, 300, 1024, 1, None) == -1 and b"taps" in lib.tl_last_error()
    assert lib.tl_hilbert_ols_bl(16, 1, 16, None, 16, 16, 2, 4096, 8, 104, 1024, 1, None) == -1 and b"null" in lib.tl_last_error()
    assert lib.tl_hilbert_ols_bl(16, 1, 16, 16, 16, 16, 2, 1000, 8, 104, 1024, 1, None) == -1
    assert b"at least 1024 samples" in lib.tl_last_error()
    assert lib.tl_hilbert_ols_bl(16, 1, 16, 16, 16, 16, 2, 4096, 65, 104, 1024, 1, None) == -1 and b"bands" in lib.tl_last_error()
    assert lib.tl_fir_bank_ols(16, 1, 16, 16, 16, 1, 2, 4096, 1, 600, None) == -1 and b"taps" in lib.tl_last_error()
    assert lib.tl_stage_step(16, 16, 0., 0., 0., 1, None, None, None, 5, None) == -1 and b"at most 4" in lib.tl_last_error()
    assert lib.tl_stage_step(16, 16, 0., 0., 0., 1, None, None, None, 2, None) == -1 and b"null table" in lib.tl_last_error()
    assert lib.tl_linear_rows(None, 16, 16, 16, 4, 8, 2, 8, 0, None) == -1 and b"null" in lib.tl_last_error()
    assert lib.tl_linear_rows(16, 16, 16, 16, 4, 8, 65, 8, 0, None) == -1 and b"output columns" in lib.tl_last_error()
    assert lib.tl_linear_rows(16, 16, 16, 16, 4, 6, 2, 8, 0, None) == -1 and b"multiples of 4" in lib.tl_last_error()
    assert lib.tl_linear_rows(16, 16, 16, 16, 4, 8, 2, 8, 2, None) == -1 and b"act" in lib.tl_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(-1, "x")


def test_models_refuse_cpu_tensors_and_keep_reference_state_dict():
    from decode_tonal_langauge_amd.models import SynthesisLite, SynthesisModelCNN, SynthesisTrainer
    from decode_tonal_langauge_amd.models.simple_classifiers import LogisticRegressionClassifier
    from oracle import synthesis_oracle as so
    torch.manual_seed(0)
    m = SynthesisModelCNN(80, 4, 100)
    torch.manual_seed(0)
    ref = so.init_cnn_params(80, 4, 100)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(torch.equal(sd[k], ref[k]) for k in ref)
    assert m.get_nparams() == sum(v.numel() for v in ref.values()) and m.latent_len == 5
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.randn(2, 4, 100), torch.randn(2, 2, 5))
    torch.manual_seed(0)
    lite = SynthesisLite(80, 32, 200)
    torch.manual_seed(0)
    pl, bl = so.init_lite_params(80, 32, 200)
    sdl = lite.state_dict()
    assert all(torch.equal(sdl[k], pl[k]) for k in pl) and lite.get_nparams() == 919312
    with pytest.raises(RuntimeError):
        lite(torch.randn(2, 32, 200), torch.randn(2, 2, 5))
    with pytest.raises(RuntimeError, match="cuda"):
        SynthesisTrainer(m, LogisticRegressionClassifier(8, 4), LogisticRegressionClassifier(8, 2), {"0": [1]})
    with pytest.raises(ValueError):
        LogisticRegressionClassifier(8, 1)
    with pytest.raises(ValueError, match="Expected input dimension"):
        LogisticRegressionClassifier(8, 2)(torch.randn(3, 9))


def test_engine_geometry_north_star(monkeypatch):
    from decode_tonal_langauge_amd._cnn_engine import CnnEngine
    from oracle.synthesis_oracle import ECOG_STAGES
    stages = [(c if c else 64, k, p) for c, k, p in ECOG_STAGES]
    monkeypatch.setenv("TONAL_WINO", "4")
    e = CnnEngine(80, 128, 400, 6, 64, 0.5, 0.01, stages, [128, 128, 128, 128, 64])
    assert (e.tout1, e.tp1) == (199, 200) and not e.wino63
    assert [(s.tin, s.tc, s.tout, s.tp_in, s.tp_out) for s in e.stages] == [
        (199, 197, 98, 200, 100), (98, 96, 48, 100, 50), (48, 48, 24, 50, 25), (24, 24, 24, 25, 25)]
    assert e.lat == 24 and e.H == 18432 and e.kflat == 128 * 25 * 64 and e.ldx == 72
    # default since round 4: F(6,3) for stages 2 and 3 - sequences of whole hexes (204 = 34 x 6 rows, 102 = 17 x 6), the
    # output of stage 3 back on the default row stride (50, not 51): nothing behind stage 3 changes
    monkeypatch.delenv("TONAL_WINO")
    e = CnnEngine(80, 128, 400, 6, 64, 0.5, 0.01, stages, [128, 128, 128, 128, 64])
    assert e.wino63 and (e.tout1, e.tp1) == (199, 204)
    assert [(s.tin, s.tc, s.tout, s.tp_in, s.tp_out) for s in e.stages] == [
        (199, 197, 98, 204, 102), (98, 96, 48, 102, 50), (48, 48, 24, 50, 25), (24, 24, 24, 25, 25)]
    assert e.lat == 24 and e.H == 18432 and e.kflat == 128 * 25 * 64 and e.ldx == 72
    assert abs(e.f63_issue_factor(e.stages[0]) - 34 * 8 / (197 * 3)) < 1e-12
    # a stack the F(6,3) kernels do not cover (C_out of stage 3 not a multiple of 64) keeps the F(4,3) geometry
    e3 = CnnEngine(80, 8, 200, 4, 8, 0.0, 0.01, [(128, 3, True), (128, 3, True), (96, 3, True), (32, 1, True), (8, 1, False)], [16, 8])
    assert not e3.wino63 and e3.tp1 % 8 == 0
    e2 = CnnEngine(80, 16, 200, 6, 64, 0.0, 0.01, stages, [128, 128, 128, 128, 64])
    assert e2.lat == 11 and e2.H == 1056
    with pytest.raises(ValueError):
        CnnEngine(80, 4, 100, 6, 64, 0.0, -0.1, stages, [128])


def test_nadam_scalars_match_torch():
    from decode_tonal_langauge_amd.optim import nadam_scalars
    from oracle.synthesis_oracle import nadam_scalars as oracle_scalars
    mp = mp_o = 1.0
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5]))
    opt = torch.optim.NAdam([p], lr=5e-4, weight_decay=0.004)
    q = p.detach().clone()
    m = torch.zeros(3)
    v = torch.zeros(3)
    for step in range(1, 6):
        g = torch.tensor([0.3 * step, -0.1, 2.0])
        p.grad = g.clone()
        opt.step()
        cg, cm, bc2, mp = nadam_scalars(step, mp, 5e-4, 0.9, 0.999, 0.004)
        og, om, ob, mp_o = oracle_scalars(step, mp_o, 5e-4, 0.9, 0.999, 0.004)      # the oracle's own running product
        assert (cg, cm, bc2, mp) == (og, om, ob, mp_o)
        gg = g + 0.004 * q
        m = m + (gg - m) * 0.1
        v = 0.999 * v + 0.001 * gg * gg
        den = (v / bc2).sqrt() + 1e-8
        q = q - cg * gg / den - cm * m / den
        assert torch.allclose(q, p.detach(), rtol=1e-6, atol=1e-9)


def test_shard_rows_and_config_helpers(tmp_path):
    from decode_tonal_langauge_amd import parallel
    from decode_tonal_langauge_amd.utils import config as cfg
    for n, w in ((256, 8), (10, 4), (7, 3), (5, 8)):
        got = []
        for r in range(w):
            s = parallel.shard_rows(n, r, w)
            got += list(range(n))[s]
        assert got == list(range(n))
    ns = cfg.dict_to_namespace({"a": 1, "b": {"c": [1, {"d": 2}]}})
    assert ns.b.c[1].d == 2
    assert cfg.generate_hash_name_from_config("x", {"k": 1}) == cfg.generate_hash_name_from_config("x", {"k": 1})
    p = tmp_path / "c.yaml"
    p.write_text("a: 1\n")
    assert cfg.load_config(str(p)) == {"a": 1}


def test_frequency_filter_host_side():
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    from oracle import signal_oracle as sg
    c1, s1 = ff.gaussian_bank([70., 150.], 400)
    c2, s2 = sg.gaussian_bank([70., 150.], 400)
    assert np.array_equal(c1, c2) and np.array_equal(s1, s2)
    x = np.random.default_rng(3).standard_normal((1, 600))
    taps, half = ff.analytic_taps(600, 400, c1, s1)
    assert taps.shape[1] == 2 * half + 1 < 600
    y = np.zeros(600)
    for b in range(len(c1)):
        acc = np.zeros(600, dtype=complex)
        for k in range(taps.shape[1]):
            acc += taps[b, k] * np.roll(x[0], k - half)
        y += np.abs(acc)
    ref = sg.hilbert_filter(x, 400, [70., 150.])[0]
    assert np.max(np.abs(y / len(c1) - ref)) < 1e-12 * np.max(np.abs(ref))
    # error behaviour of the plugin entry mirrors the reference (frequency_filter.py:35-36,44-47)
    with pytest.raises(ValueError, match="bands must be specified"):
        ff.run(x, Namespace(signal_freq=400, bands=None))
    with pytest.raises(ValueError, match="freq_ranges"):
        ff.run(x, Namespace(signal_freq=400, bands=[{"method": "hilbert", "params": {}}]))
    with pytest.raises(ValueError, match="order"):
        ff.run(x, Namespace(signal_freq=400, bands=[{"method": "fir", "params": {"order": 3}}]))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ff.hilbert_filter(x, 400, [70., 150.])


def test_tone_dynamics_host_mirror():
    from decode_tonal_langauge_amd.data_loading.utils import prepare_tone_dynamics, select_non_discriminative_channels
    out = prepare_tone_dynamics({"0": [3, 3, 3], "1": [1, 2, 3]}, np.array([1, 0]), np.array([0, 1]))
    assert out.tolist() == [[[0, 0, 0], [1, 2, 3]], [[1, 1, 1], [3, 3, 3]]]
    with pytest.raises(ValueError, match="not found"):
        prepare_tone_dynamics({"0": [1]}, np.array([5]), np.array([0]))
    with pytest.raises(ValueError, match="must match"):
        prepare_tone_dynamics({"0": [1]}, np.array([0, 0]), np.array([0]))
    sel = {"active_channels": [5, 1, 2, 9], "tone_discriminative": [2], "syllable_discriminative": [9, 7]}
    assert select_non_discriminative_channels(sel, ["tone_discriminative", "syllable_discriminative"]) == [1, 5]


def test_deep_classifier_mirrors_shapes_and_errors():
    from decode_tonal_langauge_amd.models import CNNClassifier, CNNRNNClassifier
    c = CNNClassifier(input_channels=2, input_length=150, n_classes=2)
    assert c.latent_length == 1 and list(c.state_dict())[0] == "feature_extractor.0.weight"
    assert "classifier.1.weight" in c.state_dict() and c.get_layer_nparams().keys() == {"feature_extractor", "classifier"}
    with torch.no_grad():
        assert c.eval()(torch.randn(3, 2, 150)).shape == (3, 2)
    with pytest.raises(ValueError, match="too small"):
        CNNClassifier(input_channels=2, input_length=60, n_classes=2)
    r = CNNRNNClassifier(input_channels=3, input_length=100, n_classes=4, lstm_dim=200)
    assert {"lstm1.weight_ih_l0", "conv_pool_block1.0.weight", "conv_block3.2.bias", "lstm2.weight_hh_l0",
            "output.bias"} <= set(r.state_dict())
    with torch.no_grad():
        out = r.eval()(torch.randn(2, 3, 100))
    assert out.shape == (2, 4) and float(out.min()) >= 0 and float(out.max()) <= 1
    with pytest.raises(ValueError, match="divisible"):
        CNNRNNClassifier(input_channels=3, input_length=100, n_classes=4, lstm_dim=250)
    with pytest.raises(ValueError, match="Expected 3 channels"):
        r(torch.randn(2, 4, 100))


def test_preprocess_dispatcher_host_logic():
    """preprocess_signal's contract (reference preprocess/preprocessor.py:39-70) with CPU stand-in steps:
    lookup by module name, one shared Namespace, steps may mutate it, duplicate keys are refused."""
    import sys
    import types
    from argparse import Namespace
    import numpy as np
    from decode_tonal_langauge_amd.preprocess import preprocessor as pp
    halve = types.ModuleType("fake_steps_halve")

    def run_halve(data, params):
        params.signal_freq = params.signal_freq // params.factor
        return data[:, ::params.factor]
    halve.run = run_halve
    scale = types.ModuleType("fake_steps_scale")
    scale.run = lambda data, params: data * params.gain * params.signal_freq        # reads the mutated rate
    sys.modules["fake_steps_halve"], sys.modules["fake_steps_scale"] = halve, scale
    try:
        x = np.arange(24, dtype=np.float64).reshape(2, 12)
        prm = Namespace(signal_freq=1000)
        steps = [{"module": "fake_steps_halve", "params": {"factor": 2}}, {"module": "fake_steps_scale", "params": {"gain": 3.0}}]
        out, freq = pp.preprocess_signal(x, steps, prm)
        assert freq == 500 and np.array_equal(out, x[:, ::2] * 3.0 * 500)
        assert prm.factor == 2 and prm.gain == 3.0
        with pytest.raises(ValueError, match="'factor' already exists"):
            pp.preprocess_signal(x, steps + [{"module": "fake_steps_halve", "params": {"factor": 2}}], Namespace(signal_freq=1000))
        dd = pp.preprocess_modalities({"ecog": x, "ecog_sf": 1000, "audio": x, "audio_sf": 8},
                                      {"ecog": {"type": "signal", "preprocessing": {"steps": steps}},
                                       "audio": {"type": "signal"}}, Namespace())
        assert dd["ecog_sf"] == 500 and dd["audio_sf"] == 8 and dd["ecog"].shape == (2, 6)
    finally:
        del sys.modules["fake_steps_halve"], sys.modules["fake_steps_scale"]
    # the reference tree's module names (and the sample YAML's stale ones) resolve to this package
    for name in ("preprocess.signal.frequency_filter", "preprocess.downsample", "channel_zscore"):
        mod = pp.resolve_step_module(name)
        assert mod.__name__.startswith("decode_tonal_langauge_amd.preprocess.signal.") and hasattr(mod, "run")


def test_trainer_row_shards_cover_ragged_batches():
    """SynthesisTrainer._shard: weights sum to 1 and rows partition the batch for every (rows, ranks),
    including batches with fewer rows than ranks (spare ranks recompute a row with weight 0)."""
    import torch
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    for world in (2, 3, 8):
        for n in (1, 2, 5, 7, 8, 9, 16, 17):
            seen, wsum = [], 0.0
            for rank in range(world):
                tr = SynthesisTrainer.__new__(SynthesisTrainer)
                tr.rank, tr.world = rank, world
                (rows,) = tr._shard(torch.arange(n))
                assert rows.numel() >= 1                      # nobody sits out a collective
                wsum += tr._weight
                if tr._weight > 0:
                    assert abs(tr._weight - rows.numel() / n) < 1e-12 and int(rows[0]) == tr._row0
                    seen += rows.tolist()
            assert sorted(seen) == list(range(n)) and abs(wsum - 1.0) < 1e-12, (world, n)


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without WORLD_SIZE starts the ranks itself; on a box without N GPUs it
    must fail loudly (non-zero) instead of timing one GPU."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2))], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def test_state_dict_of_a_row_sharded_model_raises_instead_of_hiding_a_collective():
    """with the row-sharded label LSTM a rank holds current values only for its own rows of W_hh between
    steps.  ``model.state_dict()`` must not run the all-gather implicitly (``if rank == 0: torch.save(model.state_dict())``
    would deadlock): the pre-hook raises and names ``sync_parameters()`` / ``model_state_dict()``."""
    import types
    import pytest
    from decode_tonal_langauge_amd.models.synthesis_trainer import SynthesisTrainer
    fake = types.SimpleNamespace(_whh_dirty=True)
    with pytest.raises(RuntimeError, match="sync_parameters"):
        SynthesisTrainer._state_dict_guard(fake, None, "", False)
    fake._whh_dirty = False
    SynthesisTrainer._state_dict_guard(fake, None, "", False)      # in sync: nothing to do


def test_kernel_setting_is_one_validated_variable(monkeypatch):
    """TONAL_KERNELS: unknown keys / values raise where the setting is read; the per-switch variables of rounds 1-4 are
    honoured only under TONAL_AB=1 (the A/B tests) and raise otherwise, naming their replacement."""
    from decode_tonal_langauge_amd import _kernels
    monkeypatch.delenv("TONAL_KERNELS", raising=False)
    monkeypatch.delenv("TONAL_WINO", raising=False)
    assert _kernels.get("wino") == "6" and _kernels.get("conv7") == "wino63" and _kernels.get("hilbert_f32") == "0"
    monkeypatch.setenv("TONAL_KERNELS", "wino=4, whh_dh=0,hilbert=ols_full")
    assert _kernels.get("wino") == "4" and _kernels.get("whh_dh") == "0" and _kernels.get("hilbert") == "ols_full"
    assert _kernels.get("f63_yprod") == "1"                            # not mentioned: the default
    assert len(_kernels.KEYS) <= 12                                    # round 6: one switch per decision that still exists
    for bad in ("wino=5", "wino=1", "winograd=4", "wino", "fuse_c1=0", "hilbert=fast", "butter=fast"):
        monkeypatch.setenv("TONAL_KERNELS", bad)
        with pytest.raises(ValueError):
            _kernels.validate()
    monkeypatch.setenv("TONAL_KERNELS", "wino=0")
    monkeypatch.setenv("TONAL_WINO", "4")                              # TONAL_KERNELS wins over a legacy variable
    assert _kernels.get("wino") == "0"
    monkeypatch.delenv("TONAL_KERNELS")
    monkeypatch.setenv("TONAL_AB", "1")
    assert _kernels.get("wino") == "4"
    monkeypatch.setenv("TONAL_WINO", "7")
    with pytest.raises(ValueError):
        _kernels.get("wino")
    monkeypatch.setenv("TONAL_WINO", "4")
    monkeypatch.delenv("TONAL_AB")
    with pytest.raises(RuntimeError, match="TONAL_KERNELS=wino=4"):
        _kernels.get("wino")
    with pytest.raises(RuntimeError):
        _kernels.validate()
