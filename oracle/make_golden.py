"""Generate tests/golden/*.npz from the REAL reference and pin the oracle to it.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  The reference is imported read-only, nothing is copied: only inputs, seeds and the
reference's *outputs* are written.  ``models/__init__.py`` of the reference imports
pytorch-lightning (not installed), so the ``models`` package is pre-seeded with an empty
module object whose ``__path__`` points at the reference directory.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--out tests/golden]
Each golden records the max deviation oracle-vs-reference observed when it was made.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; goldens can only be regenerated in the build container")
    sys.path.insert(0, REF)
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    import models.synthesis_models as rsm          # noqa
    import models.synthesis_trainer as rst         # noqa
    import models.simple_classifiers as rsc        # noqa
    import data_loading.utils as rdu               # noqa
    import data_loading.dataloaders as rdl         # noqa
    import preprocess.signal.frequency_filter as rff   # noqa
    return rsm, rst, rsc, rdu, rdl, rff


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def maxrel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))

def pack_big(prefix_map, limit=60000):
    """Full tensors up to `limit` elements; a strided sample + (sum, abs-sum) for bigger ones."""
    keep = {}
    for name, arr in prefix_map.items():
        a = np.asarray(arr)
        if a.size <= limit:
            keep[name] = a
        else:
            keep[name + "@s97"] = a.reshape(-1)[::97].copy()
            keep[name + "@sum"] = np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum()])
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    rsm, rst, rsc, rdu, rdl, rff = _import_reference()
    from oracle import synthesis_oracle as so
    from oracle import signal_oracle as sg
    from tests import golden_inputs as gi
    TONE_MAP = gi.TONE_MAP
    torch.set_num_threads(8)
    report = {}

    # ---- G1: SynthesisModelCNN(80,4,100) eval forward -------------------------------
    torch.manual_seed(0)
    net = rsm.SynthesisModelCNN(80, 4, 100).eval()
    x, lab = gi.g1_inputs()
    with torch.no_grad():
        ref = net(x, lab)
    torch.manual_seed(0)
    p = so.init_cnn_params(80, 4, 100)
    sd = net.state_dict()
    assert list(sd.keys()) == list(p.keys()), (list(sd.keys()), list(p.keys()))
    for k in sd:
        assert torch.equal(sd[k], p[k]), k
    with torch.no_grad():
        mine, inter = so.cnn_forward(p, x, lab, return_intermediates=True)
    report["g1_cnn_forward"] = maxrel(mine, ref)
    with torch.no_grad():
        ref_e5 = net.ecog_conv_block(x.unsqueeze(1).permute(0, 1, 3, 2))
        ref_h = net.label_lstm(lab.permute(0, 2, 1))[0][:, -1, :]
    report["g1_cnn_ecog5"] = maxrel(inter["ecog5"], ref_e5)
    report["g1_cnn_lstm_h"] = maxrel(inter["lstm_h"], ref_h)
    np.savez_compressed(os.path.join(args.out, "g1_cnn_forward.npz"), out=ref.numpy(), seed=0,
                        dims=np.array([80, 4, 100]), ecog5=ref_e5.numpy(), lstm_h=ref_h.numpy(),
                        in_checksum=gi.checksum(x, lab), param_checksum=gi.checksum(*sd.values()))

    # ---- G2: SynthesisLite(80,32,200) eval forward -----------------------------------
    torch.manual_seed(0)
    net = rsm.SynthesisLite(80, 32, 200).eval()
    x, lab = gi.g2_inputs()
    with torch.no_grad():
        ref = net(x, lab)
    torch.manual_seed(0)
    p, b = so.init_lite_params(80, 32, 200)
    sd = net.state_dict()
    for k in p:
        assert torch.equal(sd[k], p[k]), k
    with torch.no_grad():
        mine = so.lite_forward(p, b, x, lab, training=False)
    report["g2_lite_forward"] = maxrel(mine, ref)
    np.savez_compressed(os.path.join(args.out, "g2_lite_forward.npz"), out=ref.numpy(), seed=0,
                        dims=np.array([80, 32, 200]), in_checksum=gi.checksum(x, lab),
                        param_checksum=gi.checksum(*sd.values()))

    # ---- G3: Lite, dropout 0, 3 NAdam train steps through the reference trainer body ----
    def ref_steps(model, xs, labs, tgts, nsteps):
        opt = torch.optim.NAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.004)
        crit = torch.nn.L1Loss()
        losses, mcds, grads1, outs = [], [], None, []
        model.train()
        for s in range(nsteps):
            opt.zero_grad()
            o = model(xs[s], labs[s])
            t = tgts[s].long()
            loss = crit(o, t)
            loss.backward()
            if s == 0:
                grads1 = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
            opt.step()
            losses.append(loss.item())
            mcds.append(rst.compute_mcd(t, o))
            outs.append(o.detach().clone())
        return losses, mcds, grads1, outs

    xs, tones, syls, labs, tg = gi.train_batches(3, 64, 32, 200)
    for t_, s_, l_ in zip(tones, syls, labs):   # the reference's own label builder agrees
        assert np.array_equal(rdu.prepare_tone_dynamics(gi.TONE_MAP, t_.numpy(), s_.numpy()), l_.numpy())
    torch.manual_seed(0)
    net = rsm.SynthesisLite(80, 32, 200, dropout=0.0)
    losses, mcds, grads1, outs = ref_steps(net, xs, labs, tg, 3)
    torch.manual_seed(0)
    p, b = so.init_lite_params(80, 32, 200)
    st = so.NAdamState(p)
    ml, mm = [], []
    for s in range(3):
        l_, m_ = so.train_step("lite", p, b, st, xs[s], labs[s], tg[s])
        ml.append(l_)
        mm.append(m_)
    report["g3_lite_train_loss"] = maxrel(ml, losses)
    torch.manual_seed(0)
    p0, _ = so.init_lite_params(80, 32, 200)
    report["g3_lite_train_update_l2"] = max(
        gi.update_rel_l2(p[k].numpy(), v.detach().numpy(), p0[k].numpy()) for k, v in net.named_parameters())
    np.savez_compressed(os.path.join(args.out, "g3_lite_train.npz"),
                        losses=np.array(losses), mcds=np.array(mcds), out_step0=outs[0].numpy(),
                        out_step2=outs[2].numpy(), seed=0, data_seed=1234,
                        in_checksum=gi.checksum(*xs, *labs, *tg),
                        run_mean0=net.state_dict()["ecog_conv.1.running_mean"].numpy(),
                        run_var0=net.state_dict()["ecog_conv.1.running_var"].numpy(),
                        run_mean1=net.state_dict()["ecog_conv.5.running_mean"].numpy(),
                        run_var1=net.state_dict()["ecog_conv.5.running_var"].numpy(),
                        **pack_big({"grad1." + k: v.numpy() for k, v in grads1.items()}),
                        **pack_big({"final." + k: v.detach().numpy() for k, v in net.named_parameters()}))

    # ---- G4: SynthesisModelCNN(80,16,200, dropout=0) 3 train steps ---------------------
    xs, tones, syls, labs, tg = gi.train_batches(3, 8, 16, 200)
    torch.manual_seed(0)
    net = rsm.SynthesisModelCNN(80, 16, 200, dropout=0.0)
    assert net.get_nparams() == 7169232
    losses, mcds, grads1, outs = ref_steps(net, xs, labs, tg, 3)
    torch.manual_seed(0)
    p = so.init_cnn_params(80, 16, 200)
    st = so.NAdamState(p)
    ml = []
    g_or = None
    for s in range(3):
        l_, m_, g_, _o = so.train_step("cnn", p, None, st, xs[s], labs[s], tg[s], return_grads=True)
        if s == 0:
            g_or = g_
        ml.append(l_)
    report["g4_cnn_train_loss"] = maxrel(ml, losses)
    report["g4_cnn_train_grads"] = max(maxrel(g_or[k], grads1[k]) for k in grads1)
    torch.manual_seed(0)
    p0 = so.init_cnn_params(80, 16, 200)
    report["g4_cnn_train_update_l2"] = max(
        gi.update_rel_l2(p[k].numpy(), v.detach().numpy(), p0[k].numpy()) for k, v in net.named_parameters())
    np.savez_compressed(os.path.join(args.out, "g4_cnn_train.npz"),
                        losses=np.array(losses), mcds=np.array(mcds), out_step0=outs[0].numpy(),
                        out_step2=outs[2].numpy(), seed=0, data_seed=1234,
                        in_checksum=gi.checksum(*xs, *labs, *tg),
                        **pack_big({"grad1." + k: v.numpy() for k, v in grads1.items()}),
                        **pack_big({"final." + k: v.detach().numpy() for k, v in net.named_parameters()}))

    # ---- G5: prepare_tone_dynamics ----------------------------------------------------
    m = {"0": [3, 3, 3], "1": [1, 2, 3]}
    ref = rdu.prepare_tone_dynamics(m, np.array([1, 0]), np.array([0, 1]))
    mine = so.prepare_tone_dynamics(m, [1, 0], [0, 1])
    assert np.array_equal(ref, mine)
    big_t = np.random.default_rng(5).integers(0, 4, 50)
    big_s = np.random.default_rng(6).integers(0, 2, 50)
    ref2 = rdu.prepare_tone_dynamics(TONE_MAP, big_t, big_s)
    assert np.array_equal(ref2, so.prepare_tone_dynamics(TONE_MAP, big_t, big_s))
    np.savez_compressed(os.path.join(args.out, "g5_tone_dynamics.npz"), out_small=ref, tones=big_t, syls=big_s, out=ref2)
    report["g5_tone_dynamics"] = 0.0

    # ---- G6: signal filters -------------------------------------------------------------
    x, x2 = gi.g6_inputs()
    hil = rff.hilbert_filter(x, 400, freq_ranges=[70., 150.])
    hil_re = rff.hilbert_filter(x, 400, freq_ranges=[70., 150.], envelope=False)
    but = rff.butter_filter(x, [0.3, 100], 400)
    but_c = rff.butter_filter(x, [0.3, 100], 400, causal=True)
    fir = rff.fir_bandpass_filter(x, 400, 390, [100.])
    fir2 = rff.fir_bandpass_filter(x, 400, 64, [60., 120.])
    report["g6_hilbert"] = maxrel(sg.hilbert_filter(x, 400, [70., 150.]), hil)
    report["g6_hilbert_real"] = maxrel(sg.hilbert_filter(x, 400, [70., 150.], envelope=False), hil_re)
    report["g6_butter"] = maxrel(sg.butter_filter(x, [0.3, 100], 400), but)
    report["g6_butter_causal"] = maxrel(sg.butter_filter(x, [0.3, 100], 400, causal=True), but_c)
    report["g6_fir"] = maxrel(sg.fir_bandpass_filter(x, 400, 390, [100.]), fir)
    report["g6_fir2"] = maxrel(sg.fir_bandpass_filter(x, 400, 64, [60., 120.]), fir2)
    # multi-range, odd length, float32 input
    hil2 = rff.hilbert_filter(x2, 400, freq_ranges=[(70., 110.), (110., 150.)])
    report["g6_hilbert_f32_odd"] = maxrel(sg.hilbert_filter(x2, 400, [(70., 110.), (110., 150.)]), hil2)
    prm = Namespace(signal_freq=400, bands=[
        {"method": "hilbert", "params": {"freq_ranges": [70., 150.], "envelope": True}},
        {"method": "butter", "params": {"freqs": [0.3, 100], "filter_type": "bandpass"}},
        {"method": "fir", "params": {"order": 390, "center_frequencies": [100.]}}])
    runout = rff.run(x, prm)
    report["g6_run"] = maxrel(sg.run(x, prm), runout)
    np.savez_compressed(os.path.join(args.out, "g6_signal.npz"), in_checksum=gi.checksum(x, x2), hilbert=hil, hilbert_real=hil_re,
                        butter=but, butter_causal=but_c, fir=fir, fir2=fir2, hilbert2=hil2, run=runout)

    # ---- G7: the other preprocess/signal steps ----------------------------------------------
    import preprocess.signal.channel_zscore as r_cz
    import preprocess.signal.zscore_rereference as r_zr
    import preprocess.signal.car_rereference as r_car
    import preprocess.signal.rolling_zscore as r_rz
    xs_ = np.random.default_rng(7).standard_normal((5, 900)) * 3.0 + 1.5
    xs32 = xs_.astype(np.float32)
    cz = r_cz.run(xs_, Namespace())
    cz32 = r_cz.run(xs32, Namespace())
    zr = r_zr.run(xs_, Namespace(rereference_interval=[0.25, 1.5], signal_freq=200))
    car = r_car.run(xs_, Namespace(exclude_channels=[1, 3]))
    rz = r_rz.run(xs_, Namespace(window_length=0.25, signal_freq=200))
    xn = xs_.copy()
    xn[2, 100:130] = np.nan
    rzn = r_rz.run(xn, Namespace(window_length=0.25, signal_freq=200, preserve_nans=False))
    report["g7_channel_zscore"] = maxrel(sg.channel_zscore(xs_), cz)
    report["g7_zscore_rereference"] = maxrel(sg.zscore_rereference(xs_, 50, 300), zr)
    report["g7_car"] = maxrel(sg.car_rereference(xs_, [1, 3]), car)
    report["g7_rolling"] = maxrel(np.nan_to_num(sg.rolling_zscore(xs_, 50)), np.nan_to_num(rz))
    report["g7_rolling_nan"] = maxrel(sg.rolling_zscore(xn, 50, preserve_nans=False), rzn)
    assert np.array_equal(np.isnan(sg.rolling_zscore(xs_, 50)), np.isnan(rz))
    import preprocess.signal.downsample as r_ds
    p_ds = Namespace(signal_freq=1000, downsample_freq=400)
    ds = r_ds.run(xs_, p_ds)
    assert p_ds.signal_freq == 400 and ds.shape == (5, 360)
    ds32 = r_ds.run(xs32, Namespace(signal_freq=1000))
    ds_up = r_ds.run(xs_[:, :601], Namespace(signal_freq=300, downsample_freq=400))      # odd length, up-sampling
    report["g7_downsample"] = maxrel(sg.downsample(xs_, 1000, 400)[0], ds)
    report["g7_downsample_f32"] = maxrel(sg.downsample(xs32, 1000)[0], ds32)
    report["g7_downsample_up"] = maxrel(sg.downsample(xs_[:, :601], 300, 400)[0], ds_up)
    np.savez_compressed(os.path.join(args.out, "g7_steps.npz"), in_checksum=gi.checksum(xs_), channel_zscore=cz,
                        channel_zscore_f32=cz32, zscore_rereference=zr, car=car, rolling=rz, rolling_nan=rzn,
                        downsample=ds, downsample_f32=ds32, downsample_up=ds_up)

    # ---- G8: split_dataset order --------------------------------------------------------
    ds = torch.utils.data.TensorDataset(torch.arange(100).float())
    loaders = rdl.split_dataset(ds, [0.9, 0.1], [True, False], batch_size=8, seed=42)
    idx = [list(l.dataset.indices) for l in loaders]
    mine = so.split_indices(100, [0.9, 0.1], 42)
    assert idx == mine, "split_indices mismatch"
    # first epoch order of the shuffled train loader right after split_dataset
    loaders = rdl.split_dataset(ds, [0.9, 0.1], [True, False], batch_size=8, seed=42)
    first_epoch = np.concatenate([b[0].numpy() for b in loaders[0]])
    np.savez_compressed(os.path.join(args.out, "g8_split.npz"), train_idx=np.array(idx[0]), test_idx=np.array(idx[1]),
                        first_epoch=first_epoch)
    report["g8_split"] = 0.0

    # ---- G9: SynthesisTrainer.train history, Lite + LogisticRegression classifiers ------
    N, C, T = 96, 32, 200
    e_non, e_syl, e_tone, tgt = gi.g9_dataset(N, C, T)
    ds = torch.utils.data.TensorDataset(e_non, e_syl, e_tone, tgt)
    torch.manual_seed(7)
    tone_model = rsc.LogisticRegressionClassifier(8 * T, 4)
    syl_model = rsc.LogisticRegressionClassifier(8 * T, 2)
    loaders = rdl.split_dataset(ds, [0.75, 0.25], [True, False], batch_size=16, seed=11)
    torch.manual_seed(0)
    model = rsm.SynthesisLite(80, C, T, dropout=0.0)
    trainer = rst.SynthesisTrainer(model, tone_model, syl_model, TONE_MAP, device=torch.device("cpu"), verbose=False)
    hist = trainer.train(loaders[0], 2, verbose=False)
    mcd, recon, origin = trainer.evaluate(loaders[1])
    np.savez_compressed(os.path.join(args.out, "g9_trainer.npz"), history=np.array(hist), eval_mcd=mcd,
                        recon=recon, origin=origin, data_seed=1234, cls_seed=7, split_seed=11, model_seed=0,
                        in_checksum=gi.checksum(e_non, e_syl, e_tone, tgt),
                        cls_checksum=gi.checksum(tone_model.linear.weight, syl_model.linear.weight))
    report["g9_trainer"] = 0.0

    # ---- G10: classifier-training plumbing (BASELINE config C1): metrics, sample loading, factory ----
    import json
    import tempfile
    import utils.metrics as rmet
    import data_loading.sample_loading as rsl
    import models.classifier_factory as rcf
    rng = np.random.default_rng(5)
    t_tone, p_tone = rng.integers(0, 4, 300), rng.integers(0, 4, 300)
    t_syl, p_syl = rng.integers(0, 2, 300), rng.integers(0, 2, 300)
    p_tone[:150] = t_tone[:150]
    names = ["accuracy", "f1_score", "precision", "recall", "cohen_kappa", "confusion_matrix", "balanced_accuracy_score"]
    single = rmet.compute_classification_metrics(t_tone, p_tone, names)
    joint = rmet.compute_classification_metrics_joint({"syllable": t_syl, "tone": t_tone}, {"syllable": p_syl, "tone": p_tone}, names)
    subj = gi.c1_subject()
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "subject_1.npz"), **subj)
        with open(os.path.join(td, "subject_1.json"), "w") as f:
            json.dump({"active_channels": list(range(16)), "tone_discriminative": [3, 1, 9, 12],
                       "syllable_discriminative": [1, 2, 14]}, f)
        prm = Namespace(sample_path=os.path.join(td, "subject_1.npz"), channel_file=os.path.join(td, "subject_1.json"),
                        targets=["syllable", "tone"], features="ecog", class_labels={"tone": None, "syllable": ["i", "a"]})
        h = rsl.ClassificationSampleHandler(prm)
        data = h.load_data()
        joint_names = list(h.prepare_class_labels(data["n_classes_dict"]))
        h1 = rsl.ClassificationSampleHandler(Namespace(sample_path=prm.sample_path, targets="tone", features="ecog",
                                                       class_labels={"tone": None}))
        data1 = h1.load_data()
        names1 = list(h1.prepare_class_labels(data1["n_classes_dict"]))
    lr = rcf.get_classifier_by_name("models.simple_classifiers.LogisticRegressionClassifier", "cpu", 4, 16, 100)
    sh = rcf.get_classifier_by_name("models.simple_classifiers.ShallowNNClassifier", "cpu", 4, 16, 100,
                                    classifier_kwargs={"hidden_dim": 32})
    np.savez_compressed(os.path.join(args.out, "g10_classifier_plumbing.npz"),
                        **{"single." + k: np.asarray(v) for k, v in single.items()},
                        **{"joint." + k: np.asarray(v) for k, v in joint.items()},
                        labels=data["labels"], channels=data["selected_channels"], feat_sum=float(data["features"].sum()),
                        joint_names=np.array(joint_names), labels1=data1["labels"], channels1=data1["selected_channels"],
                        names1=np.array(names1), lr_nparams=lr.get_nparams(), shallow_nparams=sh.get_nparams(),
                        in_checksum=gi.checksum(subj["ecog"], subj["tone"], subj["syllable"]))
    report["g10_classifier_plumbing"] = 0.0

    for k, v in report.items():
        print(f"{k:28s} oracle-vs-reference max rel dev = {v:.3e}")
    bad = {k: v for k, v in report.items() if v > (2e-3 if 'update_l2' in k else 2e-5)}   # f32 resample: 3e-7
    with open(os.path.join(args.out, "PINNING.txt"), "w") as f:
        f.write("oracle-vs-reference max relative deviation when the goldens were generated\n")
        f.write(f"torch {torch.__version__}, numpy {np.__version__}\n")
        for k, v in report.items():
            f.write(f"{k} {v:.3e}\n")
    if bad:
        raise SystemExit(f"oracle deviates from reference: {bad}")


if __name__ == "__main__":
    main()
