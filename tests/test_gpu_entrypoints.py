"""GPU: the synthesis entry points end to end on synthetic files in the reference's on-disk
schemas (subject npz, channel JSON, config JSON -> results CSV), through main.py's YAML runner."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import yaml

pytestmark = pytest.mark.gpu


def _make_inputs(tmp, N=48, C=24, T=200):
    rng = np.random.default_rng(0)
    tone = rng.integers(0, 4, N)
    syl = rng.integers(0, 2, N)
    ecog = rng.standard_normal((N, C, T)).astype(np.float32)
    np.savez(os.path.join(tmp, "subject_1.npz"), ecog=ecog, ecog_sf=200,
             mel=(10 * rng.standard_normal((N, 80))).astype(np.float32), tone=tone, syllable=syl)
    ch = {"active_channels": list(range(C)), "tone_discriminative": [0, 1, 2, 3], "syllable_discriminative": [4, 5, 6, 7]}
    json.dump(ch, open(os.path.join(tmp, "channels.json"), "w"))
    cfg = {"mel_kwargs": {"n_mels": 80}, "n_syllables": 2, "n_tones": 4,
           "tone_dynamic_mapping": {"0": [3, 3, 3, 3, 3], "1": [1, 2, 3, 4, 5], "2": [3, 2, 1, 2, 4], "3": [5, 4, 3, 2, 1]}}
    json.dump(cfg, open(os.path.join(tmp, "config.json"), "w"))


def test_train_synthesizer_via_yaml_runner(tmp_path):
    from decode_tonal_langauge_amd.main import run_pipeline
    tmp = str(tmp_path)
    _make_inputs(tmp)
    params = dict(sample_path=os.path.join(tmp, "subject_1.npz"), subject_id="1",
                  result_file=os.path.join(tmp, "out", "results.csv"), figure_dir=os.path.join(tmp, "fig"),
                  channel_file=os.path.join(tmp, "channels.json"), config_file=os.path.join(tmp, "config.json"),
                  model_name="lite-test", synthesis_model_name="SynthesisLite", syllable_model_name="logistic",
                  tone_model_name="logistic", device="cuda:0", batch_size=8, epochs=2, repeat=2, verbose=0)
    y = {"training": {"module": "decode_tonal_langauge_amd.train_synthesizer", "params": params}}
    ypath = os.path.join(tmp, "cfg.yaml")
    yaml.safe_dump(y, open(ypath, "w"))
    run_pipeline(ypath)
    df = pd.read_csv(params["result_file"])
    assert list(df.columns) == ['model_name', 'model_size', 'tone_model', 'tone_model_kwargs', 'syllable_model',
                                'syllable_model_kwargs', 'subject', 'mel_kwargs', 'seeds', 'batch_size', 'epochs',
                                'learning_rate', 'mcd_mean', 'mcd_std', 'all_mcds']
    assert len(df) == 1 and np.isfinite(df.mcd_mean[0]) and df.batch_size[0] == 8
    assert os.path.exists(os.path.join(tmp, "fig", "training_losses.png"))


def test_train_synthesizer_cli_full_model(tmp_path):
    from decode_tonal_langauge_amd import train_synthesizer as ts
    tmp = str(tmp_path)
    _make_inputs(tmp, N=16, C=12, T=100)
    args = ts.build_parser().parse_args([
        "--sample_path", os.path.join(tmp, "subject_1.npz"), "--subject_id", "1",
        "--result_file", os.path.join(tmp, "r", "res.csv"), "--channel_file", os.path.join(tmp, "channels.json"),
        "--config_file", os.path.join(tmp, "config.json"), "--model_name", "full-test",
        "--synthesis_model_name", "SynthesisFull", "--syllable_model_name", "ShallowNN", "--tone_model_name", "logistic",
        "--epochs", "2", "--batch_size", "4", "--verbose", "0"])
    res = ts.train(args)
    assert np.isfinite(res["mcd_mean"]) and len(res["losses"][0]) == 2
    with pytest.raises(ValueError, match="Unknown"):
        ts._build_classifier("nope", 4, 10, 2, {}, "tone")
