"""Synthesizer training entry point (counterpart of reference train_synthesizer.py).

Two ways in:
 * the reference's argparse CLI (``python -m decode_tonal_langauge_amd.train_synthesizer --sample_path ...``),
   same flags (:29-132);
 * ``run(config)`` for the YAML stage runner (``main.py``): the reference has none - its script is
   not dispatchable by ``main.py`` and fails on import (SURVEY.md section 0 finding 5) - so the
   ``training`` stage can say ``module: decode_tonal_langauge_amd.train_synthesizer`` with
   ``params:`` carrying the same names as the CLI flags.

Inputs (reference :161-201): ``sample_path`` .npz with ``ecog (N,C,T)`` and either ``mel (N, n_mels*frames)``
(pre-computed, what the synthetic generators emit) or ``audio`` (converted by ``utils.audio.audio_to_mel``, the
librosa calls of the reference restated with NumPy / SciPy);
``channel_file`` JSON {active_channels, tone_discriminative, syllable_discriminative};
``config_file`` JSON {mel_kwargs, tone_dynamic_mapping, n_syllables, n_tones[, *_model_kwargs]}.
Output: one CSV row per run with the reference's columns (:369-385).
"""
from __future__ import annotations

import argparse
import json
import os
from argparse import Namespace
from typing import Optional

import numpy as np
import pandas as pd
import torch
from torch.utils.data import TensorDataset

from . import parallel
from .data_loading.dataloaders import split_dataset
from .data_loading.utils import select_non_discriminative_channels
from .models.deep_classifiers import CNNClassifier, CNNRNNClassifier
from .models.simple_classifiers import LogisticRegressionClassifier, ShallowNNClassifier
from .models.synthesis_models import SynthesisLite, SynthesisModelCNN
from .models.synthesis_trainer import SynthesisTrainer
from .utils.utils import set_seeds
from .utils.visualise import plot_training_losses

synthesis_models = ['SynthesisLite', 'SynthesisFull']


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train an audio synthesizer on ECoG data.")
    p.add_argument('--sample_path', type=str, required=True)
    p.add_argument('--subject_id', type=str, required=True)
    p.add_argument('--result_file', type=str, required=True)
    p.add_argument('--figure_dir', type=str, default=None)
    p.add_argument('--audio_dir', type=str, default=None)
    p.add_argument('--channel_file', type=str, default='channel_selections.json')
    p.add_argument('--config_file', type=str, default='config.json')
    p.add_argument('--model_name', type=str, required=True)
    p.add_argument('--syllable_model_path', type=str, default=None)
    p.add_argument('--tone_model_path', type=str, default=None)
    p.add_argument('--synthesis_model_name', type=str, required=True)
    p.add_argument('--syllable_model_name', type=str, required=True)
    p.add_argument('--tone_model_name', type=str, required=True)
    p.add_argument('--audio_sampling_rate', type=int, default=24414)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--repeat', type=int, default=1)
    p.add_argument('--verbose', type=int, default=1)
    p.add_argument('--train_ratio', type=float, default=0.9)
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--epochs', type=int, default=100)
    p.add_argument('--lr', type=float, default=0.0005)
    return p


def _build_classifier(name: str, n_channels: int, seq_length: int, n_classes: int, kwargs: dict, role: str):
    if name == 'ShallowNN':
        return ShallowNNClassifier(input_dim=n_channels * seq_length, n_classes=n_classes, **kwargs)
    if name == 'logistic':
        return LogisticRegressionClassifier(input_dim=n_channels * seq_length, n_classes=n_classes, **kwargs)
    if name == 'CNN':
        return CNNClassifier(input_channels=n_channels, input_length=seq_length, n_classes=n_classes, **kwargs)
    if name == 'CNNRNN':
        return CNNRNNClassifier(input_channels=n_channels, input_length=seq_length, n_classes=n_classes, **kwargs)
    raise ValueError(f"Unknown {role} model name: {name}. Supported models: CNN, ShallowNN, logistic, CNNRNN.")


def _mels_from_dataset(dataset, params, mel_kwargs) -> np.ndarray:
    """Mel targets: a pre-computed 'mel' array of the sample file, else the audio of every sample through
    ``utils.audio.audio_to_mel`` (reference train_synthesizer.py:189-201)."""
    if 'mel' in dataset:
        return np.asarray(dataset['mel'], dtype=np.float32)
    from .utils.audio import audio_to_mel
    return np.array([audio_to_mel(audio, params.audio_sampling_rate, mel_kwargs=mel_kwargs) for audio in dataset['audio']])


def train(params: Namespace) -> dict:
    """Body of the reference script (:137-400) on the MI355X trainer.  Returns the result row."""
    if not os.path.exists(params.sample_path):
        raise FileNotFoundError(f"Data file '{params.sample_path}' does not exist.")
    # data parallel: one process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE).  The process group
    # is created before any GPU call; rank r trains on cuda:LOCAL_RANK, every rank sees the same loaders
    # (same seeds) and takes its row shard of each batch inside SynthesisTrainer; only rank 0 writes files.
    rank, world, local = parallel.init_from_env()
    if world > 1:
        params.device = f"cuda:{local}"
    if 'cuda' in params.device and not torch.cuda.is_available():
        raise RuntimeError("CUDA is not available. Please use 'cpu' as device.")
    chief = rank == 0
    say = print if chief else (lambda *a, **k: None)
    for d in (params.figure_dir, params.audio_dir, os.path.dirname(params.result_file)):
        if chief and d and not os.path.exists(d):
            os.makedirs(d)
    with open(params.channel_file, 'r') as f:
        channel_selections = json.load(f)
    non_disc = select_non_discriminative_channels(channel_selections,
                                                  ['tone_discriminative', 'syllable_discriminative'])
    say('Found {} non-discriminative channels.'.format(len(non_disc)))
    with open(params.config_file, 'r') as f:
        config = json.load(f)
    mel_kwargs = config['mel_kwargs']
    tone_dynamic_mapping = config['tone_dynamic_mapping']
    n_syllables, n_tones = config['n_syllables'], config['n_tones']

    dataset = np.load(params.sample_path)
    ecog = dataset['ecog']
    ecog_non = ecog[:, non_disc, :]
    ecog_syl = ecog[:, channel_selections['syllable_discriminative'], :]
    ecog_tone = ecog[:, channel_selections['tone_discriminative'], :]
    mels = _mels_from_dataset(dataset, params, mel_kwargs)
    say('Number of Mel spectrogram coefficients', mels.shape[1:])
    mels_dim = mels.shape[1]
    seq_length = ecog.shape[2]

    syl_kwargs = config.get('syllable_model_kwargs', {})
    tone_kwargs = config.get('tone_model_kwargs', {})
    syllable_model = _build_classifier(params.syllable_model_name, ecog_syl.shape[1], seq_length, n_syllables,
                                       syl_kwargs, "syllable")
    tone_model = _build_classifier(params.tone_model_name, ecog_tone.shape[1], seq_length, n_tones, tone_kwargs, "tone")
    if params.syllable_model_path is not None:
        syllable_model.load_state_dict(torch.load(params.syllable_model_path))
    if params.tone_model_path is not None:
        tone_model.load_state_dict(torch.load(params.tone_model_path))
    train_classifiers = not (params.syllable_model_path is not None and params.tone_model_path is not None)

    n_samples, n_channels, n_timepoints = ecog_non.shape
    if params.verbose > 0:
        say(f"Prepared {n_samples} ECoG samples with shape {ecog.shape[1:]}")
    tds = TensorDataset(torch.tensor(ecog_non, dtype=torch.float32), torch.tensor(ecog_syl, dtype=torch.float32),
                        torch.tensor(ecog_tone, dtype=torch.float32), torch.tensor(mels, dtype=torch.float32))

    mcds, losses = [], []
    np.random.seed(params.seed)
    seeds = np.random.randint(0, 10000, params.repeat)
    model = None
    for i, seed in enumerate(seeds):
        set_seeds(int(seed))
        ratios = [params.train_ratio, 1 - params.train_ratio]
        loaders = split_dataset(tds, ratios, shuffling=[True, False], batch_size=params.batch_size, seed=int(seed))
        if params.synthesis_model_name == 'SynthesisLite':
            model = SynthesisLite(output_dim=mels_dim, n_channels=n_channels, n_timepoints=n_timepoints)
        elif params.synthesis_model_name == 'SynthesisFull':
            model = SynthesisModelCNN(output_dim=mels_dim, n_channels=n_channels, n_timepoints=n_timepoints)
        else:
            raise ValueError(f"Unknown synthesizer model name: {params.synthesis_model_name}. "
                             f"Supported models: {synthesis_models}.")
        trainer = SynthesisTrainer(synthesize_model=model, syllable_model=syllable_model, tone_model=tone_model,
                                   device=params.device, tone_dynamic_mapping=tone_dynamic_mapping,
                                   learning_rate=params.lr, verbose=chief and params.verbose > 0 and i == 0,
                                   train_classifiers=train_classifiers)
        if params.verbose > 0:
            say(f"Training synthesizer with seed {seed}...")
        history = trainer.train(loaders[0], params.epochs, verbose=chief and params.verbose > 1)
        mcd, recon_mels, origin_mels = trainer.evaluate(loaders[1])
        mcds.append(mcd)
        if params.verbose > 0:
            say(f"Finished trial {i+1} / {params.repeat}. MCD: {mcd:.4f} dB")
        losses.append([loss for loss, _ in history])

    mean_mcd, std_mcd = float(np.mean(mcds)), float(np.std(mcds))
    total_size = model.get_nparams() + syllable_model.get_nparams() + tone_model.get_nparams()
    results = {
        'model_name': params.model_name, 'model_size': total_size,
        'tone_model': params.tone_model_name, 'tone_model_kwargs': str(tone_kwargs),
        'syllable_model': params.syllable_model_name, 'syllable_model_kwargs': str(syl_kwargs),
        'subject': params.subject_id, 'mel_kwargs': str(mel_kwargs), 'seeds': str(seeds.tolist()),
        'batch_size': params.batch_size, 'epochs': params.epochs, 'learning_rate': params.lr,
        'mcd_mean': mean_mcd, 'mcd_std': std_mcd, 'all_mcds': str(mcds),
    }
    results['losses'] = losses
    if not chief:
        return results
    df = pd.DataFrame([{k: v for k, v in results.items() if k != 'losses'}])
    if os.path.exists(params.result_file):
        df.to_csv(params.result_file, mode='a', header=False, index=False)
    else:
        df.to_csv(params.result_file, mode='w', header=True, index=False)
    print('Saved results to ', params.result_file)
    print(f"-------- Training completed over {params.repeat} runs --------")
    print(f"MCD (Mel-Cepstral Distortion): {mean_mcd:.4f} dB ± {std_mcd:.4f} dB")
    if params.figure_dir:
        path = os.path.join(params.figure_dir, 'training_losses.png')
        plot_training_losses(losses, figure_path=path)
        print("Saved training losses figure to ", path)
    if params.audio_dir:
        np.savez(os.path.join(params.audio_dir, 'mels.npz'), origin=origin_mels[:10], recon=recon_mels[:10])
        # the first samples as audio (reference :408-428): Griffin-Lim inversion of the original / synthesised dB mels
        from scipy.io.wavfile import write as write_wave
        from .utils.audio import mel_to_audio
        stft_kw = {k: mel_kwargs[k] for k in ("n_fft", "hop_length", "win_length", "fmin", "fmax") if k in mel_kwargs}
        for i in range(min(int(getattr(params, "n_audio_samples", 10)), len(origin_mels))):
            for tag, mel in (("origin", origin_mels[i]), ("recon", recon_mels[i])):
                wave = mel_to_audio(np.asarray(mel), mel_kwargs['n_mels'], audio_sampling_rate=params.audio_sampling_rate,
                                    **stft_kw)
                if wave.size == 0:          # a single mel frame inverts to zero samples (centred STFT)
                    continue
                path = os.path.join(params.audio_dir, f'{tag}_audio_{i}.wav')
                write_wave(path, int(params.audio_sampling_rate), wave)
                print(f"Saved {tag} audio to ", path)
    return results


def run(config: dict) -> Optional[str]:
    """YAML entry for ``main.py``: ``config['training']['params']`` holds the CLI flag names
    (nested ``io`` / ``training`` / ``experiment`` groups are flattened)."""
    stage = config.get("training", config)
    raw = dict(stage.get("params", stage))
    flat = {}
    for k, v in raw.items():
        if isinstance(v, dict) and k in ("io", "training", "experiment", "model"):
            flat.update(v)
        else:
            flat[k] = v
    defaults = vars(build_parser().parse_args(
        ['--sample_path', '_', '--subject_id', '_', '--result_file', '_', '--model_name', '_',
         '--synthesis_model_name', '_', '--syllable_model_name', '_', '--tone_model_name', '_']))
    required = ['sample_path', 'subject_id', 'result_file', 'model_name', 'synthesis_model_name',
                'syllable_model_name', 'tone_model_name']
    missing = [k for k in required if k not in flat]
    if missing:
        raise KeyError(f"train_synthesizer.run: missing training params {missing}")
    defaults.update({k: v for k, v in flat.items() if k in defaults})
    train(Namespace(**defaults))
    return os.path.dirname(defaults['result_file']) or None


if __name__ == '__main__':
    train(build_parser().parse_args())
