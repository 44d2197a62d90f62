"""Train classifiers on ECoG samples from a YAML configuration (counterpart of reference
train_classifier.py:19-155; BASELINE config C1 runs this on CPU).

``run(config)`` flattens ``training.params.{io,experiment,training}`` + ``dataset`` + ``model`` +
``evaluation`` into one namespace (reference :23-34), derives the log directory from a hash of that
configuration (:58-65), writes the merged configuration next to the results (:67-90), draws the
per-repeat seeds from ``np.random.seed(seed); randint(0, 10000, repeat)`` (:92-93) and trains every
selected ``subject_<id>.npz`` jointly or per target (:97-113).  Returns the log directory."""
from __future__ import annotations

import os
import sys
from argparse import Namespace

import numpy as np
import yaml

from .training.classifier_pipeline import save_and_plot_results, train_joint_targets, train_separate_targets
from .utils.config import dict_to_namespace, generate_hash_name_from_config, load_config


def run(config: dict) -> str:
    print("Running train_classifier ...")
    training_section = config.get("training", {})
    train_cfg = training_section.get("params", {})
    flat = {}
    for section in ("io", "experiment", "training"):
        flat.update(train_cfg.get(section, {}))
    model_cfg = config.get("model", {})
    dataset_cfg = config.get("dataset", {})
    evaluation_cfg = config.get("evaluation", {})
    combined = {**flat, **dataset_cfg, **model_cfg, **evaluation_cfg}
    params = dict_to_namespace(combined, exclude_keys=["class_labels", "model_kwargs"])
    for key, default in (("channel_selection_dir", ""), ("subject_ids", None), ("class_labels", {}),
                         ("model_kwargs", {}), ("save_checkpoints", False)):
        if not hasattr(params, key):
            setattr(params, key, default)

    sample_dir = getattr(params, "sample_dir", "data/samples")
    if not os.path.exists(sample_dir):
        raise FileNotFoundError(f"Sample directory {sample_dir} does not exist."
                                "Please specify a valid sample_dir in the config.")
    params.sample_dir = sample_dir
    subject_files = sorted(f for f in os.listdir(sample_dir) if f.endswith(".npz") and f.startswith("subject_"))
    if not subject_files:
        raise FileNotFoundError(f"No subject files found in {sample_dir}. "
                                "Ensure files are named like 'subject_<id>.npz'.")
    if getattr(params, "model_name", None) is None and "model" in model_cfg:
        params.model_name = model_cfg["model"].split(".")[-1]

    name = generate_hash_name_from_config(getattr(params, "model_name", "model"), config=combined)
    params.log_dir = os.path.join(getattr(params, "log_dir", "logs"), name)
    os.makedirs(params.log_dir, exist_ok=True)

    merged = {}
    for directory in (params.sample_dir, params.channel_selection_dir):
        path = os.path.join(directory, "config.yaml") if directory else ""
        if path and os.path.exists(path):
            merged.update(load_config(path))
    merged.update(model=model_cfg, training=training_section, dataset=dataset_cfg, evaluation=evaluation_cfg)
    with open(os.path.join(params.log_dir, "config.yaml"), "w") as f:
        yaml.dump(merged, f)

    np.random.seed(getattr(params, "seed", 42))
    seeds = np.random.randint(0, 10000, getattr(params, "repeat", 1))
    wanted = _prepare_subject_filter(params, subject_files)
    for subject_file in subject_files:
        subject_id = subject_file.split("_")[1].split(".")[0]
        if subject_id not in wanted:
            continue
        print("--------- Processing file:", subject_file, "---------")
        sp = _prepare_subject_params(params, subject_id)
        train = train_separate_targets if getattr(params, "separate_models", False) else train_joint_targets
        results, confusion, labels = train(sp, seeds)
        save_and_plot_results(sp, results, confusion, labels)
    return params.log_dir


def _prepare_subject_params(base: Namespace, subject_id: str) -> Namespace:
    sp = Namespace(**vars(base))
    sp.subject_id = subject_id
    sp.sample_path = os.path.join(base.sample_dir, f"subject_{subject_id}.npz")
    channel_file = os.path.join(base.channel_selection_dir, f"subject_{subject_id}.json") \
        if base.channel_selection_dir else None
    # the reference always points at the JSON; a missing file means "all channels" here instead of an error
    sp.channel_file = channel_file if channel_file and os.path.exists(channel_file) else None
    return sp


def _prepare_subject_filter(params: Namespace, subject_files: list) -> list:
    if getattr(params, "subject_ids", None):
        return [str(s) for s in params.subject_ids]
    return [f.replace(".npz", "").replace("subject_", "") for f in subject_files if f.startswith("subject_")]


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit("Usage: python -m decode_tonal_langauge_amd.train_classifier <config.yaml>")
    run(load_config(sys.argv[1]))
