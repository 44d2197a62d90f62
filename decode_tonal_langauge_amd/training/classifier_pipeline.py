"""Classifier experiments over seeds (counterpart of reference training/classifier_pipeline.py:28-478).

``train_joint_targets``   one model on the joint label of all targets;
``train_separate_targets`` one model per target, joint metrics from the combined predictions;
``save_and_plot_results`` one CSV row per (subject, target group) with ``<metric>_<aggregate>`` and
                          ``<metric>_all`` columns, confusion matrices as CSV + PNG.
Same parameters (a flat ``Namespace``), same result dictionaries, same files; the training loop is
``ClassifierTrainer`` instead of ``pl.Trainer`` (no Lightning in the image)."""
from __future__ import annotations

import os
from argparse import Namespace
from typing import Dict, List, Tuple

import numpy as np
import pandas as pd
import torch

from ..data_loading.dataloaders import split_dataset
from ..data_loading.sample_loading import ClassificationSampleHandler
from ..models.classifier_factory import get_classifier_by_name
from ..models.classifier_trainer import ClassifierTrainer
from ..utils.metrics import compute_classification_metrics, compute_classification_metrics_joint
from ..utils.utils import set_seeds
from ..utils.visualise import plot_confusion_matrix


def _fit_one(params: Namespace, dataset, seed: int, n_classes: int, n_channels: int, seq_length: int, tag: str,
             announce: bool):
    """Split, build, train, test and predict one model for one seed.  Returns (model, preds, true)."""
    verbose = getattr(params, "verbose", 1)
    loaders = split_dataset(dataset, [params.train_ratio, params.vali_ratio, params.test_ratio],
                            shuffling=[True, False, False], batch_size=params.batch_size, seed=int(seed))
    true = np.concatenate([b[1].cpu().numpy() for b in loaders[2]])
    model = get_classifier_by_name(params.model, params.device, n_classes, n_channels, seq_length,
                                   classifier_kwargs=getattr(params, "model_kwargs", None))
    if announce:
        print(f"Number of trainable parameters: {model.get_layer_nparams()}")
    run_dir = os.path.join(params.log_dir, f"{tag}_csv", f"subject_{params.subject_id}", "seed_" + str(seed))
    trainer = ClassifierTrainer(model, learning_rate=params.lr, weight_decay=float(getattr(params, "weight_decay", 0.0)),
                                log_dir=run_dir, verbose=verbose > 1)
    trainer.fit(loaders[0], loaders[1], max_epochs=params.epochs, patience=params.patience)
    trainer.test(loaders[2])
    preds = trainer.predict(loaders[2]).cpu().numpy()
    if getattr(params, "save_checkpoints", False):
        model_dir = os.path.join(params.log_dir, "model_checkpoints")
        os.makedirs(model_dir, exist_ok=True)
        path = os.path.join(model_dir, f"{tag}_{params.model_name}_seed_{seed}.pt")
        torch.save(model.state_dict(), path)
        if verbose > 0:
            print(f"Model saved to {path}")
    return model, preds, true


def train_joint_targets(params: Namespace, seeds: np.ndarray) -> Tuple[Dict, np.ndarray, List[str]]:
    verbose = getattr(params, "verbose", 1)
    handler = ClassificationSampleHandler(params)
    data = handler.load_data()
    dataset = handler.prepare_torch_dataset(data["features"], data["labels"], params.device)
    n_samples, n_channels, seq_length = data["features"].shape
    if verbose > 0:
        print(f"Prepared {n_samples} samples with shape {data['features'].shape} "
              f"and labels with shape {data['labels'].shape}")
    n_classes = len(np.unique(data["labels"]))
    class_labels = handler.prepare_class_labels(data["n_classes_dict"])
    metrics = getattr(params, "metrics", ["accuracy"])
    values: Dict[str, List[float]] = {m: [] for m in metrics if m != "confusion_matrix"}
    confusion = np.zeros((n_classes, n_classes)) if "confusion_matrix" in metrics else None
    model_size = 0
    tag = "_".join(params.targets) if len(params.targets) > 1 else params.targets[0]
    for i, seed in enumerate(seeds):
        set_seeds(int(seed))
        model, preds, true = _fit_one(params, dataset, int(seed), n_classes, n_channels, seq_length, tag,
                                      announce=verbose > 0 and i == 0)
        model_size = model.get_nparams()
        got = compute_classification_metrics(true, preds, metrics=metrics, verbose=verbose > 1)
        if confusion is not None and "confusion_matrix" in got:
            confusion += got["confusion_matrix"]
        for m in values:
            values[m].append(got[m])
    info = {**values, "model_size": model_size, "channels": data["selected_channels"], "class_labels": class_labels,
            "seeds": seeds.tolist()}
    return info, confusion, class_labels


def train_separate_targets(params: Namespace, seeds: np.ndarray) -> Tuple[Dict, np.ndarray, List[str]]:
    verbose = getattr(params, "verbose", 1)
    datasets, shapes, channels, n_cls, names = {}, {}, {}, {}, {}
    for target in params.targets:
        tp = Namespace(**vars(params))
        tp.targets = [target]
        handler = ClassificationSampleHandler(tp)
        data = handler.load_data()
        n_cls[target] = data["n_classes_dict"][target]
        channels[target] = data["selected_channels"]
        names[target] = handler.prepare_class_labels({target: n_cls[target]})
        datasets[target] = handler.prepare_torch_dataset(data["features"], data["labels"], params.device)
        shapes[target] = data["features"].shape[1:]
        if verbose > 0:
            print(f"Prepared {data['features'].shape[0]} samples with shape {data['features'].shape} for target {target}")
    class_labels = ClassificationSampleHandler(params).prepare_class_labels(n_cls)
    n_joint = int(np.prod(list(n_cls.values())))
    metrics = getattr(params, "metrics", ["accuracy"])
    scalar = [m for m in metrics if m != "confusion_matrix"]
    values: Dict[str, List[float]] = {m: [] for m in scalar}
    confusion = np.zeros((n_joint, n_joint)) if "confusion_matrix" in metrics else None
    per_target = {t: {m: [] for m in scalar} for t in params.targets}
    per_target_cm = ({t: np.zeros((n_cls[t], n_cls[t])) for t in params.targets}
                     if "confusion_matrix" in metrics else None)
    model_size = 0
    for i, seed in enumerate(seeds):
        set_seeds(int(seed))
        all_true, all_preds = {}, {}
        for target, dataset in datasets.items():
            if verbose > 1:
                print(f"Training for target: {target} with seed {seed}...")
            model, preds, true = _fit_one(params, dataset, int(seed), n_cls[target], shapes[target][0],
                                          shapes[target][1], target, announce=verbose > 0 and i == 0)
            model_size += model.get_nparams()
            all_true[target], all_preds[target] = true, preds
            got = compute_classification_metrics(true, preds, metrics=metrics)
            for m in scalar:
                per_target[target][m].append(got[m])
            if per_target_cm is not None and "confusion_matrix" in got:
                per_target_cm[target] += got["confusion_matrix"]
        joint = compute_classification_metrics_joint(all_true, all_preds, metrics=metrics, verbose=verbose > 1)
        for m in scalar:
            values[m].append(joint[m])
        if confusion is not None and "confusion_matrix" in joint:
            confusion += joint["confusion_matrix"]
    info = {**values, "model_size": model_size, "channels": channels, "seeds": seeds.tolist(),
            "class_labels": class_labels, "individual_metrics": per_target,
            "individual_confusion_matrix": per_target_cm, "individual_class_labels": names}
    return info, confusion, class_labels


def save_and_plot_results(params: Namespace, result_info: Dict, confusion_matrix: np.ndarray,
                          class_labels: List[str]) -> None:
    metrics = getattr(params, "metrics", ["accuracy"])
    aggregates = getattr(params, "aggregates", ["mean", "std"])
    if isinstance(aggregates, str):
        aggregates = [aggregates]
    targets = list(getattr(params, "targets", []))
    joint_label = ", ".join(targets)

    def channels_of(label: str) -> str:
        info = result_info.get("channels", [])
        if isinstance(info, dict):
            wanted = targets if label == joint_label else [label]
            chosen = set()
            for t in wanted:
                chosen.update(int(c) for c in info.get(str(t), []))
        elif info is None:
            chosen = set()
        else:
            chosen = {int(c) for c in info}
        return ",".join(str(c) for c in sorted(chosen))

    def row_of(per_metric: Dict[str, list], label: str) -> Dict[str, object]:
        row = {"model_name": params.model_name, "model_size": result_info.get("model_size"),
               "subject": params.subject_id, "target": label, "channels": channels_of(label),
               "seeds": str(result_info.get("seeds"))}
        for m in metrics:
            if m == "confusion_matrix":
                continue
            vals = per_metric.get(m, [])
            for agg in aggregates:
                fn = getattr(np, agg, None)
                if fn is None:
                    raise ValueError(f"Aggregate function '{agg}' is not recognized in numpy. "
                                     "Please change evaluation.aggregates parameter.")
                row[f"{m}_{agg}"] = float(fn(vals)) if len(vals) else np.nan
            row[f"{m}_all"] = str(list(vals))
        return row

    rows = [row_of({m: result_info[m] for m in metrics if m != "confusion_matrix"}, joint_label)]
    for target, per_metric in result_info.get("individual_metrics", {}).items():
        rows.append(row_of(per_metric, str(target)))
    result_path = os.path.join(params.log_dir, "results.csv")
    frame = pd.DataFrame(rows)
    if os.path.exists(result_path):
        frame.to_csv(result_path, mode="a", header=False, index=False)
    else:
        frame.to_csv(result_path, index=False)
    print(f"Results saved to {result_path}")

    figure_dir = os.path.join(params.log_dir, f"figures/subject_{params.subject_id}")
    cm_dir = os.path.join(params.log_dir, f"confusion_matrices/subject_{params.subject_id}")
    os.makedirs(figure_dir, exist_ok=True)
    os.makedirs(cm_dir, exist_ok=True)
    if confusion_matrix is not None and "confusion_matrix" in metrics:
        plot_confusion_matrix(confusion_matrix, confusion_matrix.shape[0] <= 10, label_names=class_labels,
                              figure_path=os.path.join(figure_dir, "confusion_matrix.png"))
        print(f"Confusion matrix saved to {figure_dir}/confusion_matrix.png")
        pd.DataFrame(confusion_matrix).to_csv(os.path.join(cm_dir, "confusion_matrix.csv"), index=False)
    for target, cm in (result_info.get("individual_confusion_matrix") or {}).items():
        names = result_info["individual_class_labels"].get(target, class_labels)
        if cm is not None:
            path = os.path.join(figure_dir, f"confusion_matrix_{target}.png")
            plot_confusion_matrix(cm, cm.shape[0] <= 10, label_names=names, figure_path=path)
            print(f"Confusion matrix for {target} saved to {path}")
        pd.DataFrame(cm).to_csv(os.path.join(cm_dir, f"confusion_matrix_{target}.csv"), index=False)
