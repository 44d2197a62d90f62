"""Classification metrics on label arrays (counterpart of reference utils/metrics.py:5-139).

Same names, arguments and results: metrics are looked up in a fixed table first
(accuracy, weighted f1_score / precision / recall, cohen_kappa, confusion_matrix) and then by name in
``sklearn.metrics`` (``average='weighted'`` when the function takes it); joint metrics over several
targets flatten the per-target labels into one mixed-radix label, first target most significant
(reference :117-137)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
from sklearn import metrics as _sk

_WEIGHTED = ("f1_score", "precision", "recall")
_TABLE = {
    "accuracy": _sk.accuracy_score,
    "f1_score": _sk.f1_score,
    "precision": _sk.precision_score,
    "recall": _sk.recall_score,
    "cohen_kappa": _sk.cohen_kappa_score,
    "confusion_matrix": _sk.confusion_matrix,
}


def compute_classification_metrics(true: np.ndarray, preds: np.ndarray, metrics: List[str] = ["accuracy"],
                                   verbose: bool = False) -> Dict[str, object]:
    if verbose:
        print("Unique labels in true: {}".format(set(true)))
        print("Unique predictions in preds: {}".format(set(preds)))
    out = {}
    for name in metrics:
        fn = _TABLE.get(name)
        if fn is not None:
            out[name] = fn(true, preds, average="weighted") if name in _WEIGHTED else fn(true, preds)
            continue
        fn = getattr(_sk, name, None)
        if fn is None:
            raise ValueError(f"Metric '{name}' is not recognized in sklearn.metrics, and "
                             f"not part of the supported metrics: {list(_TABLE.keys())}.")
        takes_average = "average" in getattr(getattr(fn, "__code__", None), "co_varnames", ())
        out[name] = fn(true, preds, average="weighted") if takes_average else fn(true, preds)
    return out


def compute_classification_metrics_joint(all_true: Dict[str, np.ndarray], all_preds: Dict[str, np.ndarray],
                                         metrics: List[str] = ["accuracy"], verbose: bool = False) -> Dict[str, object]:
    if set(all_true.keys()) != set(all_preds.keys()):
        raise ValueError("Keys in all_true and all_preds must match.")
    targets = list(all_true.keys())
    if verbose:
        for t in targets:
            print("Unique labels in {}: {}".format(t, set(all_true[t])))
            print("Unique predictions in {}: {}".format(t, set(all_preds[t])))
    true = [np.asarray(all_true[t]).astype(int) for t in targets]
    pred = [np.asarray(all_preds[t]).astype(int) for t in targets]
    radix = [len(np.unique(t)) for t in true]
    joint_true = np.zeros_like(true[0])
    joint_pred = np.zeros_like(pred[0])
    for i in range(len(targets)):
        weight = int(np.prod(radix[i + 1:]))
        joint_true = joint_true + true[i] * weight
        joint_pred = joint_pred + pred[i] * weight
    return compute_classification_metrics(joint_true, joint_pred, metrics)
