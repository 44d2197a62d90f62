"""Training loop for ``ClassifierModel``s (counterpart of reference models/classifier_trainer.py:22-177
plus the ``pl.Trainer`` / ``EarlyStopping`` / ``CSVLogger`` it is driven by in
training/classifier_pipeline.py:131-160).

pytorch_lightning and torchmetrics are not part of the MI355X image, so the same behaviour is written
out as a plain loop:
  * loss ``nn.CrossEntropyLoss`` on ``labels.long()``; optimiser ``NAdam`` with two groups - parameters
    with ndim >= 2 get ``weight_decay``, the rest 0 (reference :63-74);
  * one optimiser step per batch, one validation pass per epoch; the epoch's ``val/loss`` is the
    sample-weighted mean over batches;
  * early stopping on ``val/loss``: stop after ``patience`` consecutive epochs without a new minimum;
  * test: macro accuracy and macro F1 over the classes that occur (torchmetrics' macro average
    ignores classes with no true and no predicted sample), confusion matrix rows = true class;
  * ``metrics.csv`` per run with Lightning's column names, ``confusion_matrix_test.csv`` after test.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.optim import NAdam

from .classifier import ClassifierModel
from .utils import split_decay_groups


def _confusion(true: torch.Tensor, pred: torch.Tensor, n: int) -> torch.Tensor:
    idx = true.long() * n + pred.long()
    return torch.bincount(idx, minlength=n * n).reshape(n, n)


def macro_scores(cm: torch.Tensor) -> Dict[str, float]:
    """Macro accuracy (= mean per-class recall) and macro F1 from a confusion matrix."""
    cm = cm.double()
    tp = cm.diag()
    fn = cm.sum(1) - tp
    fp = cm.sum(0) - tp
    present = (tp + fp + fn) > 0
    recall = torch.where(tp + fn > 0, tp / (tp + fn).clamp(min=1), torch.zeros_like(tp))
    f1 = torch.where(2 * tp + fp + fn > 0, 2 * tp / (2 * tp + fp + fn).clamp(min=1), torch.zeros_like(tp))
    k = max(int(present.sum()), 1)
    return {"accuracy": float(recall[present].sum() / k), "f1": float(f1[present].sum() / k)}


class ClassifierTrainer:
    def __init__(self, model: ClassifierModel, learning_rate: float = 0.0005, weight_decay: float = 0.0,
                 log_dir: Optional[str] = None, verbose: bool = False) -> None:
        self.model = model
        self.learning_rate = float(learning_rate)
        self.weight_decay = float(weight_decay)
        self.criterion = nn.CrossEntropyLoss()
        self.log_dir = log_dir
        self.verbose = verbose
        self.optimizer = self.configure_optimizers()
        self.history: List[Dict[str, float]] = []
        self.confusion_matrix: Optional[torch.Tensor] = None
        self.test_accuracy: Optional[float] = None
        self.test_f1: Optional[float] = None
        self.stopped_epoch: Optional[int] = None

    # ------------------------------------------------------------------ optimiser
    def configure_optimizers(self) -> NAdam:
        decay, no_decay = split_decay_groups(self.model.named_parameters())
        return NAdam([{"params": decay, "weight_decay": self.weight_decay},
                      {"params": no_decay, "weight_decay": 0.0}], lr=self.learning_rate)

    # ------------------------------------------------------------------ epochs
    def _run_epoch(self, loader, train: bool) -> Dict[str, float]:
        n_cls = self.model.n_classes
        loss_sum, n_seen = 0.0, 0
        cm = torch.zeros(n_cls, n_cls, dtype=torch.long)
        self.model.train(train)
        for x, y in loader:
            y = y.long()
            with torch.set_grad_enabled(train):
                logits = self.model(x)
                loss = self.criterion(logits, y)
            if train:
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
            loss_sum += float(loss.detach()) * len(y)
            n_seen += len(y)
            cm += _confusion(y.cpu(), logits.detach().argmax(1).cpu(), n_cls)
        return {"loss": loss_sum / max(n_seen, 1), "accuracy": macro_scores(cm)["accuracy"]}

    def fit(self, train_loader, val_loader, max_epochs: int, patience: int) -> List[Dict[str, float]]:
        best, wait, step = float("inf"), 0, 0
        for epoch in range(max_epochs):
            tr = self._run_epoch(train_loader, True)
            step += len(train_loader)
            va = self._run_epoch(val_loader, False)
            row = {"epoch": epoch, "step": step - 1, "train/loss_epoch": tr["loss"], "train/accuracy": tr["accuracy"],
                   "val/loss": va["loss"], "val/accuracy": va["accuracy"], "train/weight_norm": self._weight_norm()}
            self.history.append(row)
            if self.verbose:
                print(f"epoch {epoch}: train/loss {tr['loss']:.4f}  val/loss {va['loss']:.4f}  "
                      f"val/accuracy {va['accuracy']:.3f}")
            if va["loss"] < best:
                best, wait = va["loss"], 0
            else:
                wait += 1
                if wait >= patience:
                    self.stopped_epoch = epoch
                    break
        self._write_metrics()
        return self.history

    # ------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def test(self, loader) -> Dict[str, object]:
        n_cls = self.model.n_classes
        cm = torch.zeros(n_cls, n_cls, dtype=torch.long)
        self.model.eval()
        for x, y in loader:
            cm += _confusion(y.long().cpu(), self.model(x).argmax(1).cpu(), n_cls)
        scores = macro_scores(cm)
        self.test_accuracy, self.test_f1, self.confusion_matrix = scores["accuracy"], scores["f1"], cm
        if self.log_dir is not None:
            os.makedirs(self.log_dir, exist_ok=True)
            np.savetxt(os.path.join(self.log_dir, "confusion_matrix_test.csv"), cm.numpy(), fmt="%d", delimiter=",")
        return {"accuracy": self.test_accuracy, "f1": self.test_f1, "confusion_matrix": cm}

    @torch.no_grad()
    def predict(self, loader) -> torch.Tensor:
        self.model.eval()
        return torch.cat([self.model(x).argmax(dim=1) for x, _ in loader])

    # ------------------------------------------------------------------ helpers
    def get_nparams(self) -> int:
        return self.model.get_nparams()

    def get_layer_nparams(self) -> Dict[str, int]:
        return self.model.get_layer_nparams()

    def _weight_norm(self) -> float:
        return float(sum(float(p.detach().norm(2)) ** 2 for p in self.model.parameters() if p.requires_grad) ** 0.5)

    def _write_metrics(self) -> None:
        if self.log_dir is None or not self.history:
            return
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "metrics.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(self.history[0].keys()))
            w.writeheader()
            w.writerows(self.history)
