"""Build a classifier from its dotted class path (counterpart of reference models/classifier_factory.py:10-57).

The constructor receives only the keyword arguments its signature names, drawn from
``n_classes, n_channels, seq_length, input_channels, input_length, input_dim`` and the user's
``classifier_kwargs`` (which win).  Reference paths such as ``models.simple_classifiers.X`` resolve
to this package's classes of the same name."""
from __future__ import annotations

import inspect
from importlib import import_module
from typing import Dict, Optional

from .classifier import ClassifierModel

_PKG = __name__.rsplit(".", 2)[0]          # decode_tonal_langauge_amd


def _import(module_name: str):
    if module_name.startswith("models.") or module_name == "models":
        try:
            return import_module(f"{_PKG}.{module_name}")
        except ImportError:
            pass
    return import_module(module_name)


def get_classifier_by_name(model_path: str, device: str, n_classes: int, n_channels: int, seq_length: int,
                           classifier_kwargs: Optional[Dict] = None) -> ClassifierModel:
    module_name, class_name = model_path.rsplit(".", 1)
    cls = getattr(_import(module_name), class_name)
    offered = {"n_classes": n_classes, "n_channels": n_channels, "seq_length": seq_length,
               "input_channels": n_channels, "input_length": seq_length, "input_dim": n_channels * seq_length}
    offered.update(classifier_kwargs or {})
    accepted = inspect.signature(cls).parameters
    return cls(**{k: v for k, v in offered.items() if k in accepted}).to(device)
