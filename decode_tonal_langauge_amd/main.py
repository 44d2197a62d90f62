"""YAML stage runner (counterpart of reference main.py:8-72).

For every stage of the fixed order below whose config names a ``module``, import it and call
``getattr(module, cfg.get("function", "run"))(config)``.  A ``str`` return value is remembered as
that stage's output directory and offered to later stages through their ``params.io`` block
(only where the user has not set the key), exactly as the reference wires
preprocess -> sample_collection -> channel_selection -> training.
"""
import importlib
import sys
from typing import Any, Dict

from .utils.config import load_config

STAGES = ["preprocess", "sample_collection", "channel_selection", "training", "evaluation", "visualisation"]

#: stage -> ((io key to fill, producing stage), ...)
_IO_WIRING = {
    "sample_collection": (("recording_dir", "preprocess"),),
    "channel_selection": (("sample_dir", "sample_collection"),),
    "training": (("sample_dir", "sample_collection"), ("channel_selection_dir", "channel_selection")),
}


def update_stage_cfg_io(stage_outputs: dict, stage: str, stage_cfg: dict):
    wiring = _IO_WIRING.get(stage)
    if not wiring:
        return
    io_cfg = stage_cfg.setdefault("params", {}).setdefault("io", {})
    for key, producer in wiring:
        if key not in io_cfg and producer in stage_outputs:
            io_cfg[key] = stage_outputs[producer]


def _resolve(module_name: str, func_name: str):
    # reference configs name the stage modules bare ("train_classifier", "train_synthesizer"): those
    # resolve to this package's modules of the same name
    pkg = __name__.rsplit(".", 1)[0] if "." in __name__ else None
    module = None
    if pkg and "." not in module_name:
        try:
            module = importlib.import_module(f"{pkg}.{module_name}")
        except ModuleNotFoundError as e:
            if e.name != f"{pkg}.{module_name}":
                raise
    if module is None:
        module = importlib.import_module(module_name)
    if not hasattr(module, func_name):
        raise ImportError(f"Module '{module_name}' does not have a function '{func_name}'"
                          f"Available functions: {', '.join(dir(module))}")
    return getattr(module, func_name)


def run_pipeline(config_path: str) -> None:
    config: Dict[str, Any] = load_config(config_path)
    outputs: Dict[str, str] = {}
    for stage in STAGES:
        cfg = config.get(stage)
        if not cfg or cfg.get("module") is None:
            continue
        print('----------- Running stage:', stage, '-----------')
        update_stage_cfg_io(outputs, stage, cfg)
        config[stage] = cfg
        result = _resolve(cfg["module"], cfg.get("function", "run"))(config)
        if isinstance(result, str):
            outputs[stage] = result


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit("Usage: python -m decode_tonal_langauge_amd.main <config.yaml>")
    run_pipeline(sys.argv[1])
