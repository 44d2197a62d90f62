"""YAML stage runner (mirror of reference main.py:8-72).

Walks the fixed stage list, imports ``stage_cfg["module"]`` and calls
``getattr(module, stage_cfg.get("function", "run"))(config)``; a ``str`` result is remembered as
that stage's output directory and injected into the next stage's ``params.io``."""
import importlib
from typing import Any, Dict

from .utils.config import load_config

STAGES = ["preprocess", "sample_collection", "channel_selection", "training", "evaluation", "visualisation"]


def update_stage_cfg_io(stage_outputs: dict, stage: str, stage_cfg: dict):
    if stage == "sample_collection":
        io = stage_cfg.setdefault("params", {}).setdefault("io", {})
        if "recording_dir" not in io and "preprocess" in stage_outputs:
            io["recording_dir"] = stage_outputs["preprocess"]
    elif stage == "channel_selection":
        io = stage_cfg.setdefault("params", {}).setdefault("io", {})
        if "sample_dir" not in io and "sample_collection" in stage_outputs:
            io["sample_dir"] = stage_outputs["sample_collection"]
    elif stage == "training":
        io = stage_cfg.setdefault("params", {}).setdefault("io", {})
        if "sample_dir" not in io and "sample_collection" in stage_outputs:
            io["sample_dir"] = stage_outputs["sample_collection"]
        if "channel_selection_dir" not in io and "channel_selection" in stage_outputs:
            io["channel_selection_dir"] = stage_outputs["channel_selection"]


def run_pipeline(config_path: str) -> None:
    config: Dict[str, Any] = load_config(config_path)
    stage_outputs: Dict[str, str] = {}
    for stage in STAGES:
        stage_cfg = config.get(stage)
        if not stage_cfg:
            continue
        module_name = stage_cfg.get("module")
        func_name = stage_cfg.get("function", "run")
        if module_name is None:
            continue
        print('----------- Running stage:', stage, '-----------')
        update_stage_cfg_io(stage_outputs, stage, stage_cfg)
        config[stage] = stage_cfg
        module = importlib.import_module(module_name)
        try:
            func = getattr(module, func_name)
        except AttributeError:
            raise ImportError(f"Module '{module_name}' does not have a function '{func_name}'"
                              f"Available functions: {', '.join(dir(module))}")
        result = func(config)
        if isinstance(result, str):
            stage_outputs[stage] = result


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 2:
        raise SystemExit("Usage: python -m decode_tonal_langauge_amd.main <config.yaml>")
    run_pipeline(sys.argv[1])
