"""YAML config helpers (mirror of reference utils/config.py:8-84)."""
import hashlib
import json
import os
from argparse import Namespace

import yaml


def load_config(path: str) -> dict:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def dict_to_namespace(d, exclude_keys=None):
    if exclude_keys is None:
        exclude_keys = set()
    if isinstance(d, dict):
        return Namespace(**{k: dict_to_namespace(v) if k not in exclude_keys else v for k, v in d.items()})
    if isinstance(d, list):
        return [dict_to_namespace(v) for v in d]
    return d


def update_configuration(output_path: str, previous_config_path: str, new_module: str, new_module_cfg: dict) -> None:
    if os.path.exists(previous_config_path):
        previous_cfg = load_config(previous_config_path)
    else:
        previous_cfg = {}
        print(f"Warning: config.yaml not found in {previous_config_path}")
    previous_cfg[new_module] = new_module_cfg
    with open(output_path, "w") as f:
        yaml.dump(previous_cfg, f)


def generate_hash_name_from_config(base_name: str, config: dict) -> str:
    digest = hashlib.md5(json.dumps(config, sort_keys=True).encode()).hexdigest()[:6]
    return f"{base_name}__{digest}"
