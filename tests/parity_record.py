"""Observed parity deviations of the GPU tests, kept as a record - TEST INFRASTRUCTURE.

The golden tests assert bounds; what they actually OBSERVE on the MI355X (worst deviation per quantity, per Winograd form)
goes to ``gpurun_out/parity_observed.json`` (merged back from the GPU box by gpurun); the copy judged is
``profiles/parity_observed.json``.  Failing to write is never an error: the assertions are the test."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity_observed.json")


def record(section: str, values: dict) -> None:
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        vals = {k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "__float__") else v) for k, v in values.items()}
        vals["_recorded_unix"] = int(time.time())
        data[section] = vals
        tmp = PATH + ".tmp"
        with open(tmp, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
        os.replace(tmp, PATH)
    except (OSError, ValueError, TypeError):
        pass
