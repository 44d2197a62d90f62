"""Step dispatcher of the preprocessing stage - the caller of the signal kernels (mirror of reference
preprocess/preprocessor.py:8-70).

Same contract as the reference: every step is ``{"module": dotted.path, "params": {...}}``; the step's
parameters are merged onto ONE shared ``Namespace`` (a key that is already there raises ``ValueError``),
the module is imported by name and its ``run(data, params)`` is called; steps may mutate the shared
Namespace (``downsample`` rewrites ``signal_freq`` before ``frequency_filter`` reads it).

MI355X addition: with ``resident=True`` a NumPy recording is moved to the GPU once, flows through
every step as a device tensor (each step of ``preprocess/signal`` accepts one) and comes back once -
instead of one PCIe round trip per step.  Module names of the reference layout
(``preprocess.signal.<step>`` / the stale ``preprocess.<step>`` of the sample YAML) resolve to this
package's kernels.

Multi-GPU (signal path): with ``shard_channels=True`` under an initialised process group every rank
carries rows [r C / N, (r + 1) C / N) of the recording through the channel-local steps (resampling, band extraction,
z-scores - the expensive ones) with no communication at all; only in front of a step that mixes channels
(``car_rereference``: the mean over channels at every sample) are the shards all-gathered, the step run on the whole
array by every rank, and the rank's rows cut out again.  The result (all-gathered once more at the end) is the same
array on every rank, produced by the same kernels on the same numbers as a single process produces it.
"""
from __future__ import annotations

import importlib
import os
from argparse import Namespace
from copy import deepcopy
from typing import Dict, List, Optional, Tuple

import numpy as np

_PKG = __name__.rsplit(".", 1)[0]                      # decode_tonal_langauge_amd.preprocess
_STEPS = ("downsample", "frequency_filter", "channel_zscore", "car_rereference", "rolling_zscore",
          "zscore_rereference")


def resolve_step_module(name: str):
    """Import a step module by the name a config gives it.  ``preprocess.signal.X`` and ``preprocess.X``
    (reference example_config.yaml:27,31,43) map onto this package; anything else is imported as is."""
    tail = name.rsplit(".", 1)[-1]
    if tail in _STEPS and (name.startswith("preprocess.") or name == tail):
        name = f"{_PKG}.signal.{tail}"
    return importlib.import_module(name)


# Steps of this package whose output row r depends on input row r only (they may stack band entries along the rows): these
# run on a rank's shard with no communication.  Everything else - car_rereference (the mean over channels at every
# sample) and ANY module this package does not know (the plugin ABI lets a pipeline YAML name arbitrary modules: a user
# step may mix rows or select channels) - runs on the gathered, whole array on every rank and is cut again.  A foreign
# module can opt in by defining ``CHANNEL_LOCAL = True`` at module level.
_CHANNEL_LOCAL = ("downsample", "frequency_filter", "channel_zscore", "rolling_zscore", "zscore_rereference")


def _is_channel_local(module, module_name: str) -> bool:
    flag = getattr(module, "CHANNEL_LOCAL", None)
    if flag is not None:
        return bool(flag)
    return module.__name__.startswith(_PKG + ".signal.") and module_name.rsplit(".", 1)[-1] in _CHANNEL_LOCAL


class _ChannelShards:
    """Bookkeeping of a recording split by rows over the ranks.  A band-extraction step with E entries turns (C, T) into
    (E C, T), entry-major (reference frequency_filter.py:74-77): a shard then holds E groups of its own rows, and the
    whole array is groups x (all ranks' rows)."""

    def __init__(self, n_channels: int):
        from .. import parallel
        self.parallel = parallel
        self.rank, self.world = parallel.world()
        self.C = n_channels
        self.per = -(-n_channels // self.world)                      # equal blocks for the all-gather; the last is padded
        self.lo = min(self.rank * self.per, n_channels)
        self.hi = min(self.lo + self.per, n_channels)
        self.groups = 1

    def cut(self, full):
        """This rank's rows of every group, zero-padded to ``per`` rows per group."""
        import torch
        g = full.reshape(self.groups, self.C, full.shape[-1])
        out = torch.zeros(self.groups, self.per, full.shape[-1], dtype=full.dtype, device=full.device)
        out[:, : self.hi - self.lo] = g[:, self.lo:self.hi]
        if self.hi - self.lo < self.per:
            # padding rows (dropped on gather): a copy of the shard's last real row (or of the recording's last row when the
            # shard is empty), so that per-row statistics of the channel-local steps stay finite (a constant row is 0 / 0
            # in the z-score kernels)
            fill = g[:, self.hi - 1:self.hi] if self.hi > self.lo else g[:, self.C - 1:self.C]
            out[:, self.hi - self.lo:] = fill
        return out.reshape(self.groups * self.per, full.shape[-1])

    def gather(self, local):
        """(groups x per, T) on every rank -> (groups x C, T), the single-process row order."""
        import torch
        T = local.shape[-1]
        out = torch.empty(self.world, self.groups * self.per, T, dtype=local.dtype, device=local.device)
        self.parallel.all_gather_blocks(out, local)
        full = out.reshape(self.world, self.groups, self.per, T).permute(1, 0, 2, 3).reshape(self.groups, self.world * self.per, T)
        return full[:, : self.C].reshape(self.groups * self.C, T).contiguous()


def preprocess_signal(data, steps: List[Dict], block_params: Namespace, figure_dir: Optional[str] = None,
                      num_channels: int = 5, duration: float = 1.0, resident: bool = False, shard_channels: bool = False):
    """Apply the steps in order (reference :39-70).  Returns ``(data, block_params.signal_freq)``."""
    was_np = isinstance(data, np.ndarray)
    shards = None
    if shard_channels:
        from .. import parallel
        if parallel.world()[1] > 1 and getattr(data, "ndim", 0) == 2:
            import torch
            if was_np:
                data = torch.from_numpy(np.ascontiguousarray(data))
                data = data.to("cuda") if torch.cuda.is_available() else data
            shards = _ChannelShards(data.shape[0])
            data = shards.cut(data)
            if figure_dir is not None:
                raise ValueError("preprocess_signal: per-step figures need the whole recording; not available with shard_channels")
    if resident and was_np and shards is None:
        import torch
        data = torch.from_numpy(np.ascontiguousarray(data)).to("cuda")
    for i, step in enumerate(steps):
        module_name = step['module']
        for key, value in step.get('params', {}).items():
            if hasattr(block_params, key):
                raise ValueError(f"Parameter '{key}' already exists in params. Please ensure no conflicting "
                                 "parameter names in each preprocessing step.")
            setattr(block_params, key, value)
        before_freq = block_params.signal_freq
        before = data if figure_dir is None else (data.copy() if isinstance(data, np.ndarray) else data.clone())
        module = resolve_step_module(module_name)
        if shards is not None and not _is_channel_local(module, module_name):
            whole = module.run(shards.gather(data), block_params)
            if whole.shape[0] % (shards.groups * shards.C) == 0:
                shards.groups *= whole.shape[0] // (shards.groups * shards.C)
            else:                                                      # the step changed the channel count: shard anew
                shards = _ChannelShards(whole.shape[0])
            data = shards.cut(whole)
        else:
            rows = data.shape[0]
            data = module.run(data, block_params)
            if shards is not None and data.shape[0] != rows:           # band entries stacked along the rows
                if data.shape[0] % rows:
                    raise ValueError(f"preprocess_signal(shard_channels=True): step '{module_name}' turned {rows} rows into "
                                     f"{data.shape[0]} - not a whole number of entries per channel; a channel-local step must "
                                     "keep or multiply the rows")
                shards.groups *= data.shape[0] // rows
        if figure_dir and getattr(data, "ndim", 0) == 2:
            _plot_step(before, before_freq, data, block_params.signal_freq, figure_dir, i, module_name,
                       num_channels, duration)
    if shards is not None:
        data = shards.gather(data)
        if was_np:
            data = data.cpu().numpy()
    elif resident and was_np:
        data = data.cpu().numpy()
    return data, block_params.signal_freq


def preprocess_modalities(data_dict: Dict, modalities_cfg: Dict, base_params: Namespace,
                          figure_dir: Optional[str] = None, resident: bool = False, shard_channels: bool = False) -> Dict:
    """Per modality: copy the base parameters, take ``signal_freq`` from ``<modality>_sf``, run the
    configured steps, write the data and the new sampling rate back (reference :8-36)."""
    for modality, cfg in modalities_cfg.items():
        mod_type = cfg.get("type")
        mod_fig_dir = os.path.join(figure_dir, modality) if figure_dir else None
        if mod_fig_dir:
            os.makedirs(mod_fig_dir, exist_ok=True)
        if mod_type is None:
            raise KeyError(f"Modality '{modality}' missing 'type' field in config")
        steps = cfg.get("preprocessing", {}).get("steps", [])
        if not steps:
            continue
        params = deepcopy(base_params)
        if mod_type != "signal":
            raise ValueError(f"Modality '{modality}': unsupported type '{mod_type}' (only 'signal' is preprocessed)")
        params.signal_freq = data_dict.get(f"{modality}_sf")
        processed, freq = preprocess_signal(data_dict[modality], steps, params, figure_dir=mod_fig_dir,
                                            resident=resident, shard_channels=shard_channels)
        if freq is not None:
            data_dict[f"{modality}_sf"] = freq
        data_dict[modality] = processed
    return data_dict


def _plot_step(before, before_freq, after, after_freq, figure_dir, index, module_name, num_channels, duration) -> None:
    """First ``duration`` seconds of the first channels before / after a step (reference :74-137 draws the
    same comparison; plotting is outside the hot path and only runs when a figure directory is given)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    to_np = lambda a: a if isinstance(a, np.ndarray) else a.detach().cpu().numpy()
    before, after = to_np(before), to_np(after)
    n = min(num_channels, before.shape[0], after.shape[0])
    fig, axes = plt.subplots(n, 2, figsize=(10, 2 * n), squeeze=False)
    for c in range(n):
        for ax, arr, fs, title in ((axes[c][0], before, before_freq, "before"), (axes[c][1], after, after_freq, "after")):
            m = max(1, min(arr.shape[1], int(duration * fs)))
            ax.plot(np.arange(m) / fs, arr[c, :m], linewidth=0.6)
            if c == 0:
                ax.set_title(f"{title} ({fs} Hz)")
    fig.tight_layout()
    fig.savefig(os.path.join(figure_dir, f"step_{index}_{module_name.rsplit('.', 1)[-1]}.png"), dpi=100)
    plt.close(fig)
