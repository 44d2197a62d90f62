"""ONE validated setting for every kernel-selection switch of the package: ``TONAL_KERNELS``.

    TONAL_KERNELS="wino=4,fuse_c1=0"  python train_synthesizer.py ...

A comma-separated list of ``key=value`` pairs.  Unknown keys and values that a key does not take raise ``ValueError`` where
the setting is read - engine construction - instead of silently selecting something (round 4's review: 26 independent
``TONAL_*`` variables read in product code, any of which could quietly put a round-1 kernel on the hot path).  Nothing set =
the defaults = the kernels ``bench.py`` times.

The per-switch variables of rounds 1-4 (``TONAL_WINO``, ``TONAL_WINO_V``, ...) still exist for the A/B tests and the timing
scripts, which flip one switch at a time - but ONLY under ``TONAL_AB=1`` (tests/conftest.py sets it): a legacy variable found
in the environment without it raises, naming the ``TONAL_KERNELS`` key that replaces it.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Tuple, Union

_Allowed = Union[frozenset, Callable[[str], bool]]


def _int(lo: int, hi: int) -> Callable[[str], bool]:
    def ok(v: str) -> bool:
        try:
            return lo <= int(v) <= hi
        except ValueError:
            return False
    return ok


_B = frozenset({"0", "1"})
#: key -> (legacy variable, allowed values, default, what it selects)
KEYS: Dict[str, Tuple[str, _Allowed, str, str]] = {
    # ---- SynthesisModelCNN conv stack (_cnn_engine.py)
    "wino": ("TONAL_WINO", frozenset({"6", "4", "0"}), "6",
             "conv2 / conv3: 6 Winograd F(6,3) on pre-transformed operands (F(4,3) V form where the stack does not allow it), "
             "4 the F(4,3) V form, 0 direct MFMA kernels (also the fallback for every shape neither form covers)"),
    "f63_yprod": ("TONAL_F63_YPROD", _B, "1", "F(6,3): the backward operands Y / Vd of stages 2 and 3 come pre-transformed out of the "
                                              "input-gradient epilogue of the stage above (0: the weight-gradient kernels un-pool and "
                                              "transform the gradient rows themselves)"),
    "conv4_dgrad": ("TONAL_CONV4_DGRAD", frozenset({"nt63", "gemm"}), "nt63",
                    "F(6,3): the one-tap stage behind conv3 - nt63: its input gradient on the NT63 kernel, writing Y3 / Vd3 "
                    "(no gradient rows, no tl_wino63_unpool_yvd); gemm: one-tap GEMM + tl_wino63_unpool_yvd"),
    "store_p1": ("TONAL_STORE_P1", _B, "0", "keep the raw pooled rows of stages 1 / 2 beside V (tests)"),
    # ---- deep classifiers (_classifier_engine.py)
    "conv7": ("TONAL_CONV7", frozenset({"wino63", "direct"}), "wino63", "the CNN-RNN classifier's 7-tap convolutions"),
    # ---- optimiser / trainer
    "whh_dh": ("TONAL_WHH_DH", _B, "1", "single process: the W_hh NAdam pass also produces the last BPTT product dh_1 = dgates_2 . W_hh "
                                        "(0: a W_hh stream of its own for it)"),
    "whh_stream": ("TONAL_WHH_STREAM", _B, "1", "label LSTM backward: dgates . W_hh as a pure stream of the weight (tl_lstm_gw; 0: the skinny "
                                                "MFMA GEMM with split-K slabs, also the form for more than 8 distinct label rows)"),
    "graph": ("TONAL_GRAPH", _B, "1", "SynthesisLite step replayed as a HIP graph"),
    "lstm_shard": ("TONAL_LSTM_SHARD", _B, "1", "data parallel: label LSTM sharded by gate rows"),
    # ---- preprocess/signal
    "hilbert": ("TONAL_HILBERT", frozenset({"auto", "ols", "ols_full", "sym", "taps", "fft"}), "auto",
                "Hilbert bank: force a path (ols band-limited overlap-save, ols_full all 1024 bins per band, sym / taps the "
                "time-domain kernels with / without Hermitian symmetry, fft the DFT-domain form)"),
    "hilbert_f32": ("TONAL_HILBERT_F32", _B, "0", "float32 recordings: fp32 transforms end to end (default: fp64 math)"),
    "butter": ("TONAL_BUTTER", frozenset({"seq", "scan"}), "seq",
               "zero-phase Butterworth: seq the sequential recurrence (bit-identical to scipy's loop), scan the time-parallel "
               "block scan (2e-8 - 5e-8 from it: the size of the reference's own rounding; ~20 x faster)"),
}
_LEGACY = {v[0]: k for k, v in KEYS.items()}
_cache: Tuple[str, Dict[str, str]] = ("", {})


def _check(key: str, value: str) -> str:
    allowed = KEYS[key][1]
    ok = allowed(value) if callable(allowed) else value in allowed
    if not ok:
        what = "an integer in range" if callable(allowed) else "one of " + ", ".join(sorted(allowed))
        raise ValueError(f"TONAL_KERNELS: {key}={value!r} is not a value this switch takes ({what})")
    return value


def _parsed() -> Dict[str, str]:
    global _cache
    raw = os.environ.get("TONAL_KERNELS", "")
    if raw == _cache[0]:
        return _cache[1]
    out: Dict[str, str] = {}
    for item in raw.split(","):
        item = item.strip()
        if not item:
            continue
        key, sep, value = item.partition("=")
        key, value = key.strip(), value.strip()
        if not sep or key not in KEYS:
            raise ValueError(f"TONAL_KERNELS: unknown setting {item!r}; keys: " + ", ".join(sorted(KEYS)))
        out[key] = _check(key, value)
    _cache = (raw, out)
    return out


def get(key: str, default: str = None) -> str:
    """The value of switch ``key``: from ``TONAL_KERNELS``, else (under ``TONAL_AB=1`` only) from its legacy variable, else
    ``default`` (or the table's default)."""
    legacy, _allowed, table_default, _doc = KEYS[key]
    spec = _parsed()
    if key in spec:
        return spec[key]
    if legacy in os.environ:
        if os.environ.get("TONAL_AB") != "1":
            raise RuntimeError(f"{legacy} is an A/B switch of the test suite and is ignored by the product: use "
                               f"TONAL_KERNELS={key}={os.environ[legacy]} (or set TONAL_AB=1 to run an A/B comparison)")
        return _check(key, os.environ[legacy])
    return table_default if default is None else default


def validate() -> None:
    """Raise for an unknown key / value in ``TONAL_KERNELS`` or a legacy switch without ``TONAL_AB=1`` (engine construction)."""
    _parsed()
    if os.environ.get("TONAL_AB") != "1":
        for legacy, key in _LEGACY.items():
            if legacy in os.environ:
                get(key)
