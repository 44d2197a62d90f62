"""Baseline-interval z-score (mirror of reference preprocess/signal/zscore_rereference.py:6-70):
mean / population std of ``rereference_interval`` (seconds) per channel, applied to the whole signal."""
from argparse import Namespace
from typing import Tuple

import torch

from ... import _lib
from ..._lib import check, ptr
from ._common import ret, stream, to_device


def rereference(data, reference_time: Tuple[float, float]):
    try:
        start, end = reference_time
    except ValueError:
        raise ValueError("reference_time must be a tuple of (start, end)")
    n_t = data.shape[1]
    if start < 0 or end > n_t:
        raise ValueError("Reference time indices are out of bounds.")
    if start >= end:
        raise ValueError("Start time must be less than end time.")
    x, was_np = to_device(data, "zscore_rereference")
    C, T = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(C, 2, dtype=torch.float64, device=x.device)
    check(_lib.load().tl_row_zscore(ptr(x), int(x.dtype == torch.float64), ptr(y), ptr(stats), C, T, int(start),
                                    int(end), 0, stream()), "tl_row_zscore")
    return ret(y, was_np)


def run(data, params: Namespace):
    if not hasattr(params, 'rereference_interval') or not hasattr(params, 'signal_freq'):
        raise ValueError("params must have 'rereference_interval' and 'signal_freq' attributes.")
    start, end = params.rereference_interval
    return rereference(data, (int(start * params.signal_freq), int(end * params.signal_freq)))
