"""Common average re-reference (mirror of reference preprocess/signal/car_rereference.py:5-41)."""
from argparse import Namespace

import numpy as np
import torch

from ... import _lib
from ..._lib import check, ptr
from ._common import ret, stream, to_device


def run(data, params: Namespace):
    if not hasattr(params, 'exclude_channels'):
        params.exclude_channels = []
    exclude = params.exclude_channels
    if not isinstance(exclude, list):
        raise ValueError("exclude_channels must be a list of integers.")
    n_ch = data.shape[0]
    if any(ch < 0 or ch >= n_ch for ch in exclude):
        raise ValueError("exclude_channels contains invalid channel indices.")
    x, was_np = to_device(data, "car_rereference")
    C, T = x.shape
    inc = np.ones(C, dtype=np.int32)
    inc[exclude] = 0
    n_inc = int(inc.sum())
    if n_inc == 0:
        return ret(torch.full_like(x, float("nan")), was_np)      # numpy: mean of an empty slice is NaN
    incd = torch.from_numpy(inc).to(x.device)
    y = torch.empty_like(x)
    check(_lib.load().tl_car(ptr(x), int(x.dtype == torch.float64), ptr(incd), ptr(y), C, T, n_inc, stream()), "tl_car")
    return ret(y, was_np)
