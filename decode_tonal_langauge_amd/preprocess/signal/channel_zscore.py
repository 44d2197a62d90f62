"""Per-channel z-score (mirror of reference preprocess/signal/channel_zscore.py:5-29): population
std over the whole recording; ``preserve_nans=False`` zero-fills NaNs."""
from argparse import Namespace

import torch

from ... import _lib
from ..._lib import check, ptr
from ._common import ret, stream, to_device


def run(data, params: Namespace):
    preserve_nans = getattr(params, "preserve_nans", True)
    x, was_np = to_device(data, "channel_zscore")
    C, T = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(C, 2, dtype=torch.float64, device=x.device)
    check(_lib.load().tl_row_zscore(ptr(x), int(x.dtype == torch.float64), ptr(y), ptr(stats), C, T, 0, T,
                                    int(not preserve_nans), stream()), "tl_row_zscore")
    return ret(y, was_np)
