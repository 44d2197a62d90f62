"""Rolling z-score (mirror of reference preprocess/signal/rolling_zscore.py:5-51): pandas
``rolling(window, min_periods=1)`` mean and sample std (ddof=1) per channel; float64 out."""
import torch

from ... import _lib
from ..._lib import check, ptr
from ._common import ret, stream, to_device


def run(data, params: object):
    window_length = getattr(params, "window_length", 10)
    window_size = int(window_length * params.signal_freq)
    preserve_nans = getattr(params, "preserve_nans", True)
    if window_size <= 1:
        raise ValueError("window_size must be greater than 1.")
    x, was_np = to_device(data, "rolling_zscore")
    C, T = x.shape
    y = torch.empty(C, T, dtype=torch.float64, device=x.device)
    check(_lib.load().tl_rolling_zscore(ptr(x), int(x.dtype == torch.float64), ptr(y), C, T, window_size,
                                        int(not preserve_nans), stream()), "tl_rolling_zscore")
    return ret(y, was_np)
