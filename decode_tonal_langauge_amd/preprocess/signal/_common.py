"""Shared host plumbing of the signal steps: NumPy in -> device -> NumPy out, or CUDA tensor in/out."""
import numpy as np
import torch

from ... import _lib


def to_device(data, what: str):
    if isinstance(data, torch.Tensor):
        _lib.require_gpu(data, what)
        t, was_np = data, False
    else:
        if not torch.cuda.is_available():
            raise RuntimeError(f"{what} (MI355X build): no GPU visible; this package has no CPU fallback")
        arr = np.asarray(data)
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        t, was_np = torch.from_numpy(np.ascontiguousarray(arr)).to(torch.device("cuda", torch.cuda.current_device())), True
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    if t.dim() != 2:
        raise ValueError("expected data of shape (n_channels, n_timepoints)")
    return t.contiguous(), was_np


def ret(t, was_np):
    return t.cpu().numpy() if was_np else t


def stream():
    return torch.cuda.current_stream().cuda_stream
