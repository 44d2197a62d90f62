"""GPU: the time-parallel filtfilt (``TONAL_KERNELS=butter=scan``, tl_filtfilt_scan_f64) against the sequential kernel
(bit-identical to scipy's loop) and the reference goldens.  Bound 2e-7 of the largest output sample: what the scan observes is
2e-8 - 5e-8, the size of the reference's own rounding (tests/test_signal_scan_notes.py) - which is why the form is opt-in."""
import os

import numpy as np
import pytest
import torch

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-7


def _both(x, freqs, fs, **kw):
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    seq = ff.butter_filter(x, freqs, fs, **kw)
    os.environ["TONAL_BUTTER"] = "scan"                  # (A/B switch; the product setting is TONAL_KERNELS=butter=scan)
    try:
        scan = ff.butter_filter(x, freqs, fs, **kw)
    finally:
        os.environ.pop("TONAL_BUTTER", None)
    return np.asarray(seq), np.asarray(scan)


def _dev(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_scan_filtfilt_against_golden_and_sequential_kernel():
    g = np.load(os.path.join(GOLD, "g6_signal.npz"))
    x, _x2 = gi.g6_inputs()
    seq, scan = _both(x, [0.3, 100], 400)
    assert _dev(seq, g["butter"]) < 1e-9                 # the default is untouched
    assert scan.dtype == np.float64 and scan.shape == seq.shape
    assert _dev(scan, g["butter"]) < TOL and _dev(scan, seq) < TOL
    assert not np.array_equal(scan, seq)                 # (really the other kernel)


@pytest.mark.parametrize("C,T,dtype", [(256, 24000, np.float32), (70, 24001, np.float64), (3, 60, np.float64), (1, 300007, np.float64),
                                       (65, 129, np.float32)])
def test_scan_filtfilt_sizes(C, T, dtype):
    """the C5 size, ragged channel counts, a recording shorter than one block, one that spans several 512-block chunks of the
    scan, and block edges (T + 2 * 27 = k * 128 +- 1)"""
    x = np.random.default_rng(C * 7 + T).standard_normal((C, T)).astype(dtype)
    seq, scan = _both(x, [0.3, 100], 400)
    d = _dev(scan, seq)
    assert np.isfinite(scan).all() and d < TOL, d


@pytest.mark.parametrize("freqs,order,ftype", [(30.0, 2, "lowpass"), ([70.0, 150.0], 2, "bandpass"), (1.0, 4, "highpass")])
def test_scan_filtfilt_other_designs(freqs, order, ftype):
    x = np.random.default_rng(11).standard_normal((5, 5000))
    seq, scan = _both(x, freqs, 400, order=order, filter_type=ftype)
    assert _dev(scan, seq) < TOL


def test_scan_filtfilt_device_tensor_and_property():
    """linearity (odd extension, zi scaling and the recurrence are all linear in the recording) at the C5 size, on a resident
    device tensor"""
    from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    a = torch.randn(256, 24000, device=dev, generator=g, dtype=torch.float64)
    b = torch.randn(256, 24000, device=dev, generator=g, dtype=torch.float64)
    os.environ["TONAL_BUTTER"] = "scan"
    try:
        fa, fb = ff.butter_filter(a, [0.3, 100], 400), ff.butter_filter(b, [0.3, 100], 400)
        fab = ff.butter_filter(2.0 * a - 3.0 * b, [0.3, 100], 400)
    finally:
        os.environ.pop("TONAL_BUTTER", None)
    assert isinstance(fa, torch.Tensor) and fa.is_cuda
    scale = float(fab.abs().max())
    assert float((fab - (2.0 * fa - 3.0 * fb)).abs().max()) < 1e-6 * scale
