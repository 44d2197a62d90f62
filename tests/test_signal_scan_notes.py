"""Why the time-parallel filtfilt (TONAL_KERNELS=butter=scan) is opt-in and held to 2e-7, not 1e-9 - CPU, no GPU.

The reference filters with the (b, a) form of ``butter(4, [0.3, 100] / nyq, 'bandpass')`` (reference
preprocess/signal/frequency_filter.py:187-229): eight poles clustered at z = 1, direct-form states ~1e6 x the output.  scipy's
result is one particular rounding trajectory of that recurrence:

* the SAME loop (same operations, same order) in 64-bit-mantissa arithmetic differs from scipy's double loop by 1e-8 - 3e-8 -
  so the reference itself is only defined to ~3e-8, and no re-association of the arithmetic can promise 1e-9;
* the block scan with exactly propagated block-start states (the algorithm of tl_filtfilt_scan_f64, restated in numpy below)
  lands 2e-8 - 5e-8 from scipy - the same size; the same scan with plain fp64 matrix products lands orders of magnitude
  further away (round 2's segment-parallel kernel: 6e-3).
"""
from fractions import Fraction

import numpy as np
import scipy.signal as ss

from decode_tonal_langauge_amd.preprocess.signal.frequency_filter import _scan_matrices

FS = 400.0
B_, A_ = ss.butter(4, [0.3 / (FS / 2), 100.0 / (FS / 2)], btype="bandpass")


def _lfilter_loop(b, a, x, zi, dt):
    """scipy's lfilter loop (direct form II transposed), same operation order, in dtype ``dt``."""
    b = [dt(v) for v in b]
    a = [dt(v) for v in a]
    z = [dt(v) for v in zi]
    n = len(b)
    y = np.empty(len(x), dt)
    for i in range(len(x)):
        xv = x[i]
        yv = z[0] + b[0] * xv
        for k in range(n - 2):
            z[k] = (z[k + 1] + xv * b[k + 1]) - yv * a[k + 1]
        z[n - 2] = xv * b[n - 1] - yv * a[n - 1]
        y[i] = yv
    return y


def _filtfilt_loop(x, dt, lfilt=_lfilter_loop):
    edge = 3 * max(len(A_), len(B_))
    zi = ss.lfilter_zi(B_, A_)
    r = x.astype(dt)
    ext = np.concatenate([2 * r[0] - r[edge:0:-1], r, 2 * r[-1] - r[-2:-edge - 2:-1]])
    y1 = lfilt(B_, A_, ext, zi * ext[0], dt)
    y2 = lfilt(B_, A_, y1[::-1].copy(), zi * y1[-1], dt)
    return y2[::-1][edge:-edge]


def test_the_reference_result_is_a_rounding_trajectory():
    x = np.random.default_rng(0).standard_normal(2000)
    ref = ss.filtfilt(B_, A_, x)
    assert np.array_equal(_filtfilt_loop(x, np.float64), ref)            # the restated loop IS scipy's, bit for bit
    if np.finfo(np.longdouble).nmant <= 52:
        return                                                           # no extended precision on this host
    wide = _filtfilt_loop(x, np.longdouble).astype(np.float64)
    dev = float(np.abs(wide - ref).max() / np.abs(ref).max())
    assert 2e-9 < dev < 2e-7, dev            # observed 1.1e-8 (T = 1000 .. 4000), 2.4e-8 at T = 24 000


def test_scan_matrices_are_the_exact_powers_rounded_once():
    L = 128
    M = _scan_matrices(A_, L, 3)
    A = [[Fraction(0)] * 8 for _ in range(8)]
    for k in range(8):
        A[k][0] -= Fraction(float(A_[k + 1]))
        if k + 1 < 8:
            A[k][k + 1] += 1
    mm = lambda X, Y: [[sum(X[i][k] * Y[k][j] for k in range(8)) for j in range(8)] for i in range(8)]
    P = [[Fraction(int(i == j)) for j in range(8)] for i in range(8)]
    base, e = A, L
    while e:
        if e & 1:
            P = mm(P, base)
        base = mm(base, base)
        e >>= 1
    for m in range(3):
        scale = max(abs(float(v)) for r in P for v in r)
        err = max(abs(Fraction(float(M[m, i, j, 0])) + Fraction(float(M[m, i, j, 1])) - P[i][j]) for i in range(8) for j in range(8))
        assert float(err) < 1e-30 * scale, (m, float(err), scale)
        P = mm(P, P)


def _scan_lfilter(b, a, x, zinit, L, M, compensated=True):
    """numpy restatement of the three device steps for one channel."""
    n, ns = len(b), 8
    N = len(x)
    nb = (N + L - 1) // L
    xp = np.concatenate([x, np.zeros(nb * L - N)]).reshape(nb, L)
    bp, ap = np.zeros(9), np.zeros(9)
    bp[:n], ap[:n] = b, a

    def run(z0):
        z = np.concatenate([z0, np.zeros((nb, 1))], axis=1)
        y = np.empty_like(xp)
        for t in range(L):
            xv = xp[:, t]
            yv = z[:, 0] + bp[0] * xv
            for k in range(ns):
                z[:, k] = (z[:, k + 1] + xv * bp[k + 1]) - yv * ap[k + 1]
            y[:, t] = yv
        return y, z[:, :ns]

    _, s = run(np.zeros((nb, ns)))
    zs = np.empty((nb, ns))
    z = np.zeros(ns)
    z[:len(zinit)] = zinit
    for j in range(nb):                                   # (the device scans; the recurrence it evaluates is this one)
        zs[j] = z
        if compensated:                                   # exact sum, rounded once: what the compensated products deliver
            z = np.array([float(Fraction(float(s[j, i])) + sum((Fraction(float(M[0, i, k, 0])) + Fraction(float(M[0, i, k, 1])))
                                                                   * Fraction(float(z[k])) for k in range(ns))) for i in range(ns)])
        else:
            z = M[0, :, :, 0] @ z + s[j]
    y, _ = run(zs)
    return y.reshape(-1)[:N]


def test_block_scan_restated_in_numpy_lands_at_the_reference_noise_floor():
    x = np.random.default_rng(1).standard_normal(3000)
    ref = ss.filtfilt(B_, A_, x)
    L = 128
    M = _scan_matrices(A_, L, 1)
    got = _filtfilt_loop(x, np.float64, lambda b, a, xs, zi, dt: _scan_lfilter(b, a, xs, zi, L, M))
    dev = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert dev < 2e-7, dev                                               # observed 2e-8 - 5e-8
    plain = _filtfilt_loop(x, np.float64, lambda b, a, xs, zi, dt: _scan_lfilter(b, a, xs, zi, L, M, compensated=False))
    dev_plain = float(np.abs(plain - ref).max() / np.abs(ref).max())
    assert dev_plain > 20 * dev, (dev_plain, dev)                        # the block-start states need more than fp64
