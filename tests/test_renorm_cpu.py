"""CPU checks behind nadavca_amd/csrc/kernels_renorm.hip: the summation order its per-event means
restate (numpy's pairwise scheme for contiguous float64) really is what ``numpy.mean`` does on this
numpy, and the median rule (middle element / mean of the two middle ones) is numpy's and
``statistics.median``'s.  The device code itself is compared with numpy in tests/test_gpu_renorm.py."""
import statistics

import numpy as np


def np_block_sum(a):
    n = len(a)
    if n < 8:
        res = 0.0
        for v in a:
            res += v
        return res
    r = [float(v) for v in a[:8]]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res += a[i]
        i += 1
    return res


def np_pairwise_sum(a):
    n = len(a)
    if n <= 128:
        return np_block_sum(a)
    n2 = n // 2
    n2 -= n2 % 8
    return np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:])


def np_sum(a, chunk=8192):
    """numpy.add.reduce of a contiguous float64 vector: the reduction loop is handed the data in pieces
    of 8192 elements (numpy's buffer size), each summed pairwise and added to the running result."""
    res = 0.0
    for o in range(0, len(a), chunk):
        res = res + np_pairwise_sum(a[o:o + chunk])
    return res


def test_pairwise_order_is_numpy_mean():
    rng = np.random.default_rng(5)
    sizes = list(range(1, 200)) + [255, 256, 257, 300, 511, 1000, 1023, 1024, 1025, 4097, 8191, 8192, 8193,
                                   10001, 16385, 30000]
    for n in sizes:
        a = rng.normal(0, 3, n) * 10.0 ** rng.integers(-3, 4, n)
        assert np_sum(a) / n == np.mean(a), n


def test_median_rule():
    rng = np.random.default_rng(6)
    for n in (1, 2, 3, 4, 7, 100, 101, 1000):
        a = np.round(rng.normal(0, 30, n))  # many ties, like raw ADC counts
        s = np.sort(a)
        want = s[n // 2] if n % 2 else (s[n // 2 - 1] + s[n // 2]) / 2
        assert want == np.median(a) == statistics.median(a.tolist())
