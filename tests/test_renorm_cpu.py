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


# ---- FITPACK splev (de Boor evaluation), as restated by nvk_splev_groups_dev -----------------------------
def fp_splev(t, c, k, xs):
    """scipy.interpolate.splev(xs, (t, c, k)) with ext=0: FITPACK's splev.f / fpbspl.f operation for
    operation (knot interval by position, the stable B-spline recurrence, the dot product in index order)."""
    t = np.asarray(t, dtype=float)
    n = len(t)
    k1 = k + 1
    nk1 = n - k1
    out = np.empty(len(xs))
    for m, x in enumerate(xs):
        # t[l-1] <= x < t[l] in 1-based FITPACK terms, l clamped to [k1, nk1]
        l = k1
        while not (x < t[l] or l == nk1):
            l += 1
        h = [0.0] * (k1 + 1)
        hh = [0.0] * (k1 + 1)
        h[0] = 1.0
        for j in range(1, k + 1):
            for i in range(j):
                hh[i] = h[i]
            h[0] = 0.0
            for i in range(1, j + 1):
                li = l + i
                lj = li - j
                tli, tlj = t[li - 1], t[lj - 1]
                if tli == tlj:
                    h[i] = 0.0
                    continue
                f = hh[i - 1] / (tli - tlj)
                h[i - 1] = h[i - 1] + f * (tli - x)
                h[i] = f * (x - tlj)
        sp = 0.0
        ll = l - k1
        for j in range(k1):
            sp = sp + c[ll + j] * h[j]
        out[m] = sp
    return out


def test_splev_restatement_is_scipy():
    from scipy import interpolate
    rng = np.random.default_rng(9)
    for trial in range(6):
        npts = int(rng.integers(30, 400))
        x = np.sort(rng.normal(0, 1.2, npts))
        y = 1.05 * x + 0.1 + 0.3 * np.sin(2 * x) + rng.normal(0, 0.2, npts)
        tck = interpolate.splrep(x, y, s=npts * (0.2 if trial % 2 else 1.0))
        xs = np.concatenate([rng.normal(0, 1.5, 500), [-5.0, 5.0, x[0], x[-1]], tck[0]])  # incl. outside, knots
        assert np.array_equal(fp_splev(tck[0], tck[1], tck[2], xs), interpolate.splev(xs, tck))
