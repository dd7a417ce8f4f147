"""CPU check behind nadavca_amd/csrc/splfit.h (the fit kernel of the spline tweak, read.py:83-93): the header's
restatement of FITPACK's first trial, compiled for the host, against ``scipy.interpolate.splrep`` itself — knots
and coefficients bit for bit — and the claim the kernel rests on: under the reference's filter (|level - mean| <= 1)
and its smoothing factor (s = number of points) FITPACK never places a knot.  The device build of the same header is
compared with scipy in tests/test_gpu_splfit.py."""
import ctypes as C
import os
import subprocess
import warnings

import numpy as np
import pytest
from scipy import interpolate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def first_pass(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('splfit') / 'splfit_host.so')
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC',
                    os.path.join(ROOT, 'tests', 'host_shims', 'splfit_host.cpp'), '-o', so], check=True)
    lib = C.CDLL(so)
    f = lib.splfit_host_cubic_first_pass
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
    f.restype = C.c_int

    def run(x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        t, c = np.zeros(8), np.zeros(8)
        st = f(x.ctypes.data, y.ctypes.data, len(x), t.ctypes.data, c.ctypes.data)
        return st, t, c
    return run


def tweak_pairs(rng, m, noise, quantised):
    """(means, levels) as read.py:88-90 hands them to splrep: filtered, sorted by mean then level"""
    y = rng.normal(0, 1, m) * rng.choice([0.3, 1.0, 2.0])
    x = y + rng.uniform(-1, 1, m) * noise
    if quantised:                        # equal means and equal (mean, level) pairs, as integer ADC data gives
        x, y = np.round(x * 12) / 12, np.round(y * 20) / 20
    keep = np.abs(y - x) <= 1
    x, y = x[keep], y[keep]
    order = np.lexsort((y, x))
    return x[order], y[order]


def test_first_pass_equals_splrep_bit_for_bit(first_pass):
    rng = np.random.default_rng(1)
    done = 0
    for trial in range(1500):
        x, y = tweak_pairs(rng, int(rng.integers(4, 900)), rng.choice([0.01, 0.3, 1.0, 1.5]), trial % 4 == 0)
        if len(x) < 4 or x[0] == x[-1]:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter('error')               # FITPACK's "s too small" etc. would show here
            t, c, k = interpolate.splrep(x, y, s=len(x))
        st, t8, c8 = first_pass(x, y)
        assert st == 0 and k == 3
        assert len(t) == 8, 'FITPACK placed a knot under the filter'
        assert np.array_equal(t, t8) and np.array_equal(c[:4], c8[:4]) and not c8[4:].any()
        done += 1
    assert done > 1400


def test_outside_the_filter_the_kernel_says_so_exactly_when_fitpack_goes_on(first_pass):
    """Without the filter the residual of the cubic can exceed s: the restatement must report it (the caller then
    uses FITPACK) exactly when FITPACK places knots — the acceptance test itself is restated, not approximated."""
    rng = np.random.default_rng(2)
    went_on = 0
    for trial in range(400):
        m = int(rng.integers(8, 300))
        x = np.sort(rng.normal(0, 1, m))
        y = np.sin(3 * x) * rng.choice([0.5, 2.0, 4.0]) + rng.normal(0, rng.choice([0.3, 1.0, 1.5]), m)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            t, c, k = interpolate.splrep(x, y, s=m)
        st, t8, c8 = first_pass(x, y)
        assert (st == 2) == (len(t) > 8)
        went_on += st == 2
        if st == 0:
            assert np.array_equal(t, t8) and np.array_equal(c[:4], c8[:4])
    assert 50 < went_on < 350


def test_degenerate_inputs_are_reported(first_pass):
    assert first_pass([0.5] * 6, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6])[0] == 2           # all means equal
    assert first_pass([0.0, 0.1, 0.2, 0.3, 0.4], [0.0, np.nan, 0.2, 0.3, 0.4])[0] == 2   # NaN residual
    st, t, c = first_pass([0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 8.0, 27.0])            # 4 points: the cubic interpolates
    assert st == 0 and np.allclose(interpolate.splev([0.5, 2.5], (t, c, 3)), [0.125, 15.625], rtol=1e-12)
