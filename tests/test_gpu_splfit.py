"""GPU parity of the spline tweak's FIT (nadavca_amd/csrc/kernels_splfit.hip, /root/reference/nadavca/read.py:83-93)
through the C-ABI: per read the filter, the sort and FITPACK's fit, against numpy + ``scipy.interpolate.splrep``
themselves — knots and coefficients bit for bit — on short and long reads (LDS and global-memory sort), tied means,
reads with too few usable events, reads that failed the pre-alignment, empty events (NaN means); the classes the
kernel must hand back to FITPACK; and ``estimate_snps_batch`` with the fit on the device against the same call
with scipy doing the fits."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from nadavca_amd import _lib
    ctx = _lib.default_context()
    return dict(torch=torch, ctx=ctx, dev=torch.device('cuda', ctx.device))


def _up(env, a, dt):
    return env['torch'].from_numpy(np.ascontiguousarray(a, dtype=dt)).to(env['dev'])


def reference_fit(observed, levels):
    """read.py:88-92 with the reference's own calls"""
    from scipy import interpolate
    with np.errstate(invalid='ignore'):
        keep = np.abs(levels - observed) <= 1
    xs, ys = observed[keep], levels[keep]
    if xs.size < 4:
        return None
    order = np.lexsort((ys, xs))
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        return interpolate.splrep(xs[order], ys[order], s=len(xs))


def make_reads(rng, sizes):
    means, levels = [], []
    for j, R in enumerate(sizes):
        lv = rng.normal(0, 1.1, R)
        ob = lv * rng.uniform(0.9, 1.1) + rng.uniform(-0.2, 0.2) + rng.normal(0, rng.choice([0.05, 0.3, 0.8]), R)
        if j % 3 == 0:                          # integer ADC data: equal means, equal (mean, level) pairs
            ob, lv = np.round(ob * 12) / 12, np.round(lv * 16) / 16
        if j % 4 == 1 and R > 10:               # empty events
            ob[rng.integers(0, R, 3)] = np.nan
        means.append(ob)
        levels.append(lv)
    return means, levels


def test_fit_equals_numpy_and_splrep_bit_for_bit(env):
    from nadavca_amd.device import spline_fit_dev
    rng = np.random.default_rng(77)
    sizes = [5, 4, 3, 0, 1, 17, 63, 64, 65, 127, 128, 129, 480, 511, 512, 513, 1000, 1023, 1024, 1025, 1500, 2049,
             5200, 9000] + [int(v) for v in rng.integers(300, 700, 400)]
    means, levels = make_reads(rng, sizes)
    # a read of which only 3 events survive the filter, and one that failed the pre-alignment
    means[5] = levels[5] + 3.0
    means[5][:3] = levels[5][:3]
    status = np.zeros(len(sizes), dtype=np.int32)
    status[6] = 2
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    t, c, fit = spline_fit_dev(env['ctx'], _up(env, np.concatenate(means), np.float64),
                               _up(env, np.concatenate(levels), np.float64), _up(env, off, np.int64),
                               _up(env, status, np.int32))
    t, c, fit = t.cpu().numpy(), c.cpu().numpy(), fit.cpu().numpy()
    n_fit = 0
    for j in range(len(sizes)):
        want = reference_fit(means[j], levels[j]) if status[j] == 0 else None
        if want is None:
            assert fit[j] == 1, j
            continue
        assert fit[j] == 0, j
        assert len(want[0]) == 8, 'FITPACK placed a knot under the filter'
        assert np.array_equal(t[j], want[0]) and np.array_equal(c[j, :4], want[1][:4]) and not c[j, 4:].any(), j
        n_fit += 1
    assert n_fit > 400 and (fit == 1).sum() == 5     # reads 2, 3, 4 (too short), 5 (filter), 6 (status)


def test_reads_outside_the_polynomial_case_are_handed_back(env):
    from nadavca_amd.device import spline_fit_dev
    lv = np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    cases = [(np.full(6, 0.5), lv),                                   # all means equal: no interval to fit on
             (lv.copy(), np.array([0.1, np.nan, 0.3, 0.4, 0.5, 0.6])),   # (a NaN level is dropped by the filter:
             (lv + 0.01, lv)]                                          #  an ordinary read)
    off = np.arange(4, dtype=np.int64) * 6
    t, c, fit = spline_fit_dev(env['ctx'], _up(env, np.concatenate([a for a, _ in cases]), np.float64),
                               _up(env, np.concatenate([b for _, b in cases]), np.float64), _up(env, off, np.int64))
    assert fit.cpu().tolist() == [2, 0, 0]


def test_estimate_snps_batch_device_fit_equals_the_fitpack_path():
    """``estimate_snps_batch`` with the spline fit on the device against the same call with scipy's splrep doing
    the fits on the host (``spline_fit='host'``, in this process and in worker processes): identical chunks."""
    from nadavca_amd import _lib, dtw, synthetic
    from nadavca_amd.estimate_snps import estimate_snps_batch, last_batch_counts
    model = synthetic.load_model_arrays()
    km = dtw.KmerModel(*model, context=_lib.default_context())
    rb, aligner, genome = synthetic.make_read_batch(360, model, seed=91, genome_length=3000, length=220, spread=40,
                                                    substitution_rate=0.03)
    cfg = dict(bandwidth=150, snp_prior_probability=0.001, min_event_length=2, model_wobbling=True,
               model_transitions=True, tweak_signal_normalization=True, normalization_event_length=10)
    got = estimate_snps_batch(genome, rb, config=cfg, kmer_model=km, aligner=aligner)
    fitted = last_batch_counts['reads_fitted']
    assert fitted == 360 and last_batch_counts['reads_ok'] == 360
    for workers in (0, 2):
        host = estimate_snps_batch(genome, rb, config=cfg, kmer_model=km, aligner=aligner, fit_workers=workers,
                                   spline_fit='host')
        assert last_batch_counts['reads_fitted'] == fitted
        assert len(got) == len(host) >= 1
        for a, b in zip(got, host):
            assert (a.start, a.end) == (b.start, b.end)
            assert np.array_equal(a.coverage, b.coverage)
            assert np.max(np.abs(a.values - b.values)) < 1e-12     # (f64 atomics: the order of the sums differs)
