"""GPU parity of the device-side host steps (nadavca_amd/csrc/kernels_renorm.hip)
through the C-ABI: normalisation (exact medians) and per-event means against numpy — bit for bit —,
the linear re-fit against scipy.stats.linregress (a few ulp: numpy hands the centred products to BLAS),
and the device-resident renormalise / re-align loop against the same loop done on the host with the
reference's own numpy/scipy calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from nadavca_amd import _lib, dtw, synthetic
    ctx = _lib.default_context()
    model = synthetic.load_model_arrays()
    km = dtw.KmerModel(*model, context=ctx)
    return dict(torch=torch, ctx=ctx, km=km, model=model, dev=torch.device('cuda', ctx.device))


def _up(env, a, dt):
    return env['torch'].from_numpy(np.ascontiguousarray(a, dtype=dt)).to(env['dev'])


def test_normalize_groups_equals_numpy(env):
    from nadavca_amd.device import normalize_groups_dev
    rng = np.random.default_rng(11)
    groups = [np.round(rng.normal(90, 12, n)) for n in (1, 2, 3, 4, 5, 64, 1001, 4300, 50000)]  # ADC-like: ties
    groups += [rng.normal(0, 1, n) for n in (7, 8, 1000, 4097)]
    groups += [np.array([-3.0, -3.0, 2.0, 1e300, -1e300, 0.0, -0.0, 5.5])]
    off = np.zeros(len(groups) + 1, dtype=np.int64)
    np.cumsum([len(g) for g in groups], out=off[1:])
    out, cs = normalize_groups_dev(env['ctx'], _up(env, np.concatenate(groups), np.float64), _up(env, off, np.int64))
    out, cs = out.cpu().numpy(), cs.cpu().numpy()
    for j, g in enumerate(groups):
        centre = np.median(g)
        scale = np.median(abs(g - centre))
        assert cs[j, 0] == centre and cs[j, 1] == scale, j
        with np.errstate(divide='ignore', invalid='ignore'):
            want = np.clip((g - centre) / scale, -5, 5)
        assert np.array_equal(out[off[j]:off[j + 1]], want, equal_nan=True), j


@pytest.mark.parametrize('n', [65537, 200000, 300001])
def test_normalize_one_large_group_equals_numpy(env, n):
    """one group above 64 k samples takes the chip-wide selection (estimate_snps: all reads pooled)"""
    from nadavca_amd.device import normalize_groups_dev
    rng = np.random.default_rng(n)
    g = np.round(rng.normal(90, 12, n)) if n % 2 else rng.normal(90, 12, n)
    pad = rng.normal(0, 1, 5)  # the group need not start at the buffer's first sample
    raw = np.concatenate([pad, g])
    out, cs = normalize_groups_dev(env['ctx'], _up(env, raw, np.float64), _up(env, [5, 5 + n], np.int64))
    out, cs = out.cpu().numpy(), cs.cpu().numpy()
    centre = np.median(g)
    scale = np.median(abs(g - centre))
    assert cs[0, 0] == centre and cs[0, 1] == scale
    assert np.array_equal(out[5:], np.clip((g - centre) / scale, -5, 5))


def test_normalize_reads_device_matches_reference_fixture():
    """G5: Read.normalize_reads of the reference on the fixture's reads (one centre/scale for all)."""
    from est_fixture import EstimatorFixture
    from nadavca_amd.read import Read
    fx = EstimatorFixture()
    reads = fx.reads(normalize=False)
    Read.normalize_reads_device(reads)
    for i, r in enumerate(reads):
        assert np.array_equal(r.normalized_signal[:64], fx.z['r%d_normalized_head' % i])
    assert np.array_equal(np.array([float(np.sum(r.normalized_signal)) for r in reads]),
                          fx.z['normalized_checksum'])
    # per read: the same as the host routine applied to one read at a time
    host = fx.reads(normalize=False)
    for r in host:
        Read.normalize_reads([r])
    Read.normalize_reads_device(reads, per_read=True)
    for a, b in zip(reads, host):
        assert np.array_equal(a.normalized_signal, b.normalized_signal)


def _batch(env, n=24, **kw):
    from nadavca_amd import synthetic
    from nadavca_amd.device import DeviceBatch
    batch = synthetic.make_batch(n, env['model'], seed=77, **kw)
    return batch, DeviceBatch(batch, env['dev'])


def test_event_means_equal_numpy_mean(env):
    from nadavca_amd.device import event_means_dev
    rng = np.random.default_rng(12)
    batch, dbatch = _batch(env, 6, R=300, R_spread=40, bandwidth=100)
    # arbitrary events, lengths 0 .. ~700 (beyond numpy's 128-element blocks), not an alignment
    events = np.zeros((dbatch.total_ref, 2), dtype=np.int32)
    for j in range(batch.n):
        N = int(batch.sig_off[j + 1] - batch.sig_off[j])
        R = int(batch.ref_off[j + 1] - batch.ref_off[j])
        length = np.where(rng.random(R) < 0.2, rng.integers(0, min(N, 700), R), rng.integers(0, 24, R))
        start = rng.integers(0, N - length + 1)
        events[batch.ref_off[j]:batch.ref_off[j + 1], 0] = start
        events[batch.ref_off[j]:batch.ref_off[j + 1], 1] = start + length
    status = np.zeros(batch.n, dtype=np.int32)
    status[3] = 1
    got = event_means_dev(dbatch, env['ctx'], _up(env, events, np.int32), _up(env, status, np.int32)).cpu().numpy()
    for j in range(batch.n):
        sig = batch.signal[batch.sig_off[j]:batch.sig_off[j + 1]]
        for g in range(int(batch.ref_off[j]), int(batch.ref_off[j + 1])):
            s, e = events[g]
            if status[j] or e == s:
                assert np.isnan(got[g])
            else:
                assert got[g] == np.mean(sig[s:e]), (j, g, e - s)


def test_linfit_rescale_matches_linregress(env):
    from scipy.stats import linregress
    from nadavca_amd.device import linfit_rescale_dev
    rng = np.random.default_rng(13)
    batch, dbatch = _batch(env, 8, R=200, R_spread=60, bandwidth=80)
    x = rng.normal(0, 1.2, dbatch.total_ref)
    y = 1.07 * x + 0.13 + rng.normal(0, 0.2, dbatch.total_ref)
    status = np.zeros(batch.n, dtype=np.int32)
    status[5] = 1
    before = dbatch.signal.cpu().numpy().copy()
    fit = linfit_rescale_dev(dbatch, env['ctx'], _up(env, x, np.float64), _up(env, y, np.float64),
                             _up(env, status, np.int32)).cpu().numpy()
    after = dbatch.signal.cpu().numpy()
    for j in range(batch.n):
        s0, s1 = int(batch.sig_off[j]), int(batch.sig_off[j + 1])
        if status[j]:
            assert np.isnan(fit[j]).all() and np.array_equal(after[s0:s1], before[s0:s1])
            continue
        r0, r1 = int(batch.ref_off[j]), int(batch.ref_off[j + 1])
        slope, intercept = linregress(x[r0:r1], y[r0:r1])[:2]
        assert np.isclose(fit[j, 0], slope, rtol=1e-13, atol=0) and np.isclose(fit[j, 1], intercept, rtol=1e-12, atol=1e-15)
        # the rescale itself is exact given the fitted line
        assert np.array_equal(after[s0:s1], (before[s0:s1] - fit[j, 1]) / fit[j, 0])


def test_renorm_loop_on_device_equals_host_loop(env):
    """align, re-fit, re-align, re-fit on the device vs the same rounds on the host with numpy.mean and
    scipy.stats.linregress (align_signal.py:55-80) around the same alignment kernel."""
    from scipy.stats import linregress
    from nadavca_amd import dtw
    from nadavca_amd.device import refine_renorm_loop_dev
    km = env['km']
    batch, dbatch = _batch(env, 24, R=160, R_spread=30, bandwidth=60)
    # de-normalise the reads a little so that the re-fit has something to do
    rng = np.random.default_rng(14)
    for j in range(batch.n):
        s0, s1 = int(batch.sig_off[j]), int(batch.sig_off[j + 1])
        batch.signal[s0:s1] = batch.signal[s0:s1] * rng.uniform(0.93, 1.08) + rng.uniform(-0.15, 0.15)
    from nadavca_amd.device import DeviceBatch
    dbatch = DeviceBatch(batch, env['dev'])
    events, status, fits = refine_renorm_loop_dev(dbatch, 60, 2, km, True, 3)
    events, status = events.cpu().numpy(), status.cpu().numpy()
    assert len(fits) == 2 and (status == 0).all()
    # host loop
    sig = batch.signal.copy()
    expected = km.get_expected_signal_batch(
        [(batch.reference[batch.ref_off[j]:batch.ref_off[j + 1]], [], []) for j in range(batch.n)])

    def align():
        b = dtw.FlatBatch.from_arrays(sig, batch.sig_off, batch.reference, batch.ref_off, batch.context_before,
                                      batch.cb_off, batch.context_after, batch.ca_off, batch.anchors, batch.anc_off)
        return dtw.refine_alignment_flat(b, 60, 2, km, True)[0]

    def refit(ev):
        for j in range(batch.n):
            s0, s1 = int(batch.sig_off[j]), int(batch.sig_off[j + 1])
            rows = ev[batch.ref_off[j]:batch.ref_off[j + 1]]
            means = [np.mean(sig[s0 + s:s0 + e]) for s, e in rows]
            slope, intercept = linregress(expected[j], means)[:2]
            sig[s0:s1] = (sig[s0:s1] - intercept) / slope

    ev = align()
    refit(ev)
    ev = align()
    refit(ev)
    assert np.array_equal(events, ev)
    assert np.allclose(dbatch.signal.cpu().numpy(), sig, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('degree', [1, 3, 5])
def test_splev_groups_equals_scipy(env, degree):
    """FITPACK's splev restated on the device: bit for bit scipy.interpolate.splev (ext=0), smoothing and
    interpolating splines, few knots (LDS) and many (global memory), samples outside the knots and on them."""
    from scipy import interpolate
    from nadavca_amd.device import splev_groups_dev
    rng = np.random.default_rng(21 + degree)
    xs, tcks = [], []
    for npts, smooth in ((40, 1.0), (300, 1.0), (300, 0.05), (700, 0.0), (12, 1.0)):
        x = np.sort(rng.normal(0, 1.2, npts))
        x += np.arange(npts) * 1e-9  # strictly increasing for s = 0
        y = 1.05 * x + 0.1 + 0.3 * np.sin(2 * x) + rng.normal(0, 0.2, npts)
        tck = interpolate.splrep(x, y, k=degree, s=npts * smooth)
        tcks.append(tck)
        xs.append(np.concatenate([rng.normal(0, 1.6, 3000), [-6.0, 6.0, x[0], x[-1]], tck[0]]))
    assert max(len(t[0]) for t in tcks) > 512 > min(len(t[0]) for t in tcks)
    off = np.zeros(len(xs) + 1, dtype=np.int64)
    np.cumsum([len(a) for a in xs], out=off[1:])
    koff = np.zeros(len(xs) + 1, dtype=np.int64)
    np.cumsum([len(t[0]) for t in tcks], out=koff[1:])
    got = splev_groups_dev(env['ctx'], _up(env, np.concatenate(xs), np.float64), _up(env, off, np.int64),
                           _up(env, np.concatenate([t[0] for t in tcks]), np.float64),
                           _up(env, np.concatenate([t[1][:len(t[0])] for t in tcks]), np.float64),
                           _up(env, koff, np.int64), degree).cpu().numpy()
    for j, (a, tck) in enumerate(zip(xs, tcks)):
        assert np.array_equal(got[off[j]:off[j + 1]], interpolate.splev(a, tck)), j
