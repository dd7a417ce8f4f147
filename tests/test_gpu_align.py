"""GPU parity: refine_alignment / get_expected_signal through the C-ABI vs the reference
fixtures (tests/golden, produced by the reference's own C++) and vs the CPU oracle on fresh
seeded inputs.  Integer alignments must be EXACTLY equal."""
import numpy as np
import pytest

from conftest import dp_args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dtw():
    from nadavca_amd import dtw as d
    return d


def _reads(cases):
    return [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
            for c in cases]


def _check_golden(dtw, g):
    k, c, a, mean, sigma = g.model
    m = dtw.KmerModel(k, c, a, mean, sigma)
    # group by (bandwidth, mel) so each group is one batched launch
    groups = {}
    for case in g.cases:
        groups.setdefault((int(case['bandwidth']), int(case['min_event_length'])), []).append(case)
    for (bw, mel), cases in groups.items():
        for tr in (0, 1):
            got = dtw.refine_alignment_batch(_reads(cases), bw, mel, m, bool(tr))
            for case, ev in zip(cases, got):
                exp = case['refine_t%d' % tr]
                assert ev.shape == exp.shape, (bw, mel, tr)
                assert np.array_equal(ev, exp), (bw, mel, tr)
        es = m.get_expected_signal_batch([(c_['reference'], c_['context_before'], c_['context_after'])
                                          for c_ in cases])
        for case, e in zip(cases, es):
            assert np.array_equal(e, case['expected_signal'])


def test_refine_golden_tiny(dtw, golden_tiny):
    _check_golden(dtw, golden_tiny)


def test_refine_golden_config(dtw, golden_config):
    _check_golden(dtw, golden_config)


def test_refine_golden_nopath(dtw, golden_nopath):
    _check_golden(dtw, golden_nopath)
    k, c, a, mean, sigma = golden_nopath.model
    m = dtw.KmerModel(k, c, a, mean, sigma)
    case = golden_nopath.cases[0]
    out = dtw.refine_alignment(*dp_args(case), m, True)
    assert len(out) == 0


def test_appendix_c(dtw):
    ids = np.arange(64)
    m = dtw.KmerModel(3, 1, 4, ((ids * 37) % 64) / 16 - 2, 0.4 + (ids % 3) * 0.1)
    ref, cb, ca = [0, 1, 2, 3, 3, 1, 0, 2], [2], [1]
    es = m.get_expected_signal(ref, cb, ca)
    assert es.tolist() == [-1.6875, -0.125, 0.4375, -1.3125, -0.9375, -1.75, -0.375, -1.1875]
    sig = np.round(np.repeat(es, 3) + 0.1 * ((np.arange(24) * 7) % 5 - 2), 4)
    want = [[1, 3], [3, 6], [6, 9], [9, 12], [12, 15], [15, 18], [18, 21], [21, 24]]
    for tr in (True, False):
        got = dtw.refine_alignment(signal=sig, reference=ref, context_before=cb, context_after=ca,
                                   approximate_alignment=[[0, 0], [9, 3], [21, 7]], bandwidth=4,
                                   min_event_length=2, kmer_model=m, model_transitions=tr)
        assert got.tolist() == want


@pytest.mark.parametrize('mel', [0, 1, 2, 3, 4])
def test_refine_vs_oracle_random(dtw, oracle_port, mel):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(11, k=5, central=2)
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i in range(24):
        rng = np.random.default_rng([77, mel, i])
        R = int(rng.integers(3, 140))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 60)),
                                            dwell=(max(mel, 1), 9), jitter=6,
                                            anchor_density=float(rng.uniform(0.1, 0.9)),
                                            with_context=bool(i % 3), trim=min(3, R // 3)))
    for bw in (10, 45):
        for tr in (False, True):
            got = dtw.refine_alignment_batch(_reads(cases), bw, mel, mg, tr)
            for c_, ev in zip(cases, got):
                exp = oracle_port.refine_alignment(c_['signal'], c_['reference'], c_['context_before'],
                                                   c_['context_after'], c_['approximate_alignment'], bw, mel,
                                                   mo, tr)
                assert ev.shape == exp.shape
                assert np.array_equal(ev, exp)


def test_refine_config_sized_vs_oracle(dtw, oracle_port):
    """BASELINE config 2 shape (R~400, N~4000, bandwidth 150), 48 reads in one launch."""
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    batch = synthetic.make_batch(48, model, seed=5, R=400, R_spread=40, bandwidth=150)
    for tr in (True, False):
        got = dtw.refine_alignment_batch(_reads(batch.cases), 150, 2, mg, tr)
        for c_, ev in zip(batch.cases, got):
            exp = oracle_port.refine_alignment(c_['signal'], c_['reference'], c_['context_before'],
                                               c_['context_after'], c_['approximate_alignment'], 150, 2, mo, tr)
            assert np.array_equal(ev, exp)
            # domain properties: events ordered, inside the slice, at least min_event_length long
            assert np.all(ev[:, 1] - ev[:, 0] >= 2)
            assert np.all(ev[1:, 0] >= ev[:-1, 1])
            assert ev[0, 0] >= 0 and ev[-1, 1] <= c_['signal'].size


def test_invalid_input_raises(dtw):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(1, k=3, central=1)
    m = dtw.KmerModel(*model)
    with pytest.raises(ValueError):
        dtw.refine_alignment(np.zeros(50), [0, 1, 2], [], [], [[5, 7]], 10, 2, m, True)  # anchor outside ref


@pytest.mark.parametrize('variant', ['1'])
def test_kernel_variants_match(golden_config, variant):
    """NADAVCA_ALIGN_KERNEL=1 (the exact scaled-number kernel that also serves as the default kernel's
    fallback) is a second implementation of the same operator: run it in a child process (the variant is
    read from the environment at call time) on the config-sized fixtures and on seeded reads."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import GoldenFile
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle
g = GoldenFile("dp_config.npz")
k, c, a, mean, sigma = g.model
m = dtw.KmerModel(k, c, a, mean, sigma)
reads = [(x["signal"], x["reference"], x["context_before"], x["context_after"], x["approximate_alignment"]) for x in g.cases]
for tr in (0, 1):
    got = dtw.refine_alignment_batch(reads, 150, 2, m, bool(tr))
    for x, ev in zip(g.cases, got):
        assert np.array_equal(ev, x["refine_t%%d" %% tr]), "golden mismatch"
o = Oracle("port"); model = synthetic.synth_model_arrays(11, k=5, central=2)
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
for mel in (0, 1, 2, 3, 4):
    cases = []
    for i in range(16):
        rng = np.random.default_rng([77, mel, i]); R = int(rng.integers(3, 140))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 60)), dwell=(max(mel, 1), 9), jitter=6,
                     anchor_density=float(rng.uniform(0.1, 0.9)), with_context=bool(i %% 3), trim=min(3, R // 3)))
    rs = [(x["signal"], x["reference"], x["context_before"], x["context_after"], x["approximate_alignment"]) for x in cases]
    for bw in (10, 45):
        for tr in (False, True):
            got = dtw.refine_alignment_batch(rs, bw, mel, mg, tr)
            for x, ev in zip(cases, got):
                exp = o.refine_alignment(x["signal"], x["reference"], x["context_before"], x["context_after"], x["approximate_alignment"], bw, mel, mo, tr)
                assert ev.shape == exp.shape and np.array_equal(ev, exp), ("oracle mismatch", mel, bw, tr)
print("VARIANT-OK")
''' % (ROOT, ROOT)
    env = dict(os.environ, NADAVCA_ALIGN_KERNEL=variant)
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'VARIANT-OK' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
