"""GPU parity: estimate_log_likelihoods through the C-ABI vs the reference fixtures and the CPU
oracle.  Tolerance: log-likelihoods to 1e-9 relative (+1e-9 absolute) — far inside the 1e-5
the SNP probabilities need; the -inf pattern (impossible hypotheses) must match exactly."""
import numpy as np
import pytest

from conftest import dp_args

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-9


@pytest.fixture(scope='module')
def dtw():
    from nadavca_amd import dtw as d
    return d


def _reads(cases):
    return [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
            for c in cases]


def _close(got, exp):
    assert got.shape == exp.shape
    assert np.array_equal(np.isneginf(got), np.isneginf(exp))
    assert not np.any(np.isnan(got))
    fin = np.isfinite(exp)
    assert np.allclose(got[fin], exp[fin], rtol=RTOL, atol=ATOL), float(np.max(np.abs(got[fin] - exp[fin])))


def _check_golden(dtw, g):
    k, c, a, mean, sigma = g.model
    m = dtw.KmerModel(k, c, a, mean, sigma)
    groups = {}
    for case in g.cases:
        groups.setdefault((int(case['bandwidth']), int(case['min_event_length'])), []).append(case)
    for (bw, mel), cases in groups.items():
        for w in (0, 1):
            got = dtw.estimate_log_likelihoods_batch(_reads(cases), bw, mel, m, bool(w))
            for case, ll in zip(cases, got):
                _close(ll, case['ell_w%d' % w])


def test_ell_golden_tiny(dtw, golden_tiny):
    _check_golden(dtw, golden_tiny)


def test_ell_golden_config(dtw, golden_config):
    _check_golden(dtw, golden_config)


def test_ell_golden_nopath(dtw, golden_nopath):
    _check_golden(dtw, golden_nopath)


def test_ell_appendix_c(dtw):
    ids = np.arange(64)
    m = dtw.KmerModel(3, 1, 4, ((ids * 37) % 64) / 16 - 2, 0.4 + (ids % 3) * 0.1)
    ref, cb, ca = [0, 1, 2, 3, 3, 1, 0, 2], [2], [1]
    es = m.get_expected_signal(ref, cb, ca)
    sig = np.round(np.repeat(es, 3) + 0.1 * ((np.arange(24) * 7) % 5 - 2), 4)
    anc = [[0, 0], [9, 3], [21, 7]]
    ll = dtw.estimate_log_likelihoods(sig, ref, cb, ca, anc, 4, 2, m, True)
    assert ll[0].tolist() == pytest.approx(
        [1.0189036593406566, -4.629338921878056, -13.895064347992593, -5.317795793541511], rel=1e-10)
    assert ll[4].tolist() == pytest.approx(
        [-7.647671624614224, -25.626219563738044, -105.98247035339004, 1.0189036593406566], rel=1e-10)
    ll = dtw.estimate_log_likelihoods(sig, ref, cb, ca, anc, 4, 2, m, False)
    assert ll[7].tolist() == pytest.approx(
        [-5.049523388951718, -27.40554597190469, -0.2599469729028855, -27.310869151977293], rel=1e-10)


@pytest.mark.parametrize('mel', [0, 1, 2, 3, 4])
def test_ell_vs_oracle_random(dtw, oracle_port, mel):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(21, k=5, central=2)
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i in range(12):
        rng = np.random.default_rng([88, mel, i])
        R = int(rng.integers(3, 90))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 50)),
                                            dwell=(max(mel, 1), 9), jitter=6,
                                            anchor_density=float(rng.uniform(0.1, 0.9)),
                                            with_context=bool(i % 3), trim=min(3, R // 3)))
    for bw in (12, 40):
        for w in (False, True):
            got = dtw.estimate_log_likelihoods_batch(_reads(cases), bw, mel, mg, w)
            for c_, ll in zip(cases, got):
                exp = oracle_port.estimate_log_likelihoods(c_['signal'], c_['reference'], c_['context_before'],
                                                           c_['context_after'], c_['approximate_alignment'],
                                                           bw, mel, mo, w)
                _close(ll, exp)


def test_ell_config_sized_vs_oracle(dtw, oracle_port):
    """BASELINE config 3 shape (R~400, N~4000, bandwidth 150), 6 reads (the oracle needs ~0.4 s each)."""
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    batch = synthetic.make_batch(6, model, seed=9, R=400, R_spread=40, bandwidth=150)
    got = dtw.estimate_log_likelihoods_batch(_reads(batch.cases), 150, 2, mg, True)
    for c_, ll in zip(batch.cases, got):
        exp = oracle_port.estimate_log_likelihoods(c_['signal'], c_['reference'], c_['context_before'],
                                                   c_['context_after'], c_['approximate_alignment'], 150, 2,
                                                   mo, True)
        _close(ll, exp)
        # the reference base's column carries the same no-substitution likelihood in every row
        ref = c_['reference']
        col = ll[np.arange(len(ref)), ref]
        assert np.all(col == col[0])
        # and the true base is the most likely one almost everywhere on clean synthetic data
        assert np.mean(np.argmax(ll, axis=1) == ref) > 0.95


def test_ell_exact_variant_matches(golden_config):
    """NADAVCA_ELL_KERNEL=1 selects the original hypothesis phase (two polynomial densities per lane,
    LDS hand-over); it must reproduce the fixtures like the default one (child process: the variant is
    read from the environment at call time)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import GoldenFile
from nadavca_amd import dtw
for name in ("dp_tiny.npz", "dp_config.npz"):
    g = GoldenFile(name)
    k, c, a, mean, sigma = g.model
    m = dtw.KmerModel(k, c, a, mean, sigma)
    groups = {}
    for case in g.cases:
        groups.setdefault((int(case["bandwidth"]), int(case["min_event_length"])), []).append(case)
    for (bw, mel), cases in groups.items():
        reads = [(x["signal"], x["reference"], x["context_before"], x["context_after"], x["approximate_alignment"]) for x in cases]
        for w in (0, 1):
            got = dtw.estimate_log_likelihoods_batch(reads, bw, mel, m, bool(w))
            for case, ll in zip(cases, got):
                exp = case["ell_w%%d" %% w]
                assert np.array_equal(np.isneginf(ll), np.isneginf(exp))
                fin = np.isfinite(exp)
                assert np.allclose(ll[fin], exp[fin], rtol=1e-9, atol=1e-9)
print("ELL-EXACT-OK")
""" % (ROOT, ROOT)
    env = dict(os.environ, NADAVCA_ELL_KERNEL='1')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'ELL-EXACT-OK' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize('k,central', [(7, 3), (4, 0), (2, 1), (8, 3), (10, 4), (10, 0)])
def test_ell_other_kmer_sizes(dtw, oracle_port, k, central):
    """k-mer sizes other than the packaged 6: from k = 7 on a hypothesis takes a group of 16 lanes instead of
    8 (k + 2 lanes: the k-mer before the first touched position, the <= k touched positions, the closing
    lane) — k = 10 is the size of the table the reference names as its default
    (/root/reference/nadavca/defaults.py:4, a 4^10-row table; kmer_model.cpp:22-30 takes any k); small k and
    off-centre k-mers move the range of rows a substitution touches."""
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(31 + k, k=k, central=central)
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i in range(6):
        rng = np.random.default_rng([89, k, i])
        R = int(rng.integers(4, 70))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(10, 40)), dwell=(2, 9), jitter=5,
                                            anchor_density=float(rng.uniform(0.2, 0.9)), with_context=bool(i % 2),
                                            trim=min(3, R // 3)))
    for w in (False, True):
        got = dtw.estimate_log_likelihoods_batch(_reads(cases), 30, 2, mg, w)
        for c_, ll in zip(cases, got):
            exp = oracle_port.estimate_log_likelihoods(c_['signal'], c_['reference'], c_['context_before'],
                                                       c_['context_after'], c_['approximate_alignment'], 30, 2, mo, w)
            _close(ll, exp)
    got = dtw.refine_alignment_batch(_reads(cases), 30, 2, mg, True)
    for c_, ev in zip(cases, got):
        exp = oracle_port.refine_alignment(c_['signal'], c_['reference'], c_['context_before'], c_['context_after'],
                                           c_['approximate_alignment'], 30, 2, mo, True)
        assert np.array_equal(ev, exp)


@pytest.mark.parametrize('alphabet', [3, 5])
def test_ell_other_alphabet_sizes(dtw, oracle_port, alphabet):
    """The substitution hypotheses run over alphabet - 1 bases per position; the reference's
    KmerModel takes the alphabet size as a parameter (kmer_model.cpp:6-14)."""
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(41 + alphabet, k=4, central=1, alphabet=alphabet)
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i in range(5):
        rng = np.random.default_rng([90, alphabet, i])
        R = int(rng.integers(5, 60))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=25, dwell=(2, 8), jitter=4,
                                            anchor_density=0.6, with_context=bool(i % 2), trim=min(3, R // 3)))
    for w in (False, True):
        got = dtw.estimate_log_likelihoods_batch(_reads(cases), 25, 2, mg, w)
        for c_, ll in zip(cases, got):
            assert ll.shape == (len(c_['reference']), alphabet)
            exp = oracle_port.estimate_log_likelihoods(c_['signal'], c_['reference'], c_['context_before'],
                                                       c_['context_after'], c_['approximate_alignment'], 25, 2, mo, w)
            _close(ll, np.asarray(exp))
