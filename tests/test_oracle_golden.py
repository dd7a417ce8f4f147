"""The CPU oracle (oracle/nadavca_oracle.c) against the reference's outputs.

Fixtures in tests/golden/ were produced by the reference's own C++ compiled in place
(oracle/make_golden.py); on the same toolchain the restatement is bit-identical, the
assertions allow 1e-12 relative so that a different libm build still passes."""
import numpy as np
import pytest

from conftest import dp_args


def _check_group(o, g, ll_rtol=1e-12):
    k, c, a, mean, sigma = g.model
    m = o.KmerModel(k, c, a, mean, sigma)
    for case in g.cases:
        args = dp_args(case) + (m,)
        for tr in (0, 1):
            got = o.refine_alignment(*args, bool(tr))
            exp = case['refine_t%d' % tr]
            assert got.shape == exp.shape
            assert np.array_equal(got, exp)
        for w in (0, 1):
            got = o.estimate_log_likelihoods(*args, bool(w))
            exp = case['ell_w%d' % w]
            assert got.shape == exp.shape
            assert np.array_equal(np.isneginf(got), np.isneginf(exp))
            fin = np.isfinite(exp)
            assert np.allclose(got[fin], exp[fin], rtol=ll_rtol, atol=0)
        es = m.get_expected_signal(case['reference'], case['context_before'], case['context_after'])
        assert np.array_equal(es, case['expected_signal'])


def test_port_matches_reference_tiny(oracle_port, golden_tiny):
    _check_group(oracle_port, golden_tiny)


def test_port_matches_reference_config(oracle_port, golden_config):
    _check_group(oracle_port, golden_config)


def test_port_matches_reference_nopath(oracle_port, golden_nopath):
    _check_group(oracle_port, golden_nopath)
    case = golden_nopath.cases[0]
    assert case['refine_t1'].size == 0 and np.all(np.isneginf(case['ell_w1']))


def test_appendix_c_known_answers(oracle_port):
    """SURVEY.md Appendix C (values captured from the compiled reference)."""
    o = oracle_port
    ids = np.arange(64)
    m = o.KmerModel(3, 1, 4, ((ids * 37) % 64) / 16 - 2, 0.4 + (ids % 3) * 0.1)
    ref, cb, ca = [0, 1, 2, 3, 3, 1, 0, 2], [2], [1]
    es = m.get_expected_signal(ref, cb, ca)
    assert es.tolist() == [-1.6875, -0.125, 0.4375, -1.3125, -0.9375, -1.75, -0.375, -1.1875]
    sig = np.round(np.repeat(es, 3) + 0.1 * ((np.arange(24) * 7) % 5 - 2), 4)
    anc = [[0, 0], [9, 3], [21, 7]]
    want = [[1, 3], [3, 6], [6, 9], [9, 12], [12, 15], [15, 18], [18, 21], [21, 24]]
    for tr in (True, False):
        assert o.refine_alignment(sig, ref, cb, ca, anc, 4, 2, m, tr).tolist() == want
    ll = o.estimate_log_likelihoods(sig, ref, cb, ca, anc, 4, 2, m, True)
    assert ll[0].tolist() == pytest.approx(
        [1.0189036593406566, -4.629338921878056, -13.895064347992593, -5.317795793541511], rel=1e-13)
    assert ll[4].tolist() == pytest.approx(
        [-7.647671624614224, -25.626219563738044, -105.98247035339004, 1.0189036593406566], rel=1e-13)
    ll = o.estimate_log_likelihoods(sig, ref, cb, ca, anc, 4, 2, m, False)
    assert ll[7].tolist() == pytest.approx(
        [-5.049523388951718, -27.40554597190469, -0.2599469729028855, -27.310869151977293], rel=1e-13)


def test_port_bit_identical_to_compiled_reference_when_present(oracle_port):
    """Where oracle/_ref exists (build container, or shipped to the GPU box) the
    restatement and the reference agree bit for bit on fresh random cases."""
    from oracle.oracle import Oracle, have_reference
    if not have_reference():
        pytest.skip('oracle/_ref not built')
    from nadavca_amd import synthetic
    ref = Oracle('reference')
    model = synthetic.synth_model_arrays(3, k=4, central=1)
    mp = oracle_port.KmerModel(*model)
    mr = ref.KmerModel(*model)
    for i in range(6):
        rng = np.random.default_rng([55, i])
        mel = [2, 1, 3, 0, 2, 2][i]
        c = synthetic.make_dp_case(rng, model, R=40 + 5 * i, bandwidth=20 + i, dwell=(2, 7), jitter=4)
        a = (c['signal'], c['reference'], c['context_before'], c['context_after'],
             c['approximate_alignment'], 20 + i, mel)
        for flag in (True, False):
            assert np.array_equal(oracle_port.refine_alignment(*a, mp, flag), ref.refine_alignment(*a, mr, flag))
            x = oracle_port.estimate_log_likelihoods(*a, mp, flag)
            y = ref.estimate_log_likelihoods(*a, mr, flag)
            assert np.array_equal(x, y)


def test_long_double_referee_builds_and_disagrees_where_pinned():
    """oracle/liboracle_ld.so (the restatement in 80-bit long double) is the referee for rows the engine
    and the double reference disagree on; on the pinned fuzz cases the double and long-double results
    must differ exactly on the pinned rows (CPU only; the engine's side is checked in the GPU tests)."""
    import numpy as np
    from fuzz_cases import make_fuzz_batch
    from oracle.oracle import Oracle, LongDoubleReferee
    o = Oracle('port')
    for seed, it, case, rows in [(7, 448, 7, [140]), (7, 2897, 6, [33]), (1, 694, 2, [11, 115, 116, 136])]:
        fb = make_fuzz_batch(seed, it)
        c = fb['cases'][case]
        mo = o.KmerModel(*fb['model'])
        a = (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'],
             fb['bw'], fb['mel'])
        dbl = o.refine_alignment(*a, mo, fb['tr'])
        ld = LongDoubleReferee(*fb['model']).refine_alignment(*a, fb['tr'])
        assert np.nonzero((dbl != ld).any(axis=1))[0].tolist() == rows
