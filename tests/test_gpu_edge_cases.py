"""GPU parity on the edges the reference's code paths have: tiny reads, duplicate / unsorted
anchors ("later anchor overwrites", dtw.cpp:11-15), anchors on the last band row, empty batches,
and a long read with a wide band (BASELINE config 5 shape)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dtw():
    from nadavca_amd import dtw as d
    return d


def _both(dtw, oracle, model, case, bw, mel):
    mg = dtw.KmerModel(*model)
    mo = oracle.KmerModel(*model)
    a = (case['signal'], case['reference'], case['context_before'], case['context_after'],
         case['approximate_alignment'], bw, mel)
    for flag in (True, False):
        got = dtw.refine_alignment(*a, mg, flag)
        exp = oracle.refine_alignment(*a, mo, flag)
        assert got.shape == exp.shape and np.array_equal(got, exp), ('refine', flag)
        ll = dtw.estimate_log_likelihoods(*a, mg, flag)
        ex = oracle.estimate_log_likelihoods(*a, mo, flag)
        assert np.array_equal(np.isneginf(ll), np.isneginf(ex)), ('ell -inf pattern', flag)
        fin = np.isfinite(ex)
        assert np.allclose(ll[fin], ex[fin], rtol=1e-9, atol=1e-9), ('ell', flag)


def test_tiny_reads(dtw, oracle_port):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(5, k=4, central=1)
    for R in (1, 2, 3, 5):
        for i in range(3):
            rng = np.random.default_rng([901, R, i])
            c = synthetic.make_dp_case(rng, model, R=R, bandwidth=12, dwell=(2, 6), jitter=2,
                                       anchor_density=1.0, trim=0, pad_bases=3)
            _both(dtw, oracle_port, model, c, 12, 2)


def test_duplicate_unsorted_and_last_row_anchors(dtw, oracle_port):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(6, k=4, central=1)
    rng = np.random.default_rng(902)
    c = synthetic.make_dp_case(rng, model, R=30, bandwidth=15, dwell=(3, 7), jitter=3)
    anc = c['approximate_alignment'].copy()
    # the same reference index twice (the later row must win), rows out of order, and an anchor
    # on band row R (legal: the band arrays have R+1 entries)
    dup = np.array([[int(anc[3][0]) + 4, int(anc[3][1])]], dtype=np.int32)
    last = np.array([[len(c['signal']) - 2, 30]], dtype=np.int32)
    mixed = np.concatenate([anc[:6], dup, anc[6:][::-1], last]).astype(np.int32)
    c2 = dict(c, approximate_alignment=mixed)
    _both(dtw, oracle_port, model, c2, 15, 2)


def test_single_anchor_and_no_anchor(dtw, oracle_port):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(7, k=4, central=1)
    rng = np.random.default_rng(903)
    c = synthetic.make_dp_case(rng, model, R=12, bandwidth=40, dwell=(3, 5), jitter=0)
    one = dict(c, approximate_alignment=c['approximate_alignment'][:1])
    _both(dtw, oracle_port, model, one, 40, 2)
    none = dict(c, approximate_alignment=np.zeros((0, 2), dtype=np.int32))   # band = whole matrix
    _both(dtw, oracle_port, model, none, 40, 2)


def test_empty_batch_and_mixed_status(dtw, oracle_port):
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(8, k=4, central=1)
    m = dtw.KmerModel(*model)
    assert dtw.refine_alignment_batch([], 10, 2, m, True) == []
    assert dtw.estimate_log_likelihoods_batch([], 10, 2, m, True) == []
    rng = np.random.default_rng(904)
    good = synthetic.make_dp_case(rng, model, R=20, bandwidth=15, dwell=(3, 6), jitter=2)
    # no path: fewer samples than 2 * R
    nopath = dict(signal=rng.normal(0, 1, 30), reference=rng.integers(0, 4, 20).astype(np.int32),
                  context_before=np.zeros(0, np.int32), context_after=np.zeros(0, np.int32),
                  approximate_alignment=np.array([[0, 0], [29, 19]], dtype=np.int32))
    reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
             for c in (good, nopath, good)]
    out = dtw.refine_alignment_batch(reads, 15, 2, m, True)
    assert len(out[0]) == 20 and len(out[1]) == 0 and np.array_equal(out[0], out[2])
    ll = dtw.estimate_log_likelihoods_batch(reads, 15, 2, m, True)
    assert np.all(np.isneginf(ll[1])) and np.all(np.isfinite(ll[0][np.arange(20), good['reference']]))


def test_long_read_wide_band(dtw, oracle_port):
    """BASELINE config 5 shape, scaled to what the CPU oracle finishes in seconds: ~1200 bases,
    ~12000 samples, bandwidth 600 (skew c ~ 20, history and signal rings far larger than config 2)."""
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    rng = np.random.default_rng(905)
    c = synthetic.make_dp_case(rng, model, R=1200, bandwidth=600, jitter=60, anchor_density=0.5)
    a = (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'], 600, 2)
    for tr in (True, False):
        got = dtw.refine_alignment(*a, mg, tr)
        exp = oracle_port.refine_alignment(*a, mo, tr)
        assert np.array_equal(got, exp)


def test_mixed_narrow_and_wide_reads_in_one_batch(dtw, oracle_port):
    """A batch whose reads need different wavefront skews is served by two launches of the default
    kernel (small rings for the usual reads, large rings + longer rescale period for wide bands);
    every read must come out as if it had been aligned alone."""
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i, (R, jit) in enumerate([(60, 4), (700, 40), (45, 3), (900, 60), (30, 2)]):
        rng = np.random.default_rng([906, i])
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=400, jitter=jit, anchor_density=0.5))
    reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
             for c in cases]
    for tr in (True, False):
        got = dtw.refine_alignment_batch(reads, 400, 2, mg, tr)
        for c, ev in zip(cases, got):
            exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                               c['approximate_alignment'], 400, 2, mo, tr)
            assert np.array_equal(ev, exp)


def _fuzz_batch(seed, it):
    from fuzz_cases import make_fuzz_batch, reads_of
    fb = make_fuzz_batch(seed, it)
    return fb, reads_of(fb['cases'])


def test_sharp_model_collapse_goes_to_the_exact_kernel(dtw, oracle_port):
    """tests/dev/fuzz_parity.py seed 11, iteration 4230: a model three times sharper than the signal's noise.
    The wave's largest value collapses by more than the running scale can follow; capping the scale
    silently produced a different path while every row-mass check passed.  The read must come out as the
    reference has it (it is handed to the exact kernel)."""
    fb, reads = _fuzz_batch(11, 4230)
    mg = dtw.KmerModel(*fb['model'])
    mo = oracle_port.KmerModel(*fb['model'])
    got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])
    for c, ev in zip(fb['cases'], got):
        exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                           c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
        assert np.array_equal(ev, exp)


@pytest.mark.parametrize('seed,it,case,rows', [(7, 448, 7, [140]), (7, 2897, 6, [33]), (1, 694, 2, [11, 136])])
def test_differences_from_the_reference_are_its_own_rounding(dtw, oracle_port, seed, it, case, rows):
    """Rows on which the double-precision reference and the SAME algorithm evaluated in long double
    (oracle/liboracle_ld.so) disagree although no two adjacent k-mers are equal: the reference's
    log-sum-exp absorbs terms at |L| ~ 1e4, its own precision decides the row.  The engine gives one of the two
    answers there (round 2's arithmetic sided with long double on all of them), or flags the read."""
    from oracle.oracle import LongDoubleReferee
    fb, reads = _fuzz_batch(seed, it)
    mg = dtw.KmerModel(*fb['model'])
    mo = oracle_port.KmerModel(*fb['model'])
    c = fb['cases'][case]
    got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])[case]
    flag = int(mg.context.last_tie_flags(len(reads))[case])
    ref = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                       c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
    hp = LongDoubleReferee(*fb['model']).refine_alignment(c['signal'], c['reference'], c['context_before'],
                                                          c['context_after'], c['approximate_alignment'], fb['bw'],
                                                          fb['mel'], fb['tr'])
    for r in rows:
        assert not np.array_equal(ref[r], hp[r])      # the reference's own precision decides this row
        # the engine's answer is one of the two, or — an arg-max so ill-conditioned that three precisions give three
        # answers (seed 1, iteration 694, row 11: 95 / 96 / 99) — at least the read says so (a ULP or NEAR tie bit)
        assert np.array_equal(got[r], hp[r]) or np.array_equal(got[r], ref[r]) or (flag & 6)


def test_ell_long_read_wide_band(dtw, oracle_port):
    """SNP log-likelihoods on a 900-base read with bandwidth 400: more than 14 lane hand-overs per lane
    in the sweeps (the fast sweeps once compounded mantissas along the lanes) and a descriptor window
    that wraps many times."""
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    rng = np.random.default_rng(4242)
    c = synthetic.make_dp_case(rng, model, R=900, bandwidth=400, jitter=40, anchor_density=0.5)
    a = (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'], 400, 2)
    for w in (True, False):
        got = np.asarray(dtw.estimate_log_likelihoods(*a, mg, w))
        exp = np.asarray(oracle_port.estimate_log_likelihoods(*a, mo, w))
        assert np.array_equal(np.isneginf(got), np.isneginf(exp)) and not np.any(np.isnan(got))
        fin = np.isfinite(exp)
        assert np.allclose(got[fin], exp[fin], rtol=1e-9, atol=1e-9)


def test_base_codes_outside_the_alphabet_are_refused(dtw, oracle_port):
    """A base code < 0 or >= alphabet in the reference or in either context would index the k-mer table
    out of bounds (the reference does exactly that, kmer_model.cpp:22-30): the C-ABI refuses the read
    (NVK_READ_BAD_INPUT -> ValueError), the other reads of the batch are served, and get_expected_signal
    raises."""
    from nadavca_amd import synthetic
    model = synthetic.synth_model_arrays(9, k=4, central=1)
    m = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    rng = np.random.default_rng(907)
    good = synthetic.make_dp_case(rng, model, R=25, bandwidth=15, dwell=(3, 6), jitter=2)
    tup = lambda c: (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
    for field, pos, code in (('reference', 7, 4), ('reference', 0, -1), ('context_before', 0, 4),
                             ('context_after', 1, -1), ('context_after', 0, 1 << 30)):
        bad = dict(good)
        v = np.array(good[field]).copy()
        v[pos] = code
        bad[field] = v
        for tr in (True, False):
            with pytest.raises(ValueError):
                dtw.refine_alignment(*tup(bad), 15, 2, m, tr)
            out, st = dtw.refine_alignment_batch([tup(good), tup(bad), tup(good)], 15, 2, m, tr,
                                                 on_error='status', return_status=True)
            assert st.tolist() == [0, dtw.READ_BAD_INPUT, 0] and len(out[1]) == 0
            exp = oracle_port.refine_alignment(*tup(good), 15, 2, mo, tr)
            assert np.array_equal(out[0], exp) and np.array_equal(out[2], exp)
        with pytest.raises(ValueError):
            dtw.estimate_log_likelihoods(*tup(bad), 15, 2, m, True)
        with pytest.raises(ValueError):
            m.get_expected_signal(bad['reference'], bad['context_before'], bad['context_after'])


def test_one_over_wide_read_does_not_fail_the_batch(dtw, oracle_port):
    """A config-2-sized read WITHOUT anchors has the whole matrix as its band (~4 300 cells per row, wavefront
    skew ~ 68): more than one wave's LDS rings hold.  It gets its own status (NVK_READ_TOO_WIDE) and the
    normal reads of the same batch come out as if it were not there."""
    from nadavca_amd import synthetic, _lib
    model = synthetic.load_model_arrays()
    m = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    cases = [synthetic.make_dp_case(np.random.default_rng([908, i]), model, R=400, bandwidth=150) for i in range(5)]
    tup = lambda c: (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
    wide = dict(cases[2], approximate_alignment=np.zeros((0, 2), dtype=np.int32))
    reads = [tup(cases[0]), tup(cases[1]), tup(wide), tup(cases[3]), tup(cases[4])]
    for tr in (True, False):
        out, st = dtw.refine_alignment_batch(reads, 150, 2, m, tr, on_error='status', return_status=True)
        assert st.tolist() == [0, 0, dtw.READ_TOO_WIDE, 0, 0]
        for i in (0, 1, 3, 4):
            exp = oracle_port.refine_alignment(*tup(cases[i]), 150, 2, mo, tr)
            assert np.array_equal(out[i], exp)
        with pytest.raises(_lib.NadavcaHipError):
            dtw.refine_alignment_batch(reads, 150, 2, m, tr)
    ll, st = dtw.estimate_log_likelihoods_flat(dtw.FlatBatch(reads), 150, 2, m, True, on_error='status')
    assert st.tolist() == [0, 0, dtw.READ_TOO_WIDE, 0, 0]
    exp = oracle_port.estimate_log_likelihoods(*tup(cases[4]), 150, 2, mo, True)
    off = np.concatenate([[0], np.cumsum([len(c['reference']) for c in (cases[0], cases[1], wide, cases[3], cases[4])])])
    assert np.allclose(ll[off[4]:off[5]], exp, rtol=1e-9, atol=1e-9)


def test_workspace_limit_changes_nothing_but_the_footprint(dtw, oracle_port):
    """nvk_ctx_set_workspace_limit caps the spill the resident waves may take; fewer reads are in flight,
    results are the same."""
    from nadavca_amd import synthetic, _lib
    model = synthetic.load_model_arrays()
    ctx = _lib.Context(0)
    m = dtw.KmerModel(*model, context=ctx)
    batch = synthetic.make_batch(300, model, seed=909, R=200, R_spread=20, bandwidth=100)
    reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
             for c in batch.cases]
    free = dtw.refine_alignment_batch(reads, 100, 2, m, True)
    ctx.set_workspace_limit(32 << 20)   # 32 MB: a couple of dozen resident waves
    capped = dtw.refine_alignment_batch(reads, 100, 2, m, True)
    ctx.set_workspace_limit(0)
    assert all(np.array_equal(a, b) for a, b in zip(free, capped))


def test_little_free_memory_is_served_in_chunks(dtw):
    """With most of the HBM taken by somebody else (another rank's tensors, another process) the spill cap follows
    what is actually free — no 48 GB floor — and a batch whose spill does not fit is served in chunks of launch
    positions (halved again if an allocation still fails).  Same results as with the memory free."""
    import torch
    from nadavca_amd import synthetic, _lib
    from nadavca_amd.device import DeviceBatch, refine_alignment_dev
    model = synthetic.load_model_arrays()
    ctx = _lib.Context(0)
    m = dtw.KmerModel(*model, context=ctx)
    batch = synthetic.make_batch(1500, model, seed=910, R=400, R_spread=40, bandwidth=150)   # ~3.7 GB of spill
    dev = torch.device('cuda', 0)
    db = DeviceBatch(batch, dev)
    free_ev, free_st = refine_alignment_dev(db, 150, 2, m, True)
    want = free_ev.cpu().numpy().copy()
    del m, ctx                                   # (its workspaces go back to the driver)
    ctx = _lib.Context(0)
    m = dtw.KmerModel(*model, context=ctx)
    torch.cuda.synchronize()
    free_b, _ = torch.cuda.mem_get_info(dev)
    keep = 3 << 30                               # leave 3 GB: the cap becomes ~1.8 GB, i.e. chunks of ~700 reads
    hog = torch.empty(max(free_b - keep, 0), dtype=torch.uint8, device=dev) if free_b > keep + (1 << 30) else None
    try:
        ev, st = refine_alignment_dev(db, 150, 2, m, True)
        assert not bool(st.any())
        assert np.array_equal(ev.cpu().numpy(), want)
    finally:
        del hog
        torch.cuda.empty_cache()


@pytest.mark.parametrize('it,case', [(1630, 2), (2828, 2)])
def test_tiny_posteriors_at_the_end_of_a_short_wide_band_read(dtw, oracle_port, it, case):
    """tests/dev/fuzz_team.py seed 31415: reads of 85 and 92 bases whose band is wider than the read, min event
    length 0, random model — the best path ends in several zero-length events with posteriors of ~2^-200 each, so
    with every row of the read in flight at once the last rows' path scores lie > 900 bits below the first rows'.
    Round 3's experiment with path scores under ONE wave-uniform scale flushed them and placed the last events
    hundreds of samples late WITHOUT a tie bit; an exponent per score gives the reference's rows."""
    from fuzz_cases import make_team_batch, reads_of
    fb = make_team_batch(31415, it)
    mg = dtw.KmerModel(*fb['model'])
    mo = oracle_port.KmerModel(*fb['model'])
    got = dtw.refine_alignment_batch(reads_of(fb['cases']), fb['bw'], fb['mel'], mg, fb['tr'])
    c = fb['cases'][case]
    exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                       c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
    assert np.array_equal(got[case], exp)


def test_a_tie_of_the_long_double_reference_that_the_engine_resolves(dtw, oracle_port):
    """tests/dev/fuzz_parity.py seed 20261006, iteration 8718, read 0 (sharp model, bandwidth 4, samples clipped to
    -5): one event end differs by a sample, the double-precision and the long-double reference agree with each
    other — and the long-double reference has an EXACT tie between the path scores it compares at that base, the
    only such base of the read.  The reference adds unnormalised log posteriors (node.cpp:39-50): its scores are
    ~5e6 there, where 80-bit arithmetic resolves 1e-12 at best, so it keeps the first maximum; the engine's scores
    are products of normalised posteriors and tell the cells apart.  The read carries a tie bit of the ULP / NEAR
    kind (the contract of include/nadavca_hip.h), and the classifier accepts the difference only because of the
    referee's own tie (tests/fuzz_cases.py: 'referee-tie')."""
    from fuzz_cases import classify_difference, near_tie_bases
    from nadavca_amd._lib import TIE_ULP, TIE_NEAR
    fb, reads = _fuzz_batch(20261006, 8718)
    mg = dtw.KmerModel(*fb['model'])
    mo = oracle_port.KmerModel(*fb['model'])
    got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])
    flags = mg.context.last_tie_flags(len(reads))
    c = fb['cases'][0]
    exp = np.asarray(oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                                  c['approximate_alignment'], fb['bw'], fb['mel'], mo,
                                                  fb['tr'])).reshape(-1, 2)
    ev = np.asarray(got[0]).reshape(-1, 2)
    rows = np.nonzero((ev != exp).any(axis=1))[0]
    assert rows.tolist() == [122] and abs(int(ev[122, 1]) - int(exp[122, 1])) == 1
    assert flags[0] & (TIE_ULP | TIE_NEAR)
    assert set(near_tie_bases(c, fb['model'], fb['bw'], fb['mel'], fb['tr']).tolist()) == {122, 123}
    assert classify_difference(ev, exp, c, fb['model'], fb['k'], fb['central'], fb['alphabet'], fb['bw'], fb['mel'],
                               fb['tr']) == 'referee-tie'
