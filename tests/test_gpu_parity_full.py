"""GPU parity at BASELINE.json's full sizes, against the CPU oracle on EVERY read (16 oracle threads; the
C oracle releases the GIL inside its ctypes calls):

* config 2 — the very batch bench.py times (10 000 reads, seed 1000), transitions on and off;
* the parity contract of include/nadavca_hip.h: a read whose events differ from the reference's must carry
  the tie flag (a path comparison fell inside the tolerance band) — checked on a homopolymer-rich
  reference, where flat posterior plateaus are frequent, and on randomised models;
* config 5 shape — ~50 000-sample reads, ~5 000 bases, bandwidth 1000 — through refine_alignment and
  estimate_log_likelihoods.

Reference semantics: /root/reference/nadavca/dtw/dtw.cpp:133-228, node.cpp:39-91 (restated in oracle/)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))


def _reads(cases):
    return [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
            for c in cases]


def _oracle_refine(o, mo, cases, bw, mel, tr):
    with ThreadPoolExecutor(THREADS) as ex:
        return list(ex.map(lambda c: o.refine_alignment(c['signal'], c['reference'], c['context_before'],
                                                        c['context_after'], c['approximate_alignment'],
                                                        bw, mel, mo, tr), cases))


def _same(ev, exp):
    ev, exp = np.asarray(ev).reshape(-1, 2), np.asarray(exp).reshape(-1, 2)
    return ev.shape == exp.shape and np.array_equal(ev, exp)


def test_config2_bench_batch_every_read_equals_the_oracle(oracle_port):
    """All 10 000 reads of bench.py's default workload (seed 1000 = rank 0), both row layouts:
    exact equality of every (event_start, event_end), and the tie-flag count bench.py reports."""
    from nadavca_amd import dtw, synthetic, _lib
    model = synthetic.load_model_arrays()
    ctx = _lib.default_context()
    mg = dtw.KmerModel(*model, context=ctx)
    mo = oracle_port.KmerModel(*model)
    wl = dict(synthetic.WORKLOADS['cfg2_align'])
    n = wl.pop('n_reads')
    batch = synthetic.make_batch(n, model, seed=1000, **wl)
    flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off,
                                     batch.context_before, batch.cb_off, batch.context_after, batch.ca_off,
                                     batch.anchors, batch.anc_off)
    for tr in (True, False):
        got = dtw.refine_alignment_batch(flat, wl['bandwidth'], 2, mg, tr)
        flags = ctx.last_tie_flags(n)
        exp = _oracle_refine(oracle_port, mo, batch.cases, wl['bandwidth'], 2, tr)
        diff = [i for i in range(n) if not _same(got[i], exp[i])]
        assert diff == [], 'transitions=%s: %d of %d reads differ from the oracle, first %s' % (tr, len(diff), n, diff[:5])
        assert int((flags != 0).sum()) == ctx.last_batch_stats()['reads_tie_ambiguous']
        print('cfg2 transitions=%s: 0 of %d reads differ; %d carry the tie flag' % (tr, n, int((flags != 0).sum())))


def test_homopolymer_rich_reference_mismatches_are_flagged(oracle_port):
    """A reference with many runs of >= k+1 equal bases (adjacent identical k-mers: the boundary between
    their events is a flat posterior plateau).  Contract: events equal the reference's wherever the tie
    flag is 0; the mismatch rate on flagged reads is what DESIGN.md 2.1 quotes."""
    from nadavca_amd import dtw, synthetic, _lib
    model = synthetic.load_model_arrays()
    ctx = _lib.default_context()
    mg = dtw.KmerModel(*model, context=ctx)
    mo = oracle_port.KmerModel(*model)
    cases = []
    for i in range(1500):
        rng = np.random.default_rng([4711, i])
        cases.append(synthetic.make_dp_case(rng, model, R=int(rng.integers(150, 320)), bandwidth=100,
                                            bases=synthetic.homopolymer_rich))
    reads = _reads(cases)
    k, central, alphabet = model[0], model[1], model[2]
    n_pairs = 0
    for c in cases:
        ext = np.concatenate([c['context_before'], c['reference'], c['context_after']]).astype(np.int64)
        ids = synthetic.kmer_ids(ext, len(c['context_before']), len(c['reference']), k, central, alphabet)
        n_pairs += int((ids[1:] == ids[:-1]).sum())
    assert n_pairs > 2000   # the fixture really is rich in adjacent identical k-mers
    for tr in (True, False):
        got = dtw.refine_alignment_batch(reads, 100, 2, mg, tr)
        flags = ctx.last_tie_flags(len(reads))
        exp = _oracle_refine(oracle_port, mo, cases, 100, 2, tr)
        diff = np.array([not _same(g, e) for g, e in zip(got, exp)])
        assert not np.any(diff & (flags == 0)), 'an unflagged read differs from the reference'
        print('homopolymer-rich, transitions=%s: %d reads, %d adjacent identical k-mer pairs, %d flagged, '
              '%d differ from the double-precision reference (all flagged)'
              % (tr, len(reads), n_pairs, int((flags != 0).sum()), int(diff.sum())))


def test_random_models_mismatches_are_flagged(oracle_port):
    """Randomised k-mer models, alphabets, min event lengths and bandwidths (the generator of
    tests/dev/fuzz_parity.py): an unflagged read never differs from the reference."""
    from fuzz_cases import make_fuzz_batch, reads_of
    from nadavca_amd import dtw, _lib
    ctx = _lib.default_context()
    n_reads = n_diff = n_flag = 0
    for it in range(400):
        fb = make_fuzz_batch(20261004, it)
        mg = dtw.KmerModel(*fb['model'], context=ctx)
        mo = oracle_port.KmerModel(*fb['model'])
        reads = reads_of(fb['cases'])
        got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])
        flags = ctx.last_tie_flags(len(reads))
        for j, c in enumerate(fb['cases']):
            exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                               c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
            same = _same(got[j], exp)
            n_reads += 1
            n_diff += (not same)
            n_flag += int(flags[j] != 0)
            assert same or flags[j] != 0, ('unflagged read differs', it, j)
        mg.close()
    print('random models: %d reads, %d flagged, %d differ (all flagged)' % (n_reads, n_flag, n_diff))


def test_wide_band_reads_swept_by_teams_of_waves(oracle_port):
    """Bands too wide for one wave's rings (skew above the main launch's cap) are swept by teams of four waves
    (kernels_align3.hip, W > 1): random bandwidths 100-700 on reads of 1-700 bases, min event length 0-4,
    transitions on/off, narrow reads mixed in (tests/dev/fuzz_team.py runs the same generator for minutes).
    Every read equals the reference or carries the tie flag; most are equal."""
    from fuzz_cases import make_team_batch, reads_of
    from nadavca_amd import dtw, _lib
    ctx = _lib.default_context()
    n_reads = n_diff = n_flag = 0
    for it in range(120):
        fb = make_team_batch(77, it)
        mg = dtw.KmerModel(*fb['model'], context=ctx)
        mo = oracle_port.KmerModel(*fb['model'])
        reads = reads_of(fb['cases'])
        got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])
        flags = ctx.last_tie_flags(len(reads))
        for j, c in enumerate(fb['cases']):
            exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                               c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
            same = _same(got[j], exp)
            n_reads += 1
            n_diff += (not same)
            n_flag += int(flags[j] != 0)
            assert same or flags[j] != 0, ('unflagged read differs', it, j)
        mg.close()
    assert n_diff * 10 < n_reads
    print('wide bands (teams): %d reads, %d flagged, %d differ (all flagged)' % (n_reads, n_flag, n_diff))


@pytest.fixture(scope='module')
def cfg5_cases():
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    wl = dict(synthetic.WORKLOADS['cfg5_long'])
    wl.pop('n_reads')
    batch = synthetic.make_batch(2, model, seed=55, **wl)
    return model, wl, batch


def test_config5_shape_refine_equals_the_oracle(oracle_port, cfg5_cases):
    """BASELINE config 5 shape at full size: 2 reads of ~50 000 samples, ~5 000 bases, bandwidth 1000
    (skew ~ 30, the wide-band launch class), transitions on and off, every event against the oracle."""
    from nadavca_amd import dtw
    model, wl, batch = cfg5_cases
    assert min(len(c['signal']) for c in batch.cases) > 40000
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    reads = _reads(batch.cases)
    jobs = [(c, tr) for tr in (True, False) for c in batch.cases]
    with ThreadPoolExecutor(min(THREADS, len(jobs))) as ex:
        exp = list(ex.map(lambda j: oracle_port.refine_alignment(
            j[0]['signal'], j[0]['reference'], j[0]['context_before'], j[0]['context_after'],
            j[0]['approximate_alignment'], wl['bandwidth'], 2, mo, j[1]), jobs))
    for t, tr in enumerate((True, False)):
        got = dtw.refine_alignment_batch(reads, wl['bandwidth'], 2, mg, tr)
        for i in range(len(reads)):
            assert _same(got[i], exp[t * len(reads) + i]), (tr, i)


def test_config5_shape_log_likelihoods_equal_the_oracle(oracle_port, cfg5_cases):
    """One config-5-shaped read through estimate_log_likelihoods (wobbling on): <= 1e-9 relative, the same
    -inf pattern."""
    from nadavca_amd import dtw
    model, wl, batch = cfg5_cases
    mg = dtw.KmerModel(*model)
    mo = oracle_port.KmerModel(*model)
    c = batch.cases[0]
    a = (c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'],
         wl['bandwidth'], 2)
    got = np.asarray(dtw.estimate_log_likelihoods(*a, mg, True))
    exp = np.asarray(oracle_port.estimate_log_likelihoods(*a, mo, True))
    assert np.array_equal(np.isneginf(got), np.isneginf(exp)) and not np.any(np.isnan(got))
    fin = np.isfinite(exp)
    assert np.allclose(got[fin], exp[fin], rtol=1e-9, atol=1e-9)


def test_quantised_signals_equal_the_oracle_and_are_flagged(oracle_port):
    """Signals as a sequencer delivers them: integer ADC counts, so a read holds each normalised value many
    times.  Equal samples make equal densities, and path scores then tie EXACTLY at some comparison of almost
    every read (the flag fires) — for the engine and for the reference alike.  Such exact ties are resolved by
    "first maximum wins" on both sides: the events still equal the reference's on every read here."""
    from nadavca_amd import dtw, synthetic, _lib
    model = synthetic.load_model_arrays()
    ctx = _lib.default_context()
    mg = dtw.KmerModel(*model, context=ctx)
    mo = oracle_port.KmerModel(*model)
    batch = synthetic.make_batch(600, model, seed=31, R=300, R_spread=30, bandwidth=120)
    for c in batch.cases:
        c['signal'] = np.round(c['signal'] * 12.0) / 12.0        # 12 ADC counts per unit, as synthetic raw signals
    reads = _reads(batch.cases)
    for tr in (True, False):
        got = dtw.refine_alignment_batch(reads, 120, 2, mg, tr)
        flags = ctx.last_tie_flags(len(reads))
        exp = _oracle_refine(oracle_port, mo, batch.cases, 120, 2, tr)
        diff = np.array([not _same(g, e) for g, e in zip(got, exp)])
        print('quantised signals, transitions=%s: %d reads, %d flagged, %d differ from the reference'
              % (tr, len(reads), int((flags != 0).sum()), int(diff.sum())))
        assert not np.any(diff & (flags == 0))
        assert diff.sum() == 0
