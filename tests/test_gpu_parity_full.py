"""GPU parity at BASELINE.json's full sizes, against the CPU oracle on EVERY read (16 oracle threads; the
C oracle releases the GIL inside its ctypes calls):

* config 2 — the very batch bench.py times (10 000 reads, seed 1000), transitions on and off;
* the same shape as a sequencer delivers it: int16 ADC counts through the api_align_signal workload (device
  normalisation, both alignment passes of align_signal) and config 2 quantised to ADC steps, 10 000 reads each;
* the parity contract of include/nadavca_hip.h: a read with neither NVK_TIE_ULP nor NVK_TIE_NEAR has the
  reference's events exactly — exact ties (NVK_TIE_EXACT) included; a read that differs carries such a bit AND
  its difference is explained (flat plateau between equal k-mer levels / the long-double referee
  sides with the engine / the reference changes its own answer in long double) — checked on a
  homopolymer-rich reference, on randomised models and on wide bands;
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


TIE_EXACT, TIE_NEAR, TIE_ULP, TIE_PLATEAU = 1, 2, 4, 8   # include/nadavca_hip.h
TIE_LOOSE = TIE_NEAR | TIE_ULP           # the contract: without these two bits a read equals the reference


def _tie_counts(flags):
    """reads per class: (exact, ulp, near)"""
    return tuple(int(((flags & b) != 0).sum()) for b in (TIE_EXACT, TIE_ULP, TIE_NEAR))


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
        st = ctx.last_batch_stats()
        n_x, n_u, n_n = _tie_counts(flags)
        assert int(((flags & 7) != 0).sum()) == st['reads_tie_ambiguous']
        assert (n_x, n_u, n_n) == (st['reads_tie_exact'], st['reads_tie_ulp'], st['reads_tie_near'])
        assert (n_u + n_n) * 100 < n, 'ties on %d of %d reads' % (n_u + n_n, n)
        print('cfg2 transitions=%s: 0 of %d reads differ; reads with tie bits: %d exact, %d ulp, %d near; plateau mark: %d'
              % (tr, n, n_x, n_u, n_n, int(((flags & TIE_PLATEAU) != 0).sum())))
        # ... and against the reference's OWN code (oracle/_ref, compiled in place from /root/reference and
        # shipped to the GPU box) on a 500-read slice
        from oracle.oracle import Oracle, have_reference
        if have_reference():
            oref = Oracle('reference')
            mr = oref.KmerModel(*model)
            sl = batch.cases[:500]
            expr = _oracle_refine(oref, mr, sl, wl['bandwidth'], 2, tr)
            assert all(_same(got[i], expr[i]) for i in range(len(sl))), 'differs from oracle/_ref'
            print('cfg2 transitions=%s: 500-read slice equals oracle/_ref (the compiled reference)' % tr)


def test_homopolymer_rich_reference_mismatches_are_flagged(oracle_port):
    """A reference with many runs of >= k+1 equal bases (adjacent identical k-mers: the boundary between
    their events is a flat posterior plateau).  Contract: events equal the reference's wherever the near-tie
    bit is 0; a read that differs is explained by the classification of fuzz_cases.classify_difference."""
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
        assert not np.any(diff & ((flags & TIE_LOOSE) == 0)), 'a read without a tie bit differs from the reference'
        from fuzz_cases import classify_difference
        for j in np.nonzero(diff)[0]:
            why = classify_difference(got[j], exp[j], cases[j], model, k, central, alphabet, 100, 2, tr)
            assert why != 'UNEXPLAINED', (tr, j)
        n_x, n_u, n_n = _tie_counts(flags)
        print('homopolymer-rich, transitions=%s: %d reads, %d adjacent identical k-mer pairs, reads with tie bits %d exact / '
              '%d ulp / %d near, %d differ from the double-precision reference (%d of them without the ulp bit; all explained)'
              % (tr, len(reads), n_pairs, n_x, n_u, n_n, int(diff.sum()), int((diff & ((flags & TIE_ULP) == 0)).sum())))


def _fuzz_contract(make_batch, seed, iters, oracle_port, label, max_diff_share):
    """Shared body of the randomised contract tests: equality wherever the near-tie bit is 0 (exact ties
    included), every difference explained, bounds on the near-tie share and on the differing share."""
    from fuzz_cases import reads_of, classify_difference
    from nadavca_amd import dtw, _lib
    ctx = _lib.default_context()
    n_reads = n_diff = n_x = n_n = n_u = n_diff_no_ulp = n_p = 0
    why_count = {}
    for it in range(iters):
        fb = make_batch(seed, it)
        mg = dtw.KmerModel(*fb['model'], context=ctx)
        mo = oracle_port.KmerModel(*fb['model'])
        reads = reads_of(fb['cases'])
        got = dtw.refine_alignment_batch(reads, fb['bw'], fb['mel'], mg, fb['tr'])
        flags = ctx.last_tie_flags(len(reads))
        k, central, alphabet = fb['model'][0], fb['model'][1], fb['model'][2]
        for j, c in enumerate(fb['cases']):
            exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                               c['approximate_alignment'], fb['bw'], fb['mel'], mo, fb['tr'])
            same = _same(got[j], exp)
            n_reads += 1
            n_x += int((flags[j] & TIE_EXACT) != 0)
            n_n += int((flags[j] & TIE_NEAR) != 0)
            n_u += int((flags[j] & TIE_ULP) != 0)
            n_p += int((flags[j] & TIE_PLATEAU) != 0)
            if not same:
                n_diff += 1
                n_diff_no_ulp += int((flags[j] & TIE_ULP) == 0)
                assert flags[j] & TIE_LOOSE, ('a read without a tie bit differs', it, j)
                why = classify_difference(got[j], exp, c, fb['model'], k, central, alphabet, fb['bw'], fb['mel'],
                                          fb['tr'])
                assert why != 'UNEXPLAINED', ('unexplained difference', it, j)
                if why == 'flat-plateau':   # ... and the read is marked as holding such a boundary
                    assert flags[j] & TIE_PLATEAU, ('flat-plateau difference without the plateau mark', it, j)
                why_count[why] = why_count.get(why, 0) + 1
        mg.close()
    assert n_diff <= max_diff_share * n_reads, (n_diff, n_reads)
    print('%s: %d reads, reads with tie bits %d exact / %d ulp / %d near, %d with the plateau mark; %d differ (%d of them '
          'without the ulp bit) %s' % (label, n_reads, n_x, n_u, n_n, n_p, n_diff, n_diff_no_ulp, why_count))


def test_random_models_contract(oracle_port):
    """Randomised k-mer models (k 2-6: few levels, so equal adjacent levels are common), alphabets, min event
    lengths and bandwidths (the generator of tests/dev/fuzz_parity.py)."""
    from fuzz_cases import make_fuzz_batch
    _fuzz_contract(make_fuzz_batch, 20261004, 400, oracle_port, 'random models', 0.06)


def test_wide_band_reads_swept_by_teams_of_waves(oracle_port):
    """Bands too wide for one wave's rings (skew above the main launch's cap) are swept by teams of four waves
    (kernels_align3.hip, W > 1): random bandwidths 100-700 on reads of 1-700 bases, min event length 0-4,
    transitions on/off, narrow reads mixed in (tests/dev/fuzz_team.py runs the same generator for minutes)."""
    from fuzz_cases import make_team_batch
    _fuzz_contract(make_team_batch, 77, 120, oracle_port, 'wide bands (teams)', 0.04)


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


def test_config2_quantised_to_adc_steps_every_read_equals_the_oracle(oracle_port):
    """Config 2 at full size with the signals quantised to ADC steps (12 counts per unit, as the raw signals of
    synthetic.make_read_spec): a read then holds each value many times, equal samples make equal densities, and
    path scores tie exactly, or almost, at some comparison of almost every read — for the engine and for the
    reference alike (tests/dev/ref_tie_histogram.py counts them in the reference's own arithmetic).  Every read
    equals the oracle all the same; the per-class rates are printed (DESIGN.md 2.1 quotes them)."""
    from nadavca_amd import dtw, synthetic, _lib
    model = synthetic.load_model_arrays()
    ctx = _lib.default_context()
    mg = dtw.KmerModel(*model, context=ctx)
    mo = oracle_port.KmerModel(*model)
    wl = dict(synthetic.WORKLOADS['cfg2_align'])
    n = wl.pop('n_reads')
    batch = synthetic.make_batch(n, model, seed=1000, **wl)
    for c in batch.cases:
        c['signal'] = np.round(c['signal'] * 12.0) / 12.0
    batch = synthetic.Batch(batch.cases)
    flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off,
                                     batch.context_before, batch.cb_off, batch.context_after, batch.ca_off,
                                     batch.anchors, batch.anc_off)
    for tr in (True, False):
        got = dtw.refine_alignment_batch(flat, wl['bandwidth'], 2, mg, tr)
        flags = ctx.last_tie_flags(n)
        exp = _oracle_refine(oracle_port, mo, batch.cases, wl['bandwidth'], 2, tr)
        diff = [i for i in range(n) if not _same(got[i], exp[i])]
        n_x, n_u, n_n = _tie_counts(flags)
        print('cfg2 quantised, transitions=%s: %d of %d reads differ; reads with tie bits: %d exact, %d ulp, %d near; '
              'plateau mark: %d' % (tr, len(diff), n, n_x, n_u, n_n, int(((flags & TIE_PLATEAU) != 0).sum())))
        assert diff == [], (tr, diff[:5])


def test_int16_api_workload_every_read_equals_the_oracle(oracle_port):
    """bench.py's api_align_signal workload at full size — 10 000 simulated reads as int16 ADC counts, per-read
    median/MAD normalisation ON THE DEVICE, approximate-alignment stage, window cutting — and both alignment
    passes of align_signal (the second on the linearly re-fitted signal): every (event_start, event_end) row of
    every read against the oracle run on the very windows the kernels saw (copied back from the device)."""
    import torch
    from nadavca_amd import dtw, synthetic, _lib, readbatch
    from nadavca_amd.device import (DeviceBatch, normalize_groups_dev, refine_alignment_dev, expected_levels_dev,
                                    event_means_dev, linfit_rescale_dev)
    model = synthetic.load_model_arrays()
    ctx = _lib.default_context()
    mg = dtw.KmerModel(*model, context=ctx)
    mo = oracle_port.KmerModel(*model)
    n, bw, mel = 10000, 150, 2
    rb, aligner, _ = synthetic.make_read_batch(n, model, seed=1000, genome_length=10000)
    assert rb.raw_signal.dtype == np.int16
    device = torch.device('cuda', ctx.device)
    raw = torch.from_numpy(rb.raw_signal).to(device).to(torch.float64)
    sig_off_dev = torch.from_numpy(rb.sig_off).to(device)
    norm, _ = normalize_groups_dev(ctx, raw, sig_off_dev, out=raw)
    sa = readbatch.signal_alignments(rb, aligner.get_base_alignments(rb), bw, aligner.reference_num, model[0],
                                     model[1], device=device)
    db = DeviceBatch.from_windows(norm, sa, device)
    assert db.n == n

    def host_cases():
        sig, so = db.signal.cpu().numpy(), db.sig_off.cpu().numpy()
        ref, ro = db.reference.cpu().numpy(), db.ref_off.cpu().numpy()
        cb, cbo = db.context_before.cpu().numpy(), db.cb_off.cpu().numpy()
        ca, cao = db.context_after.cpu().numpy(), db.ca_off.cpu().numpy()
        anc, ao = db.anchors.cpu().numpy().reshape(-1, 2), db.anc_off.cpu().numpy()
        return [dict(signal=sig[so[i]:so[i + 1]], reference=ref[ro[i]:ro[i + 1]], context_before=cb[cbo[i]:cbo[i + 1]],
                     context_after=ca[cao[i]:cao[i + 1]], approximate_alignment=anc[ao[i]:ao[i + 1]])
                for i in range(n)], ro

    for rnd in (0, 1):
        cases, ro = host_cases()
        ev, st = refine_alignment_dev(db, bw, mel, mg, True)
        flags = ctx.last_tie_flags(n)
        ev, st = ev.cpu().numpy(), st.cpu().numpy()
        assert not st.any()
        exp = _oracle_refine(oracle_port, mo, cases, bw, mel, True)
        diff = [i for i in range(n) if not _same(ev[ro[i]:ro[i + 1]], exp[i])]
        n_x, n_u, n_n = _tie_counts(flags)
        print('int16 api workload, alignment pass %d: %d of %d reads differ; reads with tie bits: %d exact, %d ulp, %d near'
              % (rnd + 1, len(diff), n, n_x, n_u, n_n))
        assert diff == [], (rnd, diff[:5])
        if rnd == 0:   # align_signal.py:59-76: linear re-fit on the device, then the second pass
            d_ev, d_st = torch.from_numpy(ev).to(device), torch.from_numpy(st).to(device)
            expected = expected_levels_dev(db, mg, with_contexts=False)
            means = event_means_dev(db, ctx, d_ev, d_st)
            linfit_rescale_dev(db, ctx, expected, means, d_st)


def test_port_equals_compiled_reference_on_the_gpu_box(oracle_port):
    """The checker of this suite is the C restatement (oracle/liboracle.so); where oracle/_ref — the reference's
    own sources compiled in place — travelled to this box, the two must agree bit for bit HERE as well (this
    box's libm and CPU), not only in the build container (tests/test_oracle_golden.py runs the same check there)."""
    from oracle.oracle import have_reference
    if not have_reference():
        pytest.skip('oracle/_ref did not travel to this box')
    from test_oracle_golden import test_port_bit_identical_to_compiled_reference_when_present as check
    check(oracle_port)
