"""Randomised parity run: refine_alignment (exact equality) and estimate_log_likelihoods (1e-9) through
the C-ABI against the CPU oracle, over random k-mer models, min event lengths, bandwidths, read shapes and
flags.  usage: fuzz_parity.py SECONDS [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle, LongDoubleReferee
from fuzz_cases import make_fuzz_batch, reads_of, classify_difference

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = Oracle('port')
t_end = time.time() + budget
t_tick = time.time() + 60
n_batches = n_reads = n_bad = n_tie = 0
kinds = {}
it = 0
while time.time() < t_end:
    fb = make_fuzz_batch(seed0, it); it += 1
    model, k, central, alphabet, mel, bw, cases, tr, w = (fb[x] for x in ('model', 'k', 'central', 'alphabet', 'mel', 'bw', 'cases', 'tr', 'w'))
    mg = dtw.KmerModel(*model); mo = o.KmerModel(*model); ld = None
    reads = reads_of(cases)
    n_batches += 1; n_reads += len(cases)
    if time.time() > t_tick:
        t_tick = time.time() + 60
        print('... %d batches, %d reads, %d differ, %d unexplained' % (n_batches, n_reads, n_bad, n_bad - n_tie), file=sys.stderr, flush=True)
    got = dtw.refine_alignment_batch(reads, bw, mel, mg, tr)
    for ci, (c, ev) in enumerate(zip(cases, got)):
        exp = np.asarray(o.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                            c['approximate_alignment'], bw, mel, mo, tr)).reshape(-1, 2)
        if np.asarray(ev).reshape(-1, 2).shape != exp.shape or not np.array_equal(np.asarray(ev).reshape(-1, 2), exp):
            n_bad += 1
            # Who is right?  tests/fuzz_cases.classify_difference (shared with the -m gpu tests): flat plateau
            # between equal k-mer levels / the long-double reference sides with the engine / the reference changes
            # its own answer in long double / the long-double reference itself has a tie at that base / none.
            ev2 = np.asarray(ev).reshape(-1, 2)
            rows = np.nonzero((ev2 != exp).any(axis=1))[0] if ev2.shape == exp.shape else np.array([-1])
            if ld is None:
                ld = LongDoubleReferee(*model)
            why = classify_difference(ev, exp, c, model, k, central, alphabet, bw, mel, tr, referee=ld)
            kinds[why] = kinds.get(why, 0) + 1
            n_tie += int(why != 'UNEXPLAINED')
            print('ALIGN MISMATCH', why, 'rows', rows[:6].tolist(), 'it', it - 1, 'case', ci, 'k', k, 'central', central, 'alphabet', alphabet, 'mel', mel, 'bw', bw, 'tr', tr,
                  'R', len(c['reference']), 'N', len(c['signal']), flush=True)
    got = dtw.estimate_log_likelihoods_batch(reads, bw, mel, mg, w)
    for ci, (c, ll) in enumerate(zip(cases, got)):
        exp = np.asarray(o.estimate_log_likelihoods(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                                    c['approximate_alignment'], bw, mel, mo, w))
        fin = np.isfinite(exp)
        ok = ll.shape == exp.shape and np.array_equal(np.isneginf(ll), np.isneginf(exp)) and not np.any(np.isnan(ll)) \
            and np.allclose(ll[fin], exp[fin], rtol=1e-9, atol=1e-9)
        if not ok:
            n_bad += 1
            print('ELL MISMATCH it', it - 1, 'case', ci, 'k', k, 'central', central, 'alphabet', alphabet, 'mel', mel, 'bw', bw, 'w', w,
                  'R', len(c['reference']), 'N', len(c['signal']), flush=True)
print('fuzz: %d batches, %d reads; %d reads differ from the double reference, %d of them explained '
      '(%d: only on flat plateaus between equal k-mer levels; %d: the long-double reference sides with the engine; '
      '%d: the reference changes its own answer in long double; %d: the long-double reference has a tie at that base)'
      % (n_batches, n_reads, n_bad, n_tie, kinds.get('flat-plateau', 0), kinds.get('reference-rounding', 0),
         kinds.get('precision-decided', 0), kinds.get('referee-tie', 0)))
