"""Re-create iteration IT of tests/dev/fuzz_parity.py (seed SEED) and print the deviations in detail.
usage: fuzz_repro.py SEED IT [ell|align]"""
import sys, os
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle
from fuzz_cases import make_fuzz_batch, reads_of
seed0, it = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else 'both'
o = Oracle('port')
fb = make_fuzz_batch(seed0, it)
model, k, central, alphabet, mel, bw, cases, tr, w = (fb[x] for x in ('model', 'k', 'central', 'alphabet', 'mel', 'bw', 'cases', 'tr', 'w'))
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
reads = reads_of(cases)
print('k', k, 'central', central, 'alphabet', alphabet, 'sigma', model[4][0], 'mel', mel, 'bw', bw, 'tr', tr, 'w', w, 'reads', len(cases),
      'variant', os.environ.get('NADAVCA_ALIGN_KERNEL'), os.environ.get('NADAVCA_ELL_KERNEL'))
if what in ('both', 'align'):
    got = dtw.refine_alignment_batch(reads, bw, mel, mg, tr)
    for ci, (c, ev) in enumerate(zip(cases, got)):
        exp = np.asarray(o.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                            c['approximate_alignment'], bw, mel, mo, tr)).reshape(-1, 2)
        ev2 = np.asarray(ev).reshape(-1, 2)
        if ev2.shape != exp.shape or not np.array_equal(ev2, exp):
            rows = np.nonzero((ev2 != exp).any(axis=1))[0] if ev2.shape == exp.shape else []
            ext = np.concatenate([c['context_before'], c['reference'], c['context_after']]).astype(np.int64)
            ids = synthetic.kmer_ids(ext, len(c['context_before']), len(c['reference']), k, central, alphabet)
            print('ALIGN case', ci, 'R', len(c['reference']), 'N', len(c['signal']), 'shapes', ev2.shape, exp.shape)
            for j in rows[:8]:
                lo, hi = max(j - 2, 0), min(j + 3, len(ids))
                print('  row', j, 'got', ev2[j].tolist(), 'exp', exp[j].tolist(), 'levels', np.round(model[3][ids[lo:hi]], 4).tolist(),
                      'neighbours got', ev2[lo:hi].tolist(), 'exp', exp[lo:hi].tolist())
if what in ('both', 'ell'):
    got = dtw.estimate_log_likelihoods_batch(reads, bw, mel, mg, w)
    for ci, (c, ll) in enumerate(zip(cases, got)):
        exp = np.asarray(o.estimate_log_likelihoods(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                                    c['approximate_alignment'], bw, mel, mo, w))
        fin = np.isfinite(exp)
        ok = ll.shape == exp.shape and np.array_equal(np.isneginf(ll), np.isneginf(exp)) and not np.any(np.isnan(ll)) \
            and np.allclose(ll[fin], exp[fin], rtol=1e-9, atol=1e-9)
        if not ok:
            bad = np.argwhere(~np.isclose(ll, exp, rtol=1e-9, atol=1e-9) & ~(np.isneginf(ll) & np.isneginf(exp)))
            print('ELL case', ci, 'R', len(c['reference']), 'N', len(c['signal']), 'nan', int(np.isnan(ll).sum()),
                  'inf pattern equal', np.array_equal(np.isneginf(ll), np.isneginf(exp)), 'bad cells', len(bad))
            for p, b in bad[:8]:
                print('   pos', p, 'base', b, 'ref base', int(c['reference'][p]), 'got', ll[p, b], 'exp', exp[p, b], 'rel', abs(ll[p, b] - exp[p, b]) / max(1, abs(exp[p, b])))
