"""Re-create iteration IT of tools/fuzz_parity.py (seed SEED) and print the deviations in detail.
usage: fuzz_repro.py SEED IT [ell|align]"""
import sys, os
import numpy as np
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle
seed0, it = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else 'both'
o = Oracle('port')
rng = np.random.default_rng([seed0, it])
k = int(rng.integers(2, 7)); central = int(rng.integers(0, k))
alphabet = int(rng.choice([4, 4, 4, 3, 5]))
model = synthetic.synth_model_arrays(int(rng.integers(1 << 30)), k=k, central=central, alphabet=alphabet)
if rng.random() < 0.3:
    model = model[:4] + (model[4] * float(rng.choice([0.3, 3.0])),)
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
mel = int(rng.integers(0, 5)); bw = int(rng.integers(4, 90))
cases = []
for i in range(int(rng.integers(1, 10))):
    R = int(rng.integers(1, 260))
    cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(4, 90)),
                                        dwell=(max(mel, 1), int(rng.integers(max(mel, 1) + 1, 14))),
                                        noise=float(rng.choice([0.1, 0.35, 1.0])), jitter=int(rng.integers(0, 25)),
                                        anchor_density=float(rng.uniform(0.05, 1.0)), with_context=bool(rng.integers(2)),
                                        trim=min(3, R // 3)))
reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in cases]
tr, w = bool(rng.integers(2)), bool(rng.integers(2))
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
