"""Randomised parity run: refine_alignment (exact equality) and estimate_log_likelihoods (1e-9) through
the C-ABI against the CPU oracle, over random k-mer models, min event lengths, bandwidths, read shapes and
flags.  usage: fuzz_parity.py SECONDS [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = Oracle('port')
t_end = time.time() + budget
n_batches = n_reads = n_bad = n_tie = 0
it = 0
while time.time() < t_end:
    rng = np.random.default_rng([seed0, it]); it += 1
    k = int(rng.integers(2, 7)); central = int(rng.integers(0, k))
    alphabet = int(rng.choice([4, 4, 4, 3, 5]))
    model = synthetic.synth_model_arrays(int(rng.integers(1 << 30)), k=k, central=central, alphabet=alphabet)
    if rng.random() < 0.3:  # sharper or blunter levels
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
    n_batches += 1; n_reads += len(cases)
    got = dtw.refine_alignment_batch(reads, bw, mel, mg, tr)
    for ci, (c, ev) in enumerate(zip(cases, got)):
        exp = np.asarray(o.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                            c['approximate_alignment'], bw, mel, mo, tr)).reshape(-1, 2)
        if np.asarray(ev).reshape(-1, 2).shape != exp.shape or not np.array_equal(np.asarray(ev).reshape(-1, 2), exp):
            n_bad += 1
            # is every differing boundary one between two bases with the SAME k-mer?
            ev2 = np.asarray(ev).reshape(-1, 2)
            ext = np.concatenate([c['context_before'], c['reference'], c['context_after']]).astype(np.int64)
            ids = synthetic.kmer_ids(ext, len(c['context_before']), len(c['reference']), k, central, alphabet)
            same_level = np.concatenate([[False], model[3][ids[1:]] == model[3][ids[:-1]]])  # base j vs j-1
            rows = np.nonzero((ev2 != exp).any(axis=1))[0] if ev2.shape == exp.shape else np.array([-1])
            expl = ev2.shape == exp.shape and all(
                (ev2[j, 0] == exp[j, 0] or same_level[j]) and (ev2[j, 1] == exp[j, 1] or (j + 1 < len(ids) and same_level[j + 1]))
                for j in rows)
            n_tie += int(expl)
            print('ALIGN MISMATCH', 'tie-between-equal-kmers' if expl else 'UNEXPLAINED', 'rows', rows[:6].tolist(), 'it', it - 1, 'case', ci, 'k', k, 'central', central, 'alphabet', alphabet, 'mel', mel, 'bw', bw, 'tr', tr,
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
print('fuzz: %d batches, %d reads, %d mismatches (%d of them boundaries between bases with equal k-mer levels)' % (n_batches, n_reads, n_bad, n_tie))
