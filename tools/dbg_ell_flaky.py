"""Repeat the mel=4 random ELL parity case against the CPU oracle and report any deviation in detail."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle
o = Oracle('port')
model = synthetic.synth_model_arrays(21, k=5, central=2)
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
mels = [int(x) for x in sys.argv[2:]] or [4]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for mel in mels:
    cases = []
    for i in range(12):
        rng = np.random.default_rng([88, mel, i])
        R = int(rng.integers(3, 90))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 50)),
                                            dwell=(max(mel, 1), 9), jitter=6,
                                            anchor_density=float(rng.uniform(0.1, 0.9)),
                                            with_context=bool(i % 3), trim=min(3, R // 3)))
    reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in cases]
    exp = {}
    for bw in (12, 40):
        for w in (False, True):
            exp[(bw, w)] = [np.asarray(o.estimate_log_likelihoods(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                                                  c['approximate_alignment'], bw, mel, mo, w)) for c in cases]
    bad = 0
    for rep in range(reps):
        for (bw, w), ex in exp.items():
            got = dtw.estimate_log_likelihoods_batch(reads, bw, mel, mg, w)
            for ri, (g, e) in enumerate(zip(got, ex)):
                fin = np.isfinite(e)
                same_inf = np.array_equal(np.isneginf(g), np.isneginf(e))
                d = np.abs(g[fin] - e[fin]) / np.maximum(1.0, np.abs(e[fin])) if fin.any() else np.zeros(1)
                if not same_inf or np.any(np.isnan(g)) or d.max() > 1e-9:
                    bad += 1
                    w_ = np.argwhere(~np.isclose(g, e, rtol=1e-9, atol=1e-9, equal_nan=False))
                    print('DEVIATION rep', rep, 'mel', mel, 'bw', bw, 'w', w, 'read', ri, 'R', len(cases[ri]['reference']),
                          'N', len(cases[ri]['signal']), 'same_inf', same_inf, 'max', float(d.max()), 'where', w_[:6].tolist(),
                          'got', g[tuple(w_[0])] if len(w_) else None, 'exp', e[tuple(w_[0])] if len(w_) else None)
    print('mel', mel, 'reps', reps, 'deviations', bad)
