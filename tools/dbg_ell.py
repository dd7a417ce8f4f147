"""Compare the default (plain-double) hypothesis phase of estimate_log_likelihoods with the exact
variant (NADAVCA_ELL_KERNEL=1) on seeded cases; prints where they differ."""
import os, sys, subprocess, numpy as np, pickle
sys.path.insert(0, '.')
code = r'''
import sys, numpy as np, pickle, os
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic, _lib
res = []
for mel in (0, 1, 2, 3, 4):
    model = synthetic.synth_model_arrays(21, k=5, central=2)
    mg = dtw.KmerModel(*model)
    cases = []
    for i in range(12):
        rng = np.random.default_rng([88, mel, i])
        R = int(rng.integers(3, 90))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 50)),
                                            dwell=(max(mel, 1), 9), jitter=6,
                                            anchor_density=float(rng.uniform(0.1, 0.9)),
                                            with_context=bool(i % 3), trim=min(3, R // 3)))
    reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in cases]
    for bw in (12, 40):
        for w in (False, True):
            got = dtw.estimate_log_likelihoods_batch(reads, bw, mel, mg, w)
            st = _lib.default_context().last_batch_stats()
            res.append(((mel, bw, w), [np.asarray(g) for g in got], st['reads_redone_exact']))
pickle.dump(res, open(sys.argv[1], 'wb'))
'''
os.makedirs('gpurun_out', exist_ok=True)
outs = {}
for var in ('0', '1'):
    env = dict(os.environ); env.pop('NADAVCA_ELL_KERNEL', None)
    if var == '1': env['NADAVCA_ELL_KERNEL'] = '1'
    f = 'gpurun_out/dbg_ell_%s.pkl' % var
    subprocess.run([sys.executable, '-c', code, f], env=env, check=True)
    outs[var] = pickle.load(open(f, 'rb'))
for (key, a, ra), (_, b, rb) in zip(outs['0'], outs['1']):
    worst = 0.0; where = None
    for ri, (x, y) in enumerate(zip(a, b)):
        fin = np.isfinite(y)
        if not np.array_equal(np.isneginf(x), np.isneginf(y)) or np.any(np.isnan(x)):
            print(key, 'read', ri, 'inf/nan pattern differs')
        d = np.abs(x[fin] - y[fin]) / np.maximum(1.0, np.abs(y[fin]))
        if d.size and d.max() > worst:
            worst = d.max(); pos = np.argwhere(np.abs(x - y) / np.maximum(1.0, np.abs(y)) == worst)
            where = (ri, x.shape, pos[:4].tolist())
    print(key, 'retries', ra, 'worst rel diff %.3g' % worst, where)
