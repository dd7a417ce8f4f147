"""TEST INFRASTRUCTURE ONLY — regenerate tests/golden/dp_*.npz.

Runs the *reference's own* DP (oracle/_ref/libnadavca_ref.so, built by
``make -C oracle ref`` from /root/reference/nadavca/dtw/*.cpp) on deterministic
synthetic inputs and stores inputs + outputs as small .npz fixtures.  Only data is
committed; the reference never travels.  Usage (build container only):

    python3 oracle/make_golden.py

Groups:
  dp_tiny.npz    G1  R 8-30, bw 6-15, k=3 toy model, min_event_length 0-3, ctx on/off
  dp_config.npz  G2  R~400, bw 150, packaged 6-mer model, default config sizes, incl.
                     sparse anchors and bands clipped at 0 / N
  dp_nopath.npz  G3  a case with no valid path (refine -> [], ELL -> all -inf)
Each case stores the four ops: refine (transitions T/F), ELL (wobbling T/F), and
get_expected_signal (G6).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from nadavca_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def toy_model(seed):
    rng = np.random.default_rng(seed)
    n = 4 ** 3
    return 3, 1, 4, rng.normal(0, 1.2, n), 0.3 + 0.2 * rng.random(n)


def run_case(o, model_arrays, case, bw, mel):
    k, central, alpha, mean, sigma = model_arrays
    m = o.KmerModel(k, central, alpha, mean, sigma)
    a = (case['signal'], case['reference'], case['context_before'], case['context_after'],
         case['approximate_alignment'], bw, mel, m)
    out = dict(case)
    out.pop('true_starts', None)
    out['bandwidth'] = np.int64(bw)
    out['min_event_length'] = np.int64(mel)
    for tr in (0, 1):
        r = o.refine_alignment(*a, bool(tr))
        out['refine_t%d' % tr] = r.astype(np.int32)
    for w in (0, 1):
        out['ell_w%d' % w] = o.estimate_log_likelihoods(*a, bool(w))
    out['expected_signal'] = m.get_expected_signal(case['reference'], case['context_before'],
                                                   case['context_after'])
    return out


def save(name, model_arrays, cases, note):
    k, central, alpha, mean, sigma = model_arrays
    blob = {'model_k': np.int64(k), 'model_central': np.int64(central), 'model_alphabet': np.int64(alpha),
            'model_mean': mean, 'model_sigma': sigma, 'n_cases': np.int64(len(cases)),
            'note': np.array(json.dumps(note))}
    for i, c in enumerate(cases):
        for key, v in c.items():
            blob['c%d_%s' % (i, key)] = np.asarray(v)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **blob)
    print(name, len(cases), 'cases', os.path.getsize(path) // 1024, 'KiB')


def main():
    o = Oracle('reference')
    os.makedirs(OUT, exist_ok=True)

    # G1 tiny
    tm = toy_model(7)
    cases = []
    for i in range(24):
        rng = np.random.default_rng([101, i])
        R = int(rng.integers(8, 31))
        bw = int(rng.integers(6, 16))
        mel = [2, 2, 1, 3, 0, 2][i % 6]
        c = synthetic.make_dp_case(rng, tm, R=R, bandwidth=bw, dwell=(2, 6), noise=0.3,
                                   anchor_density=0.6, jitter=3, with_context=(i % 2 == 0), trim=1)
        cases.append(run_case(o, tm, c, bw, mel))
    save('dp_tiny.npz', tm, cases, {'generator': 'oracle/make_golden.py G1', 'seed': 101})

    # G2 config-sized (6-mer packaged model, default config)
    dm = synthetic.load_model_arrays()
    cases = []
    for i in range(10):
        rng = np.random.default_rng([202, i])
        kw = dict(R=int(400 + rng.integers(-40, 41)), bandwidth=150)
        if i in (3, 7):
            kw['anchor_density'] = 0.5
        if i in (4, 8):
            kw['pad_bases'] = 6          # slice starts at 0 / ends at len: bands clipped at 0 and N
        if i == 9:
            kw['with_context'] = False
        c = synthetic.make_dp_case(rng, dm, **kw)
        cases.append(run_case(o, dm, c, 150, 2))
    save('dp_config.npz', dm, cases, {'generator': 'oracle/make_golden.py G2', 'seed': 202})

    # G3 no path: R=20, N=30 < 2*R with mel=2, anchors [[0,0],[29,19]], bw 5
    rng = np.random.default_rng(303)
    c = dict(signal=rng.normal(0, 1, 30), reference=rng.integers(0, 4, 20).astype(np.int32),
             context_before=np.zeros(0, np.int32), context_after=np.zeros(0, np.int32),
             approximate_alignment=np.array([[0, 0], [29, 19]], dtype=np.int32))
    cases = [run_case(o, tm, c, 5, 2)]
    assert cases[0]['refine_t0'].size == 0 and cases[0]['refine_t1'].size == 0
    assert np.all(np.isneginf(cases[0]['ell_w1']))
    save('dp_nopath.npz', tm, cases, {'generator': 'oracle/make_golden.py G3', 'seed': 303})


if __name__ == '__main__':
    main()
