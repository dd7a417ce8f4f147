"""Seeded random DP problems shared by tests/dev/fuzz_parity.py, tests/dev/fuzz_repro.py and the regression
tests that pin iterations the fuzzer once failed on."""
import numpy as np


def make_fuzz_batch(seed0, it):
    """-> dict(model, k, central, alphabet, mel, bw, cases, tr, w) for iteration `it` of seed `seed0`:
    random k-mer model (30 % of them with 0.3x or 3x the usual sigma), min event length, bandwidth, 1-9
    reads of 1-259 bases with random dwell, noise, anchor density and jitter."""
    from nadavca_amd import synthetic
    rng = np.random.default_rng([seed0, it])
    k = int(rng.integers(2, 7))
    central = int(rng.integers(0, k))
    alphabet = int(rng.choice([4, 4, 4, 3, 5]))
    model = synthetic.synth_model_arrays(int(rng.integers(1 << 30)), k=k, central=central, alphabet=alphabet)
    if rng.random() < 0.3:  # sharper or blunter levels
        model = model[:4] + (model[4] * float(rng.choice([0.3, 3.0])),)
    mel = int(rng.integers(0, 5))
    bw = int(rng.integers(4, 90))
    cases = []
    for _ in range(int(rng.integers(1, 10))):
        R = int(rng.integers(1, 260))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(4, 90)),
                                            dwell=(max(mel, 1), int(rng.integers(max(mel, 1) + 1, 14))),
                                            noise=float(rng.choice([0.1, 0.35, 1.0])), jitter=int(rng.integers(0, 25)),
                                            anchor_density=float(rng.uniform(0.05, 1.0)),
                                            with_context=bool(rng.integers(2)), trim=min(3, R // 3)))
    tr, w = bool(rng.integers(2)), bool(rng.integers(2))
    return dict(model=model, k=k, central=central, alphabet=alphabet, mel=mel, bw=bw, cases=cases, tr=tr, w=w)


def make_team_batch(seed0, it):
    """-> dict(model, k, mel, bw, cases, tr) aimed at the wide-band path of refine_alignment (teams of waves,
    kernels_align3.hip): bandwidths 100-700 on reads of 20-700 bases (a few tiny ones and a few narrow bands
    mixed in), min event length 0-4, packaged or random k-mer model."""
    from nadavca_amd import synthetic
    rng = np.random.default_rng([seed0, it])
    if rng.random() < 0.5:
        model = synthetic.load_model_arrays()
        k = model[0]
    else:
        k = int(rng.integers(3, 7))
        model = synthetic.synth_model_arrays(int(rng.integers(1 << 30)), k=k, central=int(rng.integers(0, k)), alphabet=4)
    mel = int(rng.integers(0, 5))
    bw = int(rng.integers(100, 700))
    tr = bool(rng.integers(2))
    cases = []
    for _ in range(int(rng.integers(1, 6))):
        R = int(rng.integers(20, 700)) if rng.random() < 0.8 else int(rng.integers(1, 20))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(bw if rng.random() < 0.8 else rng.integers(4, 60)),
                                            dwell=(max(mel, 1), int(rng.integers(max(mel, 1) + 1, 14))),
                                            noise=float(rng.choice([0.2, 0.35, 0.8])), jitter=int(rng.integers(0, 25)),
                                            anchor_density=float(rng.uniform(0.02, 1.0)),
                                            with_context=bool(rng.integers(2)), trim=min(3, R // 3)))
    return dict(model=model, k=k, mel=mel, bw=bw, cases=cases, tr=tr)


def reads_of(cases):
    return [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'])
            for c in cases]


def classify_difference(ev, exp, case, fb_model, k, central, alphabet, bw, mel, tr, referee=None):
    """Why do the engine's rows ``ev`` differ from the double-precision reference's ``exp`` on this read?
    -> one of 'flat-plateau' (every differing boundary lies between two bases with the SAME k-mer level: the
    posterior is exactly flat there and any rounding decides), 'reference-rounding' (the same algorithm in 80-bit
    long double — oracle/liboracle_ld.so — sides with the engine on every differing row), 'precision-decided'
    (the reference changes its OWN answer on every differing row when computed in long double: an ill-conditioned
    arg-max), or 'UNEXPLAINED'.  The classification tests/dev/fuzz_parity.py prints, shared with the -m gpu
    tests so that a regression cannot hide behind the near-tie flag."""
    from nadavca_amd import synthetic
    from oracle.oracle import LongDoubleReferee
    ev2 = np.asarray(ev).reshape(-1, 2)
    exp = np.asarray(exp).reshape(-1, 2)
    if ev2.shape != exp.shape:
        return 'UNEXPLAINED'
    c = case
    ext = np.concatenate([c['context_before'], c['reference'], c['context_after']]).astype(np.int64)
    ids = synthetic.kmer_ids(ext, len(c['context_before']), len(c['reference']), k, central, alphabet)
    mean = fb_model[3]
    same_level = np.concatenate([[False], mean[ids[1:]] == mean[ids[:-1]]])  # base j vs j-1
    rows = np.nonzero((ev2 != exp).any(axis=1))[0]

    def flat(j):  # every boundary of row j that differs lies between equal levels
        return (ev2[j, 0] == exp[j, 0] or same_level[j]) and \
               (ev2[j, 1] == exp[j, 1] or (j + 1 < len(ids) and same_level[j + 1]))

    if all(flat(j) for j in rows):
        return 'flat-plateau'
    ld = referee if referee is not None else LongDoubleReferee(*fb_model)
    hp = ld.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                             c['approximate_alignment'], bw, mel, tr)
    if hp.shape != exp.shape:
        return 'UNEXPLAINED'
    if all(np.array_equal(ev2[j], hp[j]) or flat(j) for j in rows):
        return 'reference-rounding'
    if all(not np.array_equal(exp[j], hp[j]) or flat(j) for j in rows):
        return 'precision-decided'
    return 'UNEXPLAINED'
