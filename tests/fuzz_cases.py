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
