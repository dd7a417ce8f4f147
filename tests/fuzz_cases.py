"""Seeded random DP problems shared by tests/dev/fuzz_parity.py, tests/dev/fuzz_repro.py and the regression
tests that pin iterations the fuzzer once failed on."""
import os

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


_NEAR_LIB = {}


def _near_tie_lib():
    """An instrumented long-double copy of the CPU restatement (oracle/nadavca_oracle.c), built once per process
    in a temporary directory: every `a > b` of the path search (node.cpp:52,72,82 restated) whose two scores
    are equal or differ — in 80-bit arithmetic — by less than the engine's tolerance zone (NVK_TIE_ULPS = 64 ulps
    of the reference's log value: 64 * 2^-52 relative) notes its row.  Test infrastructure."""
    if 'lib' in _NEAR_LIB:
        return _NEAR_LIB['lib']
    import ctypes as C
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = open(os.path.join(root, 'oracle', 'nadavca_oracle.c')).read()
    helper = '''#include <string.h>
int g_near_n; int g_near_row[4096];
static void near_note(double a, double b, int r) {   /* (`double` is long double in this build) */
  if (!(a > -1e300) || !(b > -1e300)) return;
  double d = a - b; if (d < 0) d = -d;   /* d == 0: a tie even in 80-bit arithmetic */
  double m = b < 0 ? -b : b;
  if (d <= m * 64.0 * 2.220446049250313e-16 && g_near_n < 4096) g_near_row[g_near_n++] = r;
}
int orc_near_count(void) { return g_near_n; }
int *orc_near_rows(void) { return g_near_row; }
void orc_near_reset(void) { g_near_n = 0; }
'''
    a = '''        double pv = dp[r - 1][i - bs[r - 1]];
        if (pv > best) {'''
    b = '''          double pv = dp[r - 1][from - bs[r - 1]];
          if (pv > best) {'''
    c = '''      for (int i = bs[r]; i <= be[r]; i++)
        if (dp[r][i - bs[r]] > best) {
          best = dp[r][i - bs[r]];
          best_idx = i;
        }'''
    assert a in s and b in s and c in s and '#include <string.h>' in s, 'oracle/nadavca_oracle.c changed: adapt the patch'
    s = s.replace(a, a.replace('        if (pv > best) {', '        near_note(pv, best, r);\n        if (pv > best) {'))
    s = s.replace(b, b.replace('          if (pv > best) {', '          near_note(pv, best, r);\n          if (pv > best) {'))
    s = s.replace(c, '''      for (int i = bs[r]; i <= be[r]; i++) {
        near_note(dp[r][i - bs[r]], best, r + 1);
        if (dp[r][i - bs[r]] > best) {
          best = dp[r][i - bs[r]];
          best_idx = i;
        }
      }''')
    s = s.replace('#include <string.h>', helper, 1)
    pre = ('#include <math.h>\n#include <stdint.h>\n#include <stdlib.h>\n#include <string.h>\n#include <stdio.h>\n'
           '#define double long double\n#define exp expl\n#define log logl\n#define sqrt sqrtl\n')
    tmp = tempfile.mkdtemp(prefix='orc_near_')
    src, lib = os.path.join(tmp, 'orc_near.c'), os.path.join(tmp, 'liborc_near.so')
    open(src, 'w').write(pre + s)
    subprocess.run(['gcc', '-O2', '-fPIC', '-std=gnu11', '-ffp-contract=off', '-Wno-unused', '-shared', '-o', lib, src,
                    '-lm'], check=True)
    _NEAR_LIB['lib'] = C.CDLL(lib)
    return _NEAR_LIB['lib']


def near_tie_bases(case, fb_model, bw, mel, tr):
    """Bases (event rows) at which the long-double reference compares two path scores inside the engine's
    tolerance zone on this read."""
    import ctypes as C
    from oracle.oracle import LongDoubleReferee
    lib = _near_tie_lib()
    ld = LongDoubleReferee.__new__(LongDoubleReferee)
    ld.lib = lib
    ld._pld = C.POINTER(C.c_longdouble)
    lib.orc_model_create.restype = C.c_void_p
    lib.orc_model_create.argtypes = [C.c_int, C.c_int, C.c_int, ld._pld, ld._pld, C.c_int64]
    lib.orc_refine_alignment.restype = C.c_int
    p_i32 = C.POINTER(C.c_int32)
    lib.orc_refine_alignment.argtypes = [C.c_void_p, ld._pld, C.c_int64, p_i32, C.c_int64, p_i32, C.c_int64, p_i32,
                                         C.c_int64, p_i32, C.c_int64, C.c_int, C.c_int, C.c_int, p_i32]
    k, central, alphabet, mean, sigma = fb_model
    mean_l = np.ascontiguousarray(mean, dtype=np.longdouble)
    sigma_l = np.ascontiguousarray(sigma, dtype=np.longdouble)
    ld.handle = lib.orc_model_create(int(k), int(central), int(alphabet), mean_l.ctypes.data_as(ld._pld),
                                     sigma_l.ctypes.data_as(ld._pld), mean_l.size)
    lib.orc_near_reset()
    c = case
    ld.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                        c['approximate_alignment'], bw, mel, tr)
    lib.orc_near_rows.restype = C.POINTER(C.c_int)
    n = lib.orc_near_count()
    rows = np.array([lib.orc_near_rows()[i] for i in range(n)], dtype=np.int64)
    return np.unique(rows // 2 if tr else rows)


def classify_difference(ev, exp, case, fb_model, k, central, alphabet, bw, mel, tr, referee=None):
    """Why do the engine's rows ``ev`` differ from the double-precision reference's ``exp`` on this read?
    -> one of 'flat-plateau' (every differing boundary lies between two bases with the SAME k-mer level: the
    posterior is exactly flat there and any rounding decides), 'reference-rounding' (the same algorithm in 80-bit
    long double — oracle/liboracle_ld.so — sides with the engine on every differing row), 'precision-decided'
    (the reference changes its OWN answer on every differing row when computed in long double: an ill-conditioned
    arg-max), 'referee-tie' (reference and long double agree, and at the differing base the long-double reference
    itself compares two path scores that are equal or closer than the engine's tolerance zone: a tie for the
    reference at the magnitude of ITS scores, a resolvable difference for the engine — the read carries
    NVK_TIE_ULP or NVK_TIE_NEAR), or 'UNEXPLAINED'.  The classification tests/dev/fuzz_parity.py prints, shared with the -m gpu
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
    # The reference's doubles and the long doubles agree with each other and not with the engine: accepted only
    # where the long-double reference ITSELF finds the two path scores it compares at that base (or a neighbouring
    # one) equal, or closer than the engine's tolerance zone.  The reference adds UNNORMALISED log posteriors
    # (node.cpp:39-50): after r rows its scores are ~ r * |ln L(read)|, 5e6 on a sharp model, where even 80-bit
    # arithmetic resolves 1e-12 at best — it then keeps the first maximum — while the engine's scores are products
    # of normalised posteriors, 1e3-1e4 times smaller in the exponent, and tell such cells apart.
    near = near_tie_bases(c, fb_model, bw, mel, tr)
    if all(flat(j) or np.array_equal(ev2[j], hp[j]) or not np.array_equal(exp[j], hp[j]) or
           bool(np.any(np.abs(near - j) <= 1)) for j in rows):
        return 'referee-tie'
    return 'UNEXPLAINED'
