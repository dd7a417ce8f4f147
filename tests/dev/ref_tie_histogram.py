"""How close do path scores come in the REFERENCE's own arithmetic?  Builds an instrumented copy of the CPU
restatement (oracle/nadavca_oracle.c, bit-identical to the compiled reference) in a temporary directory that
notes |a - b| for every `a > b` of the path search (node.cpp:52,72,82 restated), and prints the histogram on
config-2-shaped reads as simulated (continuous samples) and quantised to ADC steps, with and without transition
rows.  CPU only; test infrastructure (it compiles and loads oracle code).
usage: python tests/dev/ref_tie_histogram.py [n_reads]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HELPER = '''#include <string.h>
long long g_tie[8];
static void tie_note(double a, double b) {
  if (!(a > -1e300) || !(b > -1e300)) return;
  double d = a - b; if (d < 0) d = -d;
  g_tie[0]++;
  if (d == 0.0) g_tie[1]++;
  else if (d < 1e-13) g_tie[2]++;
  else if (d < 1e-12) g_tie[3]++;
  else if (d < 1e-11) g_tie[4]++;
  else if (d < 1e-10) g_tie[5]++;
  else if (d < 1e-9) g_tie[6]++;
  else if (d < 5.96e-8) g_tie[7]++;
}
long long *orc_tie_counts(void) { return g_tie; }
'''


def build(tmp):
    s = open(os.path.join(ROOT, 'oracle', 'nadavca_oracle.c')).read()
    a = '''          double pv = dp[r - 1][from - bs[r - 1]];
          if (pv > best) {'''
    b = '''      for (int i = bs[r]; i <= be[r]; i++)
        if (dp[r][i - bs[r]] > best) {
          best = dp[r][i - bs[r]];
          best_idx = i;
        }'''
    assert a in s and b in s and '#include <string.h>' in s, 'oracle/nadavca_oracle.c changed: adapt the patch'
    s = s.replace(a, a.replace('          if (pv > best) {', '          tie_note(pv, best);\n          if (pv > best) {'))
    s = s.replace(b, '''      for (int i = bs[r]; i <= be[r]; i++) {
        tie_note(dp[r][i - bs[r]], best);
        if (dp[r][i - bs[r]] > best) {
          best = dp[r][i - bs[r]];
          best_idx = i;
        }
      }''')
    s = s.replace('#include <string.h>', HELPER, 1)
    src, lib = os.path.join(tmp, 'orc_tie.c'), os.path.join(tmp, 'liborc_tie.so')
    open(src, 'w').write(s)
    subprocess.run(['gcc', '-O2', '-fPIC', '-std=gnu11', '-ffp-contract=off', '-shared', '-o', lib, src, '-lm'], check=True)
    return lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import oracle.oracle as O
    from nadavca_amd import synthetic
    with tempfile.TemporaryDirectory() as tmp:
        O.PORT_LIB = build(tmp)
        o = O.Oracle('port')
        o.lib.orc_tie_counts.restype = C.POINTER(C.c_longlong)
        g = o.lib.orc_tie_counts()
        model = synthetic.load_model_arrays()
        mo = o.KmerModel(*model)
        batch = synthetic.make_batch(n, model, seed=1000, R=400, R_spread=40, bandwidth=150)
        edges = ['== 0', '< 1e-13', '< 1e-12', '< 1e-11', '< 1e-10', '< 1e-9', '< 2^-24']
        for quant in (False, True):
            for tr in (True, False):
                per = []
                for c in batch.cases:
                    sig = np.round(c['signal'] * 12.0) / 12.0 if quant else c['signal']
                    before = [g[i] for i in range(8)]
                    o.refine_alignment(sig, c['reference'], c['context_before'], c['context_after'],
                                       c['approximate_alignment'], 150, 2, mo, tr)
                    per.append([g[i] - before[i] for i in range(8)])
                per = np.array(per)
                print('%s signals, transitions=%s: %d comparisons in %d reads; |a - b| %s'
                      % ('quantised' if quant else 'continuous', tr, per[:, 0].sum(), n,
                         ', '.join('%s: %d' % (e, v) for e, v in zip(edges, per[:, 1:].sum(axis=0)))))
                print('    reads with an exact tie: %d, with 0 < |a-b| < 1e-11: %d, with 1e-11 <= |a-b| < 2^-24: %d'
                      % ((per[:, 1] > 0).sum(), (per[:, 2:5].sum(axis=1) > 0).sum(), (per[:, 5:].sum(axis=1) > 0).sum()))


if __name__ == '__main__':
    main()
