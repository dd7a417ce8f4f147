"""Randomised parity run aimed at the wide-band path of refine_alignment (teams of waves, kernels_align3.hip):
bandwidths 100-700 on reads of 20-700 bases, so that most reads need a skew above the one-wave launch's cap;
min event length 0-4, transitions on/off, random and packaged k-mer models, mixed narrow reads in the batch.
Checks the parity contract: every read that differs from the reference carries the near-tie bit.
usage: fuzz_team.py SECONDS [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle, have_reference
from fuzz_cases import make_team_batch, reads_of

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = Oracle('reference' if have_reference() else 'port')
t_end = time.time() + budget
t_tick = time.time() + 60
it = n_reads = n_diff = n_unflagged = n_flag = n_nopath = 0
while time.time() < t_end:
    fb = make_team_batch(seed0, it); it += 1
    model, k, mel, bw, cases, tr = (fb[x] for x in ('model', 'k', 'mel', 'bw', 'cases', 'tr'))
    mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
    got = dtw.refine_alignment_batch(reads_of(cases), bw, mel, mg, tr)
    ties = mg.context.last_tie_flags(len(cases))
    n_reads += len(cases)
    n_flag += int((ties != 0).sum())
    for ci, (c, ev) in enumerate(zip(cases, got)):
        exp = np.asarray(o.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                            c['approximate_alignment'], bw, mel, mo, tr)).reshape(-1, 2)
        ev = np.asarray(ev).reshape(-1, 2)
        n_nopath += int(exp.size == 0)
        if ev.shape != exp.shape or not np.array_equal(ev, exp):
            n_diff += 1
            if not (ties[ci] & 6):   # NVK_TIE_NEAR | NVK_TIE_ULP: the contract's only escape
                n_unflagged += 1
                print('UNFLAGGED MISMATCH it', it - 1, 'case', ci, 'k', k, 'mel', mel, 'bw', bw, 'tr', tr, 'R', len(c['reference']),
                      'N', len(c['signal']), 'shapes', ev.shape, exp.shape, flush=True)
    if time.time() > t_tick:
        t_tick = time.time() + 60
        print('... %d batches, %d reads, %d differ, %d unflagged' % (it, n_reads, n_diff, n_unflagged), file=sys.stderr, flush=True)
print('fuzz_team: %d batches, %d reads (%d without a path), %d flagged tie-ambiguous, %d differ from the reference, '
      '%d of those NOT flagged' % (it, n_reads, n_nopath, n_flag, n_diff, n_unflagged))
sys.exit(1 if n_unflagged else 0)
