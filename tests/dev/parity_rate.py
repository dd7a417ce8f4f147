"""How often does refine_alignment differ from the double-precision reference on config-2-shaped reads
with the packaged 6-mer model, and are the differences the kind DESIGN.md 2.1 describes?
usage: parity_rate.py N [seed]"""
import sys, numpy as np
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle, LongDoubleReferee
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 77
model = synthetic.load_model_arrays()
k, central, alphabet = model[0], model[1], model[2]
mg = dtw.KmerModel(*model)
o = Oracle('port'); mo = o.KmerModel(*model)
batch = synthetic.make_batch(n, model, seed=seed, R=400, R_spread=40, bandwidth=150)
reads = [(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in batch.cases]
for tr in (True, False):
    got = dtw.refine_alignment_batch(reads, 150, 2, mg, tr)
    with ThreadPoolExecutor(16) as ex:  # the C oracle releases the GIL inside ctypes calls
        exp = list(ex.map(lambda c: o.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                                        c['approximate_alignment'], 150, 2, mo, tr), batch.cases))
    ld = None
    n_diff = n_flat = n_hp = n_rows = 0
    for c, ev, e in zip(batch.cases, got, exp):
        ev = np.asarray(ev).reshape(-1, 2); e = np.asarray(e).reshape(-1, 2)
        if ev.shape == e.shape and np.array_equal(ev, e):
            continue
        n_diff += 1
        rows = np.nonzero((ev != e).any(axis=1))[0]
        n_rows += len(rows)
        ext = np.concatenate([c['context_before'], c['reference'], c['context_after']]).astype(np.int64)
        ids = synthetic.kmer_ids(ext, len(c['context_before']), len(c['reference']), k, central, alphabet)
        same = np.concatenate([[False], model[3][ids[1:]] == model[3][ids[:-1]]])
        flat = lambda j: (ev[j, 0] == e[j, 0] or same[j]) and (ev[j, 1] == e[j, 1] or (j + 1 < len(ids) and same[j + 1]))
        if all(flat(j) for j in rows):
            n_flat += 1
            continue
        if ld is None:
            ld = LongDoubleReferee(*model)
        hp = ld.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment'], 150, 2, tr)
        if all(np.array_equal(ev[j], hp[j]) or flat(j) for j in rows):
            n_hp += 1
    print('transitions', tr, ': %d reads, %d differ from the double reference in %d rows (of %d); %d only on flat plateaus between equal k-mers, '
          '%d where the long-double reference sides with the engine, %d unexplained'
          % (n, n_diff, n_rows, sum(len(c['reference']) for c in batch.cases), n_flat, n_hp, n_diff - n_flat - n_hp))
