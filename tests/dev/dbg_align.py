import sys, numpy as np
sys.path.insert(0,'.')
from nadavca_amd import dtw, synthetic
from oracle.oracle import Oracle
o=Oracle('port')
model = synthetic.synth_model_arrays(11, k=5, central=2)
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
for mel in (0,1,2,3,4):
    cases=[]
    for i in range(24):
        rng = np.random.default_rng([77, mel, i])
        R = int(rng.integers(3, 140))
        cases.append(synthetic.make_dp_case(rng, model, R=R, bandwidth=int(rng.integers(8, 60)), dwell=(max(mel, 1), 9), jitter=6, anchor_density=float(rng.uniform(0.1, 0.9)), with_context=bool(i % 3), trim=min(3, R // 3)))
    reads=[(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in cases]
    for bw in (10,45):
        for tr in (False, True):
            got = dtw.refine_alignment_batch(reads, bw, mel, mg, tr)
            for ci,(c_,ev) in enumerate(zip(cases,got)):
                exp = o.refine_alignment(c_['signal'], c_['reference'], c_['context_before'], c_['context_after'], c_['approximate_alignment'], bw, mel, mo, tr)
                if ev.shape!=exp.shape or not np.array_equal(ev,exp):
                    nd = -1 if ev.shape!=exp.shape else int((ev!=exp).any(axis=1).sum())
                    first = None if ev.shape!=exp.shape else int(np.nonzero((ev!=exp).any(axis=1))[0][0])
                    print('MISMATCH mel',mel,'bw',bw,'tr',tr,'case',ci,'R',len(c_['reference']),'N',len(c_['signal']),'shapes',ev.shape,exp.shape,'ndiff',nd,'first',first, ev[first] if first is not None else '', exp[first] if first is not None else '')
print('done')
