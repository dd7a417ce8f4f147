import os, sys, numpy as np
sys.path.insert(0,'.')
os.environ['NADAVCA_ALIGN3_NORETRY']='1'
from nadavca_amd import dtw, synthetic, _lib
from oracle.oracle import Oracle
import ctypes as C
o=Oracle('port')
model = synthetic.load_model_arrays()
mg = dtw.KmerModel(*model); mo = o.KmerModel(*model)
batch = synthetic.make_batch(64, model, seed=5, R=400, R_spread=40, bandwidth=150)
fb = dtw.FlatBatch([(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in batch.cases])
lib=_lib.load()
for tr in (True, False):
    events = np.zeros((int(fb.ref_off[-1]), 2), dtype=np.int32); status = np.zeros(fb.n, dtype=np.int32)
    rc = lib.nvk_refine_alignment_batch(mg.handle, fb.n, *fb.pointers(), 150, 2, int(tr), C.c_void_p(events.ctypes.data), C.c_void_p(status.ctypes.data))
    print('tr',tr,'rc',rc,'status counts', np.bincount(status+2))
    bad=0
    for j,c_ in enumerate(batch.cases[:16]):
        exp = o.refine_alignment(c_['signal'], c_['reference'], c_['context_before'], c_['context_after'], c_['approximate_alignment'], 150, 2, mo, tr)
        ev = events[fb.ref_off[j]:fb.ref_off[j+1]]
        if status[j]==0 and not np.array_equal(ev,exp):
            bad+=1; d=np.nonzero((ev!=exp).any(axis=1))[0]; print(' read',j,'ndiff',len(d),'first',d[0],ev[d[0]],exp[d[0]])
    print(' mismatching ok-reads among first 16:',bad)
