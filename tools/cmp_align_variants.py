"""Compare the default align path (scaled doubles + exact fallback) with the exact kernel alone on
a large seeded batch: events and statuses must be identical."""
import os, sys, subprocess, numpy as np
sys.path.insert(0, '.')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
code = r'''
import sys, numpy as np, ctypes as C, os
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic, _lib
model = synthetic.load_model_arrays(); mg = dtw.KmerModel(*model)
batch = synthetic.make_batch(%d, model, seed=31337, R=400, R_spread=40, bandwidth=150)
fb = dtw.FlatBatch([(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in batch.cases])
for tr in (1, 0):
    ev, st = dtw.refine_alignment_flat(fb, 150, 2, mg, bool(tr))
    np.save('gpurun_out/cmp_ev_%%s_%%d.npy' %% (os.environ.get('NADAVCA_ALIGN_KERNEL', '0'), tr), ev)
    np.save('gpurun_out/cmp_st_%%s_%%d.npy' %% (os.environ.get('NADAVCA_ALIGN_KERNEL', '0'), tr), st)
print('done')
''' % n
os.makedirs('gpurun_out', exist_ok=True)
VARS = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ('0', '1')
for var in VARS:
    env = dict(os.environ); env.pop('NADAVCA_ALIGN_KERNEL', None)
    if var != '0': env['NADAVCA_ALIGN_KERNEL'] = var
    subprocess.run([sys.executable, '-c', code], env=env, check=True)
for tr in (1, 0):
    a = np.load('gpurun_out/cmp_ev_%s_%d.npy' % (VARS[0], tr)); b = np.load('gpurun_out/cmp_ev_%s_%d.npy' % (VARS[1], tr))
    sa = np.load('gpurun_out/cmp_st_%s_%d.npy' % (VARS[0], tr)); sb = np.load('gpurun_out/cmp_st_%s_%d.npy' % (VARS[1], tr))
    print('transitions', tr, 'reads', len(sa), 'events equal:', np.array_equal(a, b), 'status equal:', np.array_equal(sa, sb),
          'differing rows:', int((a != b).any(axis=1).sum()))
