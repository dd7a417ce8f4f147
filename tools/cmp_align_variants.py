"""Compare two implementations of refine_alignment (NADAVCA_ALIGN_KERNEL values; 0 = default) on a
large seeded batch: events and statuses must be identical.
usage: cmp_align_variants.py N [varA varB [bandwidth mel seed R]]"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, '.')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
VARS = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ('0', '1')
bw = int(sys.argv[4]) if len(sys.argv) > 4 else 150
mel = int(sys.argv[5]) if len(sys.argv) > 5 else 2
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 31337
R = int(sys.argv[7]) if len(sys.argv) > 7 else 400
code = r'''
import sys, numpy as np, os
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic, _lib
model = synthetic.load_model_arrays(); mg = dtw.KmerModel(*model)
batch = synthetic.make_batch(%d, model, seed=%d, R=%d, R_spread=%d, bandwidth=%d)
fb = dtw.FlatBatch([(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in batch.cases])
for tr in (1, 0):
    ev, st = dtw.refine_alignment_flat(fb, %d, %d, mg, bool(tr))
    v = os.environ.get('NADAVCA_ALIGN_KERNEL', '0')
    print('variant', v, 'transitions', tr, 'ok', int((st == 0).sum()), 'no-path', int((st == 1).sum()),
          'redone by the exact kernel', _lib.default_context().last_batch_stats()['reads_redone_exact'])
    np.save('gpurun_out/cmp_ev_%%s_%%d.npy' %% (v, tr), ev)
    np.save('gpurun_out/cmp_st_%%s_%%d.npy' %% (v, tr), st)
''' % (n, seed, R, max(R // 10, 1), bw, bw, mel)
os.makedirs('gpurun_out', exist_ok=True)
for var in VARS:
    env = dict(os.environ); env.pop('NADAVCA_ALIGN_KERNEL', None)
    if var != '0': env['NADAVCA_ALIGN_KERNEL'] = var
    subprocess.run([sys.executable, '-c', code], env=env, check=True)
for tr in (1, 0):
    a = np.load('gpurun_out/cmp_ev_%s_%d.npy' % (VARS[0], tr)); b = np.load('gpurun_out/cmp_ev_%s_%d.npy' % (VARS[1], tr))
    sa = np.load('gpurun_out/cmp_st_%s_%d.npy' % (VARS[0], tr)); sb = np.load('gpurun_out/cmp_st_%s_%d.npy' % (VARS[1], tr))
    print('bw', bw, 'mel', mel, 'seed', seed, 'transitions', tr, 'reads', len(sa), 'events equal:', np.array_equal(a, b),
          'status equal:', np.array_equal(sa, sb), 'differing rows:', int((a != b).any(axis=1).sum()))
