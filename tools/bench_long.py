"""BASELINE config 5 shape (long reads, wide band): time refine_alignment with the default kernel
and with the exact kernel (NADAVCA_ALIGN_KERNEL=1) and check that the results are identical."""
import os, sys, subprocess, time, numpy as np
sys.path.insert(0, '.')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
code = r'''
import sys, time, numpy as np, os
sys.path.insert(0, '.')
from nadavca_amd import dtw, synthetic, _lib
model = synthetic.load_model_arrays(); mg = dtw.KmerModel(*model)
batch = synthetic.make_batch(%d, model, seed=5150, R=5000, R_spread=500, bandwidth=1000)
fb = dtw.FlatBatch([(c['signal'], c['reference'], c['context_before'], c['context_after'], c['approximate_alignment']) for c in batch.cases])
ev, st = dtw.refine_alignment_flat(fb, 1000, 2, mg, True)
t0 = time.perf_counter()
for _ in range(3):
    ev, st = dtw.refine_alignment_flat(fb, 1000, 2, mg, True)
dt = (time.perf_counter() - t0) / 3
s = _lib.default_context().last_batch_stats()
print('variant', os.environ.get('NADAVCA_ALIGN_KERNEL', 'default'), 'reads', len(st), 'ok', int((st == 0).sum()),
      'ms per batch %%.1f' %% (dt * 1e3), 'reads/s %%.1f' %% (len(st) / dt), 'samples/read %%.0f' %% (fb.signal.size / len(st)), s)
np.save(sys.argv[1], ev)
''' % n
os.makedirs('gpurun_out', exist_ok=True)
for var in ('0', '1'):
    env = dict(os.environ); env.pop('NADAVCA_ALIGN_KERNEL', None)
    if var == '1': env['NADAVCA_ALIGN_KERNEL'] = '1'
    subprocess.run([sys.executable, '-c', code, 'gpurun_out/long_ev_%s.npy' % var], env=env, check=True)
a = np.load('gpurun_out/long_ev_0.npy'); b = np.load('gpurun_out/long_ev_1.npy')
print('events identical:', np.array_equal(a, b))
