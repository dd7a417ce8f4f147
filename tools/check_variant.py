"""Development aid: `python tools/check_variant.py LIB.so [n_reads]` — the default refine_alignment kernel of
another build of the library on config-2-shaped reads: how many reads it handed to the exact kernel, whether
its events equal the exact kernel's (NADAVCA_ALIGN_KERNEL=1, same library), and its kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nadavca_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
from nadavca_amd import dtw, synthetic
model = synthetic.load_model_arrays()
ctx = _lib.default_context()
km = dtw.KmerModel(*model, context=ctx)
batch = synthetic.make_batch(n, model, seed=1000, R=400, R_spread=40, bandwidth=150)
flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off, batch.context_before,
                                 batch.cb_off, batch.context_after, batch.ca_off, batch.anchors, batch.anc_off)
for tr in (True, False):
    os.environ.pop('NADAVCA_ALIGN_KERNEL', None)
    ev, st = dtw.refine_alignment_flat(flat, 150, 2, km, tr)
    stats = ctx.last_batch_stats()
    ctx.timing_enable(True); ctx.timing_reset()
    dtw.refine_alignment_flat(flat, 150, 2, km, tr)
    ms = ctx.timing_read()['align'][0]
    ctx.timing_enable(False)
    os.environ['NADAVCA_ALIGN_KERNEL'] = '1'
    ev1, st1 = dtw.refine_alignment_flat(flat, 150, 2, km, tr)
    os.environ.pop('NADAVCA_ALIGN_KERNEL', None)
    print('%s transitions=%s: %d reads, redone by the exact kernel %d, tie flags %d, rows differing from the exact kernel %d, '
          'align kernels %.2f ms' % (os.path.basename(sys.argv[1]), tr, n, stats['reads_redone_exact'],
                                     stats['reads_tie_ambiguous'], int((ev != ev1).any(axis=1).sum()), ms), flush=True)
