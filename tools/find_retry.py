"""Development aid: which reads of a workload does the fast align kernel hand to the exact one?
usage: python tools/find_retry.py LIB.so workload n_reads [transitions 0/1]  (needs a library built with
APIDEBUG=1 and NADAVCA_ALIGN3_NORETRY=1 in the environment: flagged reads then keep their internal status 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nadavca_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from nadavca_amd import dtw, synthetic
wl = dict(synthetic.WORKLOADS[sys.argv[2]]); wl.pop('n_reads')
n = int(sys.argv[3])
trs = [bool(int(sys.argv[4]))] if len(sys.argv) > 4 else [True, False]
model = synthetic.load_model_arrays()
km = dtw.KmerModel(*model)
batch = synthetic.make_batch(n, model, seed=1000, **wl)
flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off, batch.context_before,
                                 batch.cb_off, batch.context_after, batch.ca_off, batch.anchors, batch.anc_off)
for tr in trs:
    ev, st = dtw.refine_alignment_flat(flat, wl['bandwidth'], 2, km, tr, on_error='status')
    bad = np.nonzero(st != 0)[0]
    print(os.path.basename(sys.argv[1]), 'transitions', tr, ': %d flagged of %d, first' % (bad.size, n), bad[:6].tolist(),
          'samples', [int(batch.sig_off[j + 1] - batch.sig_off[j]) for j in bad[:6]], flush=True)
