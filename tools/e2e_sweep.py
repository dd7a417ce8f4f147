"""Development aid (GPU box): T_e2e of the host-pointer refine_alignment entry point (pipeline.hip) for several
lane / chunk settings, next to the resident-input time.  usage: python tools/e2e_sweep.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nadavca_amd import dtw, synthetic, _lib
from nadavca_amd.device import DeviceBatch, refine_alignment_dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = synthetic.load_model_arrays()
batch = synthetic.make_batch(n, model, seed=1000, R=400, R_spread=40, bandwidth=150)
flat = dtw.FlatBatch.from_arrays(batch.signal, batch.sig_off, batch.reference, batch.ref_off, batch.context_before,
                                 batch.cb_off, batch.context_after, batch.ca_off, batch.anchors, batch.anc_off)
ctx = _lib.Context(0)
km = dtw.KmerModel(*model, context=ctx)
db = DeviceBatch(batch, torch.device('cuda', 0))
ev0, st0 = refine_alignment_dev(db, 150, 2, km, True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    refine_alignment_dev(db, 150, 2, km, True, ev0, st0)
torch.cuda.synchronize()
res_ms = (time.perf_counter() - t) / 5 * 1e3
print('resident: %.2f ms' % res_ms, flush=True)
want = ev0.cpu().numpy()
for lanes, chunks in [(1, 1), (2, 2), (2, 4), (3, 4), (3, 6), (3, 8), (4, 8), (4, 12), (4, 16), (3, 12)]:
    os.environ['NADAVCA_E2E_LANES'] = str(lanes)
    os.environ['NADAVCA_E2E_CHUNKS'] = str(chunks)
    os.environ['NADAVCA_E2E_MIN_READS'] = '1'
    c2 = _lib.Context(0)
    k2 = dtw.KmerModel(*model, context=c2)
    ev, st = dtw.refine_alignment_flat(flat, 150, 2, k2, True)
    assert np.array_equal(ev, want), 'results differ'
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        dtw.refine_alignment_flat(flat, 150, 2, k2, True)
        ts.append((time.perf_counter() - t) * 1e3)
    print('lanes %d chunks %2d: e2e %.2f ms (min %.2f)  = %.2fx resident' % (lanes, chunks, float(np.median(ts)), min(ts), float(np.median(ts)) / res_ms), flush=True)
    k2.close(); c2.close()
