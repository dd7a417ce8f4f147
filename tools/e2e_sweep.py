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
def run(lanes, chunks, growth):
    os.environ['NADAVCA_E2E_LANES'] = str(lanes)
    os.environ['NADAVCA_E2E_CHUNKS'] = str(chunks)
    os.environ['NADAVCA_E2E_GROWTH'] = str(growth)
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
    print('lanes %d chunks %2d growth %.2f: single call e2e %.2f ms (min %.2f) = %.2fx resident'
          % (lanes, chunks, growth / 100, float(np.median(ts)), min(ts), float(np.median(ts)) / res_ms), flush=True)
    return c2, k2


for cfg in [(1, 1, 100), (2, 2, 100), (2, 2, 200), (2, 2, 300), (3, 3, 150), (3, 3, 200), (3, 3, 300), (3, 4, 150),
            (3, 4, 200), (4, 4, 200), (4, 5, 170), (3, 5, 170), (2, 3, 200), (2, 4, 200)]:
    c2, k2 = run(*cfg)
    k2.close(); c2.close()

# a stream of batches (submit / wait): steady state
for lanes in (2, 3):
    os.environ['NADAVCA_E2E_LANES'] = str(lanes)
    c2 = _lib.Context(0)
    k2 = dtw.KmerModel(*model, context=c2)
    rs = dtw.RefineStream(k2, 150, 2, True)
    outs = [(np.zeros((int(flat.ref_off[-1]), 2), np.int32), np.zeros(flat.n, np.int32), np.zeros(flat.n, np.int32))
            for _ in range(lanes)]
    for depth in range(1, lanes + 1):
        for rep in range(2):
            K = 8
            t = time.perf_counter()
            tk = []
            for i in range(K):
                tk.append(rs.submit(flat, out=outs[i % lanes]))
                if i >= depth - 1 and depth > 0:
                    j = i - (depth - 1)
                    ev, st, ti = rs.wait(tk[j])
                    if depth > 1 or True:
                        pass
            # (everything waited for when depth == 1; else the tail)
            for j in range(K - (depth - 1), K):
                ev, st, ti = rs.wait(tk[j])
            dt = (time.perf_counter() - t) / K * 1e3
        assert np.array_equal(ev, want)
        print('stream lanes %d in-flight %d: %.2f ms per batch = %.2fx resident' % (lanes, depth, dt, dt / res_ms), flush=True)
    k2.close(); c2.close()
