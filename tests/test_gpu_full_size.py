"""GPU, BASELINE.json's full sizes: config 2 (10 000 reads, ~4 300 samples, bandwidth 150) through the
resident-input entry points bench.py times, checked by size-independent properties and by the oracle on a
sample; config 3 at the same shape on 2 000 reads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def big():
    import torch
    from nadavca_amd import dtw, synthetic, _lib
    from nadavca_amd.device import DeviceBatch
    model = synthetic.load_model_arrays()
    ctx = _lib.Context(0)
    km = dtw.KmerModel(*model, context=ctx)
    batch = synthetic.make_batch(10000, model, seed=2024, R=400, R_spread=40, bandwidth=150)
    dev = torch.device('cuda', 0)
    return dict(torch=torch, model=model, ctx=ctx, km=km, batch=batch, dev=dev, dbatch=DeviceBatch(batch, dev))


def test_config2_full_size_properties(big, oracle_port):
    from nadavca_amd.device import refine_alignment_dev
    torch, db, batch = big['torch'], big['dbatch'], big['batch']
    ev = torch.zeros((db.total_ref, 2), dtype=torch.int32, device=big['dev'])
    st = torch.zeros(db.n, dtype=torch.int32, device=big['dev'])
    refine_alignment_dev(db, 150, 2, big['km'], True, ev, st)
    first = ev.cpu().numpy().copy()
    assert int((st != 0).sum().item()) == 0                      # every read has a path
    refine_alignment_dev(db, 150, 2, big['km'], True, ev, st)    # idempotent (work order, retries, atomics)
    assert np.array_equal(first, ev.cpu().numpy())
    off = batch.ref_off
    n_sig = np.diff(batch.sig_off)
    starts, ends = first[:, 0].astype(np.int64), first[:, 1].astype(np.int64)
    assert np.all(ends - starts >= 2)                            # min_event_length
    inner = np.ones(first.shape[0], dtype=bool)
    inner[off[1:-1]] = False                                     # first event of each read has no predecessor
    assert np.all(starts[1:][inner[1:]] >= ends[:-1][inner[1:]])  # events ordered inside a read
    assert np.all(starts[off[:-1]] >= 0)
    assert np.all(ends[off[1:] - 1] <= n_sig)                    # inside the read's signal slice
    # the simulated truth: event starts within a few samples of the true starts almost everywhere
    err = []
    for i in range(0, batch.n, 97):
        c = batch.cases[i]
        err.append(np.abs(first[off[i]:off[i + 1], 0] - c['true_starts'][:-1]))
    assert np.mean(np.concatenate(err) <= 3) > 0.9
    # and the reference itself on a sample of the same reads
    mo = oracle_port.KmerModel(*big['model'])
    for i in range(0, batch.n, 313):
        c = batch.cases[i]
        exp = oracle_port.refine_alignment(c['signal'], c['reference'], c['context_before'], c['context_after'],
                                           c['approximate_alignment'], 150, 2, mo, True)
        assert np.array_equal(first[off[i]:off[i + 1]], exp)


def test_config3_shape_properties(big, oracle_port):
    from nadavca_amd import synthetic
    from nadavca_amd.device import DeviceBatch, estimate_log_likelihoods_dev
    torch = big['torch']
    batch = synthetic.Batch(big['batch'].cases[:2000])
    db = DeviceBatch(batch, big['dev'])
    ll = torch.zeros((db.total_ref, 4), dtype=torch.float64, device=big['dev'])
    st = torch.zeros(db.n, dtype=torch.int32, device=big['dev'])
    estimate_log_likelihoods_dev(db, 150, 2, big['km'], True, ll, st)
    out = ll.cpu().numpy()
    assert int((st != 0).sum().item()) == 0 and np.all(np.isfinite(out))
    ref = batch.reference
    col = out[np.arange(ref.size), ref]
    off = batch.ref_off
    for i in range(0, batch.n, 41):                              # one no-substitution likelihood per read
        assert np.all(col[off[i]:off[i + 1]] == col[off[i]])
    assert np.mean(np.argmax(out, axis=1) == ref) > 0.97        # clean synthetic data: the true base wins
    mo = oracle_port.KmerModel(*big['model'])
    for i in (0, 999, 1999):
        c = batch.cases[i]
        exp = oracle_port.estimate_log_likelihoods(c['signal'], c['reference'], c['context_before'],
                                                   c['context_after'], c['approximate_alignment'], 150, 2, mo, True)
        assert np.allclose(out[off[i]:off[i + 1]], exp, rtol=1e-9, atol=1e-9)
