"""The N > 1 path of ``estimate_snps_batch`` on the device: two processes share the one GPU of the test box (gloo
for the exchange — RCCL refuses two ranks on one device — so the collective buffers pass through the host; with the
nccl backend they stay on the GPU, distributed.collective_device), each with its own shard of the reads.
Checked against the same call over all reads in one process: the pooled median / MAD (exact distributed
selection, nvk_select_hist_dev) bit-equal, chunk ranges and coverage equal, posteriors to 1e-12.
Reference: /root/reference/nadavca/estimate_snps.py:57-70, read.py:68-81, estimator.py:199-236."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_READS, GENOME = 240, 3000


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    rb, aligner, genome = synthetic.make_read_batch(N_READS, model, seed=77, genome_length=GENOME, length=200,
                                                    spread=30)
    return model, rb, aligner, genome


def _shard_batch(rb, ba, lo, hi):
    from nadavca_amd.readbatch import ReadBatch, BaseAlignmentBatch, SyntheticBatchAligner
    cut = lambda arr, off: arr[off[lo]:off[hi]]
    reb = lambda off: off[lo:hi + 1] - off[lo]
    rb2 = ReadBatch(cut(rb.raw_signal, rb.sig_off), reb(rb.sig_off), cut(rb.sequence, rb.seq_off), reb(rb.seq_off),
                    cut(rb.map_base, rb.map_off), cut(rb.map_sig, rb.map_off), reb(rb.map_off))
    ba2 = BaseAlignmentBatch(cut(ba.read_idx, ba.off), cut(ba.ref_idx, ba.off), reb(ba.off), ba.reverse[lo:hi])
    return rb2, ba2


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nadavca_amd import dtw, _lib, distributed as D
    from nadavca_amd.estimate_snps import estimate_snps_batch
    from nadavca_amd.readbatch import SyntheticBatchAligner
    from nadavca_amd.device import select_hist_dev
    model, rb, aligner, genome = _setup()
    ctx = _lib.Context(0)
    km = dtw.KmerModel(*model, context=ctx)
    lo, hi = D.shard_bounds(rb.n, rank, world)
    rb2, ba2 = _shard_batch(rb, aligner.get_base_alignments(rb), lo, hi)
    cfg = dict(bandwidth=150, snp_prior_probability=0.001, min_event_length=2, model_wobbling=True,
               model_transitions=True, tweak_signal_normalization=False, normalization_event_length=10)
    raw = torch.from_numpy(rb2.raw_signal).to('cuda:0').to(torch.float64)
    cs = D.pooled_centre_scale(select_hist_dev(ctx, raw), raw.numel(), device=raw.device)
    chunks = estimate_snps_batch(genome, rb2, config=cfg, kmer_model=km, independent=False,
                                 aligner=SyntheticBatchAligner(genome, ba2), distributed=True, dst=0)
    if rank == 0:
        np.savez(os.path.join(tmp, 'dist.npz'), cs=np.array(cs), n=len(chunks),
                 **{'r%d' % i: np.array([c.start, c.end]) for i, c in enumerate(chunks)},
                 **{'v%d' % i: c.values for i, c in enumerate(chunks)},
                 **{'c%d' % i: c.coverage for i, c in enumerate(chunks)})
    else:
        assert chunks is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / 'dist.npz')
    from nadavca_amd import dtw, _lib
    from nadavca_amd.estimate_snps import estimate_snps_batch
    model, rb, aligner, genome = _setup()
    km = dtw.KmerModel(*model, context=_lib.default_context())
    cfg = dict(bandwidth=150, snp_prior_probability=0.001, min_event_length=2, model_wobbling=True,
               model_transitions=True, tweak_signal_normalization=False, normalization_event_length=10)
    one = estimate_snps_batch(genome, rb, config=cfg, kmer_model=km, independent=False, aligner=aligner,
                              distributed=False)
    x = rb.raw_signal.astype(np.float64)
    c = np.median(x)
    assert z['cs'].tolist() == [c, np.median(np.abs(x - c))]          # the pooled statistics, bit for bit
    assert int(z['n']) == len(one) and len(one) >= 1
    for i, ch in enumerate(one):
        assert z['r%d' % i].tolist() == [ch.start, ch.end]
        assert np.array_equal(z['c%d' % i], ch.coverage)
        assert np.max(np.abs(z['v%d' % i] - ch.values)) < 1e-12
