"""N>1 path on CPU: 2 ranks over gloo exercise the sharding, the interval all-gather and the
packed consensus reduce of nadavca_amd.distributed (no GPU compute involved)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nadavca_amd import distributed as D
    L = 50
    rng = np.random.default_rng(100 + rank)
    acc = rng.normal(size=(L, 4))
    cov = rng.integers(0, 5, L)
    ranges = [(3 + rank, 10 + rank), (20, 30)] if rank == 0 else [(9, 15)]
    allr = D.gather_ranges(ranges)
    tot = D.reduce_consensus(acc, cov, dst=0)
    lo, hi = D.shard_bounds(11, rank, world)
    np.savez(os.path.join(tmp, 'r%d.npz' % rank), acc=acc, cov=cov, allr=np.array(allr), lo=lo, hi=hi,
             tot_acc=tot[0] if tot is not None else np.zeros(0),
             tot_cov=tot[1] if tot is not None else np.zeros(0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_consensus_exchange(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'r0.npz')
    r1 = np.load(tmp_path / 'r1.npz')
    assert np.allclose(r0['tot_acc'], r0['acc'] + r1['acc'])
    assert np.array_equal(r0['tot_cov'], r0['cov'] + r1['cov'])
    assert r1['tot_acc'].size == 0                      # only the root receives the totals
    want = [(3, 10), (20, 30), (9, 15)]
    assert [tuple(x) for x in r0['allr'].tolist()] == want == [tuple(x) for x in r1['allr'].tolist()]
    assert (int(r0['lo']), int(r0['hi']), int(r1['lo']), int(r1['hi'])) == (0, 6, 6, 11)


def test_group_ranges_strict_overlap():
    from nadavca_amd.estimator import ProbabilityEstimator
    g = ProbabilityEstimator.group_ranges
    assert g([(9, 15), (3, 10), (20, 30)]) == [(3, 15), (20, 30)]
    assert g([(0, 5), (5, 9)]) == [(0, 5), (5, 9)]      # touching chunks do not merge
    assert g([(0, 10), (2, 4), (9, 12)]) == [(0, 12)]
    assert g([]) == []


def test_shard_bounds_cover_everything():
    from nadavca_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 200000):
        for w in (1, 2, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
