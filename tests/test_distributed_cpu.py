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


def _synthetic_chunks(n_reads=48, L=600, seed=7):
    """Per-read (start, end, normalised log-likelihood rows) as ``_estimate_log_likelihoods`` would hand them
    to the consensus sum (estimator.py:112-121): overlapping intervals, a gap, touching intervals."""
    rng = np.random.default_rng(seed)
    chunks = []
    for i in range(n_reads):
        start = int(rng.integers(0, 250)) if i % 3 else int(rng.integers(330, 520))
        end = min(L, start + int(rng.integers(40, 90)))
        chunks.append((start, end, rng.normal(-3.0, 2.0, size=(end - start, 4))))
    chunks.append((300, 330, rng.normal(size=(30, 4))))   # touches the next group's first start
    return chunks, L


def _local_sums(chunks, L):
    acc = np.zeros((L, 4))
    cov = np.zeros(L, dtype=np.int64)
    for s, e, v in chunks:
        acc[s:e] += v
        cov[s:e] += 1
    return acc, cov, [(s, e) for s, e, _ in chunks]


def _merge_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nadavca_amd import distributed as D
    chunks, L = _synthetic_chunks()
    mine = D.shard(chunks, rank, world)                      # contiguous blocks of reads per rank
    acc, cov, ranges = _local_sums(mine, L)
    merged = D.merge_consensus(acc, cov, ranges, dst=0)
    tens = D.reduce_consensus_tensors(torch.from_numpy(acc), torch.from_numpy(cov), dst=0)
    if rank == 0:
        tacc, tcov, groups, seg_off, ll_cat = merged
        np.savez(os.path.join(tmp, 'merged.npz'), tacc=tacc, tcov=tcov, groups=np.array(groups), seg_off=seg_off,
                 ll_cat=ll_cat, t_acc=tens[0].numpy(), t_cov=tens[1].numpy())
    else:
        assert merged is None and tens is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_rank(tmp_path):
    """The exchange step of estimate_snps(independent=False) end to end on two gloo ranks — local sums of
    each rank's shard, interval all-gather, ONE packed reduce, grouping, the posterior's inputs — against
    the same over all reads on one rank (estimator.py:205-235): sums to 1e-12, everything else exact."""
    import torch.multiprocessing as mp
    from nadavca_amd.estimator import ProbabilityEstimator
    port = _free_port()
    mp.spawn(_merge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / 'merged.npz')
    chunks, L = _synthetic_chunks()
    acc, cov, ranges = _local_sums(chunks, L)
    groups = ProbabilityEstimator.group_ranges(ranges)
    seg_off = np.concatenate([[0], np.cumsum([e - s for s, e in groups])])
    assert np.allclose(got['tacc'], acc, rtol=0, atol=1e-12) and np.array_equal(got['tcov'], cov)
    assert np.allclose(got['t_acc'], acc, rtol=0, atol=1e-12) and np.array_equal(got['t_cov'], cov)
    assert [tuple(g) for g in got['groups'].tolist()] == groups and len(groups) >= 2
    assert np.array_equal(got['seg_off'], seg_off)
    assert np.allclose(got['ll_cat'], np.concatenate([acc[s:e] for s, e in groups]), rtol=0, atol=1e-12)


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


def _median_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nadavca_amd import distributed as D
    out = []
    for case, x in enumerate(_median_cases()):
        lo, hi = D.shard_bounds(x.size, rank, world) if case != 3 else ((0, x.size) if rank == 0 else (0, 0))
        mine = x[lo:hi]                                       # (case 3: one rank holds nothing)
        out.append(D.pooled_centre_scale(D.numpy_hist(mine), mine.size))
    np.save(os.path.join(tmp, 'median_r%d.npy' % rank), np.array(out))
    dist.barrier()
    dist.destroy_process_group()


def _median_cases():
    rng = np.random.default_rng(99)
    adc = np.rint(12.0 * rng.normal(0.0, 1.3, 40001) + 90.0)          # ADC counts: heavy duplication, odd count
    return [adc, adc[:-1].copy(), rng.normal(0.0, 1.0, 1000) * 1e-3, adc[:257].copy(),
            np.concatenate([rng.normal(0, 1, 64), [0.0, -0.0, 5e300, -5e300]])]


def test_two_rank_pooled_median_and_mad_equal_numpy(tmp_path):
    """estimate_snps normalises ALL reads with one median / MAD (estimate_snps.py:61, read.py:68-81).  With the
    samples sharded over ranks that is an exact distributed selection: per radix pass only 256 counts cross ranks
    (distributed.pooled_centre_scale; on the GPU the counts come from nvk_select_hist_dev).  Two gloo ranks on CPU
    tensors: shift and scale bit-equal to numpy.median over the union, on both ranks."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_median_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'median_r0.npy'), np.load(tmp_path / 'median_r1.npy')
    assert np.array_equal(r0, r1)
    for (c, s), x in zip(r0, _median_cases()):
        assert c == np.median(x) and s == np.median(np.abs(x - c))
