"""One node, one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Reads are independent units of work (the reference loops over them serially,
/root/reference/nadavca/align_signal.py:52 and estimator.py:201), so they shard across ranks with
no data-path communication for ``align_signal`` and ``estimate_snps(independent=True)``.  The one
exchange step is the ``independent=False`` consensus: every rank sums its reads' normalised
log-likelihoods into a local [L_ref, 4] array (+ coverage), and ONE reduce(sum) over a packed
[L_ref, 5] f64 buffer combines them (coverage rides along as f64 — exact below 2^53).  Chunk
grouping needs every read's interval, a tiny all-gather of integer pairs.  The posterior then runs
on the root.
"""
import numpy as np


def shard_bounds(n_items, rank, world_size):
    """Contiguous block [lo, hi) of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(items, rank, world_size):
    lo, hi = shard_bounds(len(items), rank, world_size)
    return items[lo:hi]


def _torch_dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def reduce_consensus(acc, cov, dst=0, device=None, group=None):
    """Sum (acc, cov) over all ranks onto ``dst``.  Returns the totals on ``dst`` and None elsewhere.
    ``device``: torch device for the collective buffer (a cuda device for the nccl/RCCL backend,
    None/cpu for gloo)."""
    torch, dist = _torch_dist()
    packed = np.concatenate([np.asarray(acc, dtype=np.float64),
                             np.asarray(cov, dtype=np.float64)[:, None]], axis=1)
    t = torch.from_numpy(np.ascontiguousarray(packed))
    if device is not None:
        t = t.to(device)
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank(group) != dst:
        return None
    out = t.cpu().numpy()
    return out[:, :-1].copy(), np.rint(out[:, -1]).astype(np.int64)


def gather_ranges(ranges, device=None, group=None):
    """All ranks' chunk intervals, concatenated in rank order (every rank gets the full list)."""
    torch, dist = _torch_dist()
    world = dist.get_world_size(group)
    mine = np.asarray(ranges, dtype=np.int64).reshape(-1, 2)
    counts = torch.zeros(world, dtype=torch.int64)
    counts[dist.get_rank(group)] = mine.shape[0]
    if device is not None:
        counts = counts.to(device)
    dist.all_reduce(counts, group=group)
    counts = counts.cpu().numpy()
    width = int(counts.max()) if world else 0
    buf = torch.full((max(width, 1), 2), -1, dtype=torch.int64)
    buf[:mine.shape[0]] = torch.from_numpy(mine)
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r in range(world):
        rows = parts[r].cpu().numpy()[:int(counts[r])]
        out.extend((int(a), int(b)) for a, b in rows)
    return out


def reduce_consensus_tensors(acc, cov, dst=0, group=None):
    """The same exchange for torch tensors that already live where the collective runs (cuda tensors with
    the nccl/RCCL backend: the per-position sums never leave the device; cpu tensors with gloo): ONE
    reduce(sum) of the packed (L, alphabet + 1) f64 buffer.  -> (acc, cov) tensors on ``dst``, None elsewhere."""
    torch, dist = _torch_dist()
    packed = torch.cat([acc.to(torch.float64), cov.to(torch.float64).unsqueeze(1)], dim=1).contiguous()
    dist.reduce(packed, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank(group) != dst:
        return None
    return packed[:, :-1].contiguous(), torch.round(packed[:, -1]).to(torch.int64)


def merge_consensus(acc, cov, ranges, dst=0, device=None, group=None):
    """The exchange step of ``independent=False`` and nothing else (no GPU involved: covered by the gloo
    tests): every rank contributes its per-position sums, coverage and chunk intervals; ``dst`` receives
    what the posterior needs — (total acc, total cov, groups, seg_off, ll laid end to end per group) — the
    other ranks None.  Grouping follows estimator.py:205-220 over the union of all ranks' intervals."""
    from .estimator import ProbabilityEstimator
    all_ranges = gather_ranges(ranges, device=device, group=group)
    total = reduce_consensus(acc, cov, dst=dst, device=device, group=group)
    if total is None:
        return None
    tacc, tcov = total
    groups = ProbabilityEstimator.group_ranges(all_ranges)
    seg_off = np.zeros(len(groups) + 1, dtype=np.int64)
    np.cumsum([e - s for s, e in groups], out=seg_off[1:])
    ll_cat = np.concatenate([tacc[s:e] for s, e in groups]) if groups else np.zeros((0, tacc.shape[1]))
    return tacc, tcov, groups, seg_off, ll_cat


def estimate_probabilities_distributed(estimator, reference, local_reads, dst=0, device=None, group=None):
    """``ProbabilityEstimator.estimate_probabilities`` over the union of all ranks' reads.
    Every rank passes its own shard; the Chunk list is returned on ``dst`` (None elsewhere)."""
    acc, cov, ranges = estimator.local_consensus(reference, local_reads)
    merged = merge_consensus(acc, cov, ranges, dst=dst, device=device, group=group)
    if merged is None:
        return None
    tacc, tcov, groups, seg_off, ll_cat = merged
    return estimator.posterior_of_groups(reference, tcov, groups, seg_off, ll_cat)
