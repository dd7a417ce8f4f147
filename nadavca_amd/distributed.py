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


def collective_device(like=None, group=None):
    """Where a collective's buffer has to live: on the GPU for the nccl (= RCCL) backend — the sums then never
    leave the device — on the host for gloo (the CPU tests, and several ranks sharing one GPU in the GPU tests)."""
    torch, dist = _torch_dist()
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'nccl':
        if like is not None and like.is_cuda:
            return like.device
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


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
    if device is not None:
        device = collective_device(torch.empty(0, device=device), group)
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
    home = packed.device
    packed = packed.to(collective_device(packed, group))
    dist.reduce(packed, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank(group) != dst:
        return None
    packed = packed.to(home)
    return packed[:, :-1].contiguous(), torch.round(packed[:, -1]).to(torch.int64)


# ---- ONE median / MAD over the samples of ALL ranks (estimate_snps.py:61, read.py:68-81) ---------------------------
# Exact, not approximate: a radix select over the order-preserving 64-bit key of a double, 8 passes of 8 bits.  Per
# pass every rank counts its own samples by the byte of that pass (nvk_select_hist_dev on the GPU; numpy_hist below
# restates it for the CPU tests), the 256 counts are summed over the ranks — the only thing that crosses xGMI: 2 KB
# per pass — and every rank picks the same bucket from the same totals.  32 all-reduces of 2 KB at most (two
# selections each for the median and the MAD of an even count), latency-bound, once per estimate_snps call.
def key_of(x):
    """Order-preserving uint64 image of float64 values (ascending doubles -> ascending keys)."""
    b = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | np.uint64(1 << 63))


def val_of(key):
    k = np.uint64(key)
    b = (k & np.uint64((1 << 63) - 1)) if (k >> np.uint64(63)) else ~k
    return float(np.array([b], dtype=np.uint64).view(np.float64)[0])


def numpy_hist(values):
    """-> local_hist(mode, centre, prefix, pass) over a numpy array: what nvk_select_hist_dev computes."""
    values = np.ascontiguousarray(values, dtype=np.float64)

    def local_hist(mode, centre, prefix, p):
        import torch
        k = key_of(np.abs(values - centre) if mode else values)
        shift = np.uint64(56 - 8 * p)
        if p:
            keep = (k >> (shift + np.uint64(8))) == (np.uint64(prefix) >> (shift + np.uint64(8)))
            k = k[keep]
        h = np.bincount(((k >> shift) & np.uint64(255)).astype(np.int64), minlength=256)
        return torch.from_numpy(h.astype(np.int64))
    return local_hist


def pooled_select(local_hist, mode, centre, rank_k, group=None):
    """Key of the ``rank_k``-th smallest (0-based) of f(x) over the samples of all ranks."""
    torch, dist = _torch_dist()
    prefix = 0
    for p in range(8):
        h = local_hist(mode, centre, prefix, p)
        if dist.is_available() and dist.is_initialized():
            h = h.to(collective_device(h, group))
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        c = np.cumsum(h.cpu().numpy().astype(np.int64))
        b = int(np.searchsorted(c, rank_k, side='right'))
        if b > 255:
            raise ValueError('pooled_select: rank beyond the number of samples')
        rank_k -= int(c[b - 1]) if b else 0
        prefix |= b << (56 - 8 * p)
    return prefix


def pooled_median(local_hist, n_total, mode=0, centre=0.0, group=None):
    """statistics.median / numpy.median of f(x) over all ranks' samples: the middle one, or the mean of the two."""
    hi = val_of(pooled_select(local_hist, mode, centre, n_total // 2, group))
    if n_total & 1:
        return hi
    lo = val_of(pooled_select(local_hist, mode, centre, n_total // 2 - 1, group))
    return (lo + hi) / 2


def pooled_centre_scale(local_hist, n_local, device=None, group=None):
    """(median, median |x - median|) over the union of all ranks' samples — Read.normalize_reads' shift and scale."""
    torch, dist = _torch_dist()
    n = torch.tensor([int(n_local)], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        n = n.to(collective_device(None if device is None else torch.empty(0, device=device), group))
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    n_total = int(n.item())
    if n_total == 0:
        return float('nan'), float('nan')
    centre = pooled_median(local_hist, n_total, 0, 0.0, group)
    scale = pooled_median(local_hist, n_total, 1, centre, group)
    return centre, scale


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
