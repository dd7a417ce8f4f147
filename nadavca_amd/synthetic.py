"""Deterministic synthetic workloads for tests and bench.py (recipe: SURVEY.md §8d).

Nothing here comes from the reference: there is no sample data offline, so reads are
simulated from a k-mer table — per base a dwell of U{lo..hi} samples at the k-mer's
expected level plus Gaussian noise, anchors = a random subset of the true
base->sample map, jittered, exactly the shape
``ApproximateAligner.get_signal_alignment`` hands to the DP
(/root/reference/nadavca/alignment.py:142-186).
"""
import os

import numpy as np

from . import defaults


def load_model_arrays(path=None):
    """-> (k, central_position, alphabet_size, mean[4^k], sigma[4^k]) from an .npz table."""
    z = np.load(path or defaults.KMER_MODEL_FILE)
    return int(z['k']), int(z['central_pos']), int(z['alphabet_size']), \
        np.array(z['mean'], dtype=np.float64), np.array(z['sigma'], dtype=np.float64)


def synth_model_arrays(seed=0, k=6, central=2, alphabet=4):
    """A random table with the real 6-mer table's moments (mean sd 1.26, sigma 0.33288)."""
    rng = np.random.default_rng(seed)
    n = alphabet ** k
    return k, central, alphabet, rng.normal(0.0, 1.26, n), np.full(n, 0.3328800486427912)


def kmer_ids(seq_ext, offset, length, k, central, alphabet=4):
    """k-mer id of positions 0..length-1 of the sequence that starts at ``offset`` inside
    ``seq_ext`` (out-of-range bases read as 0, as the reference's ExtendedSequence does)."""
    seq_ext = np.asarray(seq_ext, dtype=np.int64)
    ids = np.zeros(length, dtype=np.int64)
    pos = np.arange(length) + offset - central
    for m in range(k):
        p = pos + m
        ok = (p >= 0) & (p < seq_ext.size)
        b = np.where(ok, seq_ext[np.clip(p, 0, max(seq_ext.size - 1, 0))] if seq_ext.size else 0, 0)
        ids = ids * alphabet + b
    return ids


def homopolymer_rich(rng, n, alphabet, mean_run=3.0, long_run_share=0.1, long_run=(7, 12)):
    """A base sequence made of runs: geometric run lengths (mean ``mean_run``), a share of them long —
    real genomes hold many runs of k+1 equal bases, iid sequences almost none (P = 4^-k per position)."""
    out = np.empty(0, dtype=np.int64)
    prev = -1
    while out.size < n:
        b = int(rng.integers(0, alphabet - 1))
        b += (b >= prev) if prev >= 0 else 0          # a new run never repeats the previous base
        run = int(rng.integers(long_run[0], long_run[1] + 1)) if rng.random() < long_run_share \
            else int(rng.geometric(1.0 / mean_run))
        out = np.concatenate([out, np.full(run, b, dtype=np.int64)])
        prev = b
    return out[:n]


def make_dp_case(rng, model, R=400, bandwidth=150, dwell=(3, 17), noise=0.35,
                 anchor_density=0.75, jitter=20, with_context=True, trim=3, pad_bases=None, bases=None):
    """One DP problem in the argument shape of ``dtw.refine_alignment`` /
    ``dtw.estimate_log_likelihoods``.  Returns a dict with signal (f64), reference,
    context_before, context_after (int32), approximate_alignment (int32 (A,2)),
    and the true event starts (slice coordinates) for sanity checks."""
    k, central, alphabet, mean, sigma = model
    lo, hi = dwell
    if pad_bases is None:
        pad_bases = int(np.ceil(bandwidth / ((lo + hi) / 2.0))) + 6
    # `bases`: callable (rng, n, alphabet) -> base codes, default iid uniform
    full = rng.integers(0, alphabet, R + 2 * pad_bases) if bases is None else \
        np.asarray(bases(rng, R + 2 * pad_bases, alphabet), dtype=np.int64)
    ids = kmer_ids(full, 0, full.size, k, central, alphabet)
    dw = rng.integers(lo, hi + 1, full.size)
    starts = np.concatenate([[0], np.cumsum(dw)])
    total = int(starts[-1])
    x = np.repeat(mean[ids], dw) + rng.normal(0.0, noise, total)
    x = np.clip(x, -5.0, 5.0)

    # anchors: matched bases of the R-base segment, thinned, jittered, monotone
    cand = np.arange(trim, R - trim) if R > 2 * trim + 1 else np.arange(R)
    keep = cand[rng.random(cand.size) < anchor_density]
    keep = np.unique(np.concatenate([[0], keep, [R - 1]])) if R > 0 else keep
    sig_idx = starts[pad_bases + keep] + rng.integers(-jitter, jitter + 1, keep.size)
    sig_idx = np.maximum.accumulate(np.clip(sig_idx, 0, total - 1))
    s_first, s_last = int(sig_idx[0]), int(sig_idx[-1])
    a0 = max(0, s_first - bandwidth)
    a1 = min(total, s_last + 1 + bandwidth)
    anchors = np.stack([sig_idx - a0, keep], axis=1).astype(np.int32)
    ref = full[pad_bases:pad_bases + R].astype(np.int32)
    if with_context:
        cb = full[pad_bases - central:pad_bases].astype(np.int32)
        ca = full[pad_bases + R:pad_bases + R + k - central - 1].astype(np.int32)
    else:
        cb = np.zeros(0, dtype=np.int32)
        ca = np.zeros(0, dtype=np.int32)
    return dict(signal=np.ascontiguousarray(x[a0:a1]), reference=ref, context_before=cb,
                context_after=ca, approximate_alignment=anchors,
                true_starts=(starts[pad_bases:pad_bases + R + 1] - a0).astype(np.int64))


class Batch:
    """Flat (CSR-style) batch of DP problems — the layout of the C-ABI batch calls
    (include/nadavca_hip.h)."""

    def __init__(self, cases):
        self.n = len(cases)
        self.cases = cases
        cat = lambda key, dt: (np.concatenate([np.asarray(c[key]).reshape(-1) for c in cases]).astype(dt)
                               if cases else np.zeros(0, dtype=dt))
        off = lambda key, div=1: np.concatenate(
            [[0], np.cumsum([np.asarray(c[key]).size // div for c in cases])]).astype(np.int64)
        self.signal = cat('signal', np.float64)
        self.sig_off = off('signal')
        self.reference = cat('reference', np.int32)
        self.ref_off = off('reference')
        self.context_before = cat('context_before', np.int32)
        self.cb_off = off('context_before')
        self.context_after = cat('context_after', np.int32)
        self.ca_off = off('context_after')
        self.anchors = cat('approximate_alignment', np.int32)
        self.anc_off = off('approximate_alignment', 2)


def make_batch(n_reads, model, seed=0, R=400, R_spread=40, **kw):
    """``n_reads`` independent DP problems, read i drawn from default_rng([seed, i])."""
    cases = []
    for i in range(n_reads):
        rng = np.random.default_rng([seed, i])
        r = int(R + rng.integers(-R_spread, R_spread + 1)) if R_spread else int(R)
        cases.append(make_dp_case(rng, model, R=r, **kw))
    return Batch(cases)


def tile_batch(batch, n):
    """``batch`` repeated to ``n`` reads (whole copies, then a prefix): BASELINE config 4 asks for 200 000 reads, and
    simulating each of them in Python would take minutes; throughput does not care that read i + len(batch) is
    read i again.  ``cases`` stays the unique list (the CPU baseline samples from it)."""
    if n <= batch.n:
        return Batch(batch.cases[:n])
    reps, rem = divmod(n, batch.n)
    head = Batch(batch.cases[:rem]) if rem else None
    out = Batch.__new__(Batch)
    out.n, out.cases = n, batch.cases

    def rep(name):
        parts = [getattr(batch, name)] * reps + ([getattr(head, name)] if head is not None else [])
        return np.concatenate(parts)

    def rep_off(name):
        off = getattr(batch, name)
        lens = np.diff(off)
        lens = np.concatenate([lens] * reps + ([np.diff(getattr(head, name))] if head is not None else []))
        return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    for a, o in (('signal', 'sig_off'), ('reference', 'ref_off'), ('context_before', 'cb_off'),
                 ('context_after', 'ca_off'), ('anchors', 'anc_off')):
        setattr(out, a, rep(a))
        setattr(out, o, rep_off(o))
    return out


# BASELINE.json configs 2-5 (sizes: SURVEY.md §8 header)
WORKLOADS = {
    'cfg2_align': dict(n_reads=10000, R=400, R_spread=40, bandwidth=150),
    'cfg3_snps': dict(n_reads=10000, R=400, R_spread=40, bandwidth=150),
    # (n_reads: the whole job's 200 000 reads; bench.py gives every rank 200 000 / world of them)
    'cfg4_consensus': dict(n_reads=200000, R=400, R_spread=40, bandwidth=150, reference_length=10000),
    'cfg5_long': dict(n_reads=64, R=5000, R_spread=500, bandwidth=1000),
}


def env_int(name, default):
    v = os.environ.get(name)
    return int(v) if v else default


# ------------------------------------------------------------------------------------------------
# Read-level synthetic data: reads sampled from a genome on both strands + an aligner that returns
# the true base mapping (stands in for BWA, which does not exist offline)
# ------------------------------------------------------------------------------------------------
_BASES = np.array(['A', 'C', 'G', 'T'])
_COMPLEMENT_NUM = np.array([3, 2, 1, 0])


def make_genome(length, seed):
    """iid uniform ACGT reference as an array of single characters (Genome.bases layout)."""
    rng = np.random.default_rng(seed)
    return _BASES[rng.integers(0, 4, length)]


def make_read_spec(rng, genome_num, model, index, length=400, spread=40, dwell=(3, 17), noise=0.35,
                   anchor_density=0.75, jitter=20, trim=3, substitution_rate=0.0, raw_scale=12.0,
                   raw_shift=90.0):
    """Plain-data description of one simulated read (no classes): raw signal, basecalled sequence,
    base->sample map, and the true base mapping in the shape ``_get_base_alignment`` returns."""
    k, central, alphabet, mean, sigma = model
    G = genome_num.size
    L = int(min(G, length + rng.integers(-spread, spread + 1)))
    g0 = int(rng.integers(0, G - L + 1))
    reverse = bool(index % 2)
    seg = genome_num[g0:g0 + L]
    oriented = _COMPLEMENT_NUM[seg][::-1] if reverse else seg.copy()   # read orientation
    seq = oriented.copy()
    subs = rng.random(L) < substitution_rate
    seq[subs] = (seq[subs] + rng.integers(1, 4, int(subs.sum()))) % 4
    ids = kmer_ids(seq, 0, L, k, central, alphabet)
    dw = rng.integers(dwell[0], dwell[1] + 1, L)
    starts = np.concatenate([[0], np.cumsum(dw)])
    x = np.clip(np.repeat(mean[ids], dw) + rng.normal(0.0, noise, int(starts[-1])), -5.0, 5.0)
    raw = raw_scale * x + raw_shift
    # anchors: matched, thinned, with a jittered (but monotone) base->sample map
    matched = np.nonzero(~subs)[0]
    cand = matched[(matched >= trim) & (matched < L - trim)]
    keep = cand[rng.random(cand.size) < anchor_density]
    pos = np.clip(starts[keep] + rng.integers(-jitter, jitter + 1, keep.size), 0, int(starts[-1]) - 1)
    pos = np.maximum.accumulate(pos)
    seq_to_sig = {int(b): int(p) for b, p in zip(keep, pos)}
    # base mapping rows (read index, reference index in the read's orientation), alignment.py:128-134
    ref_idx = (G - 1 - (g0 + L - 1 - matched)) if reverse else (g0 + matched)
    base_mapping = np.stack([matched, ref_idx], axis=1).astype(int)
    return dict(raw_signal=raw, sequence=_BASES[seq], sequence_to_signal_mapping=seq_to_sig,
                base_mapping=base_mapping, reverse=reverse, g0=g0, length=L, true_starts=starts)


def make_read_specs(n, genome, model, seed=0, **kw):
    genome_num = np.array([{'A': 0, 'C': 1, 'G': 2, 'T': 3}[b] for b in genome])
    return [make_read_spec(np.random.default_rng([seed, i]), genome_num, model, i, **kw) for i in range(n)]


def reads_from_specs(specs, read_class=None):
    """Instantiate ``Read`` objects (this package's class by default) from plain specs."""
    if read_class is None:
        from .read import Read as read_class
    out = []
    for s in specs:
        r = read_class()
        r.raw_signal = np.array(s['raw_signal'])
        r.sequence = np.array(s['sequence'])
        r.sequence_to_signal_mapping = dict(s['sequence_to_signal_mapping'])
        r._spec = s
        out.append(r)
    return out


def make_synthetic_aligner(base_class, reference):
    """Subclass ``base_class`` (an ApproximateAligner) so that ``_get_base_alignment`` returns the
    simulated truth attached to each read; everything downstream (convert_mapping,
    get_signal_alignment) is the base class's own code."""

    class SyntheticAligner(base_class):
        def __init__(self, reference):
            self.reference = reference
            self.references_dict = None
            self.reference_filename = None
            self.bwa_executable = None
            self.bwapy_aligner = None

        def _get_base_alignment(self, read):
            s = read._spec
            if len(s['base_mapping']) == 0:
                return None
            return np.array(s['base_mapping'], dtype=int), s['reverse'], 'synthetic'

    return SyntheticAligner(reference)


def make_read_batch(n, model, seed=0, genome_length=10000, raw_dtype=np.int16, **kw):
    """``n`` simulated reads against one random genome as a struct-of-arrays ``readbatch.ReadBatch`` plus the
    batch aligner that knows their true base mapping: -> (ReadBatch, SyntheticBatchAligner, genome codes).
    Raw signals are ADC-like counts (``raw_dtype`` int16, as fast5 files hold them)."""
    from .readbatch import ReadBatch, BaseAlignmentBatch, SyntheticBatchAligner
    genome = np.random.default_rng(seed).integers(0, 4, genome_length).astype(np.int32)
    specs = [make_read_spec(np.random.default_rng([seed, i]), genome, model, i, **kw) for i in range(n)]
    inv = {'A': 0, 'C': 1, 'G': 2, 'T': 3}
    off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
    raws = [np.rint(s['raw_signal']).astype(raw_dtype) if np.issubdtype(raw_dtype, np.integer)
            else np.asarray(s['raw_signal'], dtype=raw_dtype) for s in specs]
    seqs = [np.array([inv[b] for b in s['sequence']], dtype=np.int32) for s in specs]
    maps = [sorted(s['sequence_to_signal_mapping'].items()) for s in specs]
    rb = ReadBatch(np.concatenate(raws), off(raws), np.concatenate(seqs), off(seqs),
                   np.array([k for m in maps for k, _ in m], dtype=np.int64),
                   np.array([v for m in maps for _, v in m], dtype=np.int64), off(maps))
    bms = [np.asarray(s['base_mapping'], dtype=np.int64).reshape(-1, 2) for s in specs]
    ba = BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]),
                            off(bms), np.array([s['reverse'] for s in specs], dtype=bool))
    return rb, SyntheticBatchAligner(genome, ba), genome
