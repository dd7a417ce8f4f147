"""Struct-of-arrays form of a set of reads and of the approximate-alignment stage, for the batched workflows.

The reference walks a Python list of ``Read`` objects and builds, per read, a dictionary look-up table, an
anchor list and a handful of slices (/root/reference/nadavca/alignment.py:54-60,142-186,
/root/reference/nadavca/estimator.py:49-74,158-170).  With the dynamic programming on the GPU that per-read
Python is all that is left of the run time, so the same arithmetic is done here once for all
reads with tensor operations on flat arrays (on the device when given one):

* ``ReadBatch``            raw signals, basecalled sequences and base -> sample tables laid end to end;
* ``BaseAlignmentBatch``   what ``ApproximateAligner._get_base_alignment`` returns per read — matched (read
                           base, oriented reference base) pairs — for all reads;
* ``signal_alignments()``  ``convert_mapping`` + ``get_signal_alignment`` + ``_get_read_context`` for all reads:
                           anchors, signal windows, reference parts and k-mer contexts in the flat layout of
                           the C-ABI (include/nadavca_hip.h).

Nothing here computes the alignment itself; the arrays go to the device and through the kernels
(nadavca_amd/align_signal.py: ``align_signal_batch``)."""
import numpy as np

def _offsets(lengths):
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=off[1:])
    return off


class ReadBatch:
    """n reads end to end.  raw_signal: any real dtype (fast5 holds int16 ADC counts), read j at
    [sig_off[j], sig_off[j+1]); sequence: base codes 0..3, read j at [seq_off[j], seq_off[j+1]); the
    basecaller's base -> sample table (``Read.sequence_to_signal_mapping``) as parallel arrays map_base /
    map_sig, read j at [map_off[j], map_off[j+1]), bases ascending inside a read.  The two tables are kept as int32
    (positions inside ONE read): they are 2 x 16 MB per 10 000 reads to carry across PCIe instead of 2 x 32."""

    def __init__(self, raw_signal, sig_off, sequence, seq_off, map_base, map_sig, map_off):
        self.raw_signal = np.ascontiguousarray(raw_signal)
        self.sig_off = np.ascontiguousarray(sig_off, dtype=np.int64)
        self.sequence = np.ascontiguousarray(sequence, dtype=np.int32)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self.map_base = np.ascontiguousarray(map_base, dtype=np.int32)
        self.map_sig = np.ascontiguousarray(map_sig, dtype=np.int32)
        self.map_off = np.ascontiguousarray(map_off, dtype=np.int64)
        self.n = self.sig_off.size - 1
        self.normalized = None   # device tensor (f64, layout of raw_signal) once normalised

    @classmethod
    def from_reads(cls, reads):
        """From ``Read`` objects (per-read Python: the compatible way in, not the fast one)."""
        from .genome import Genome
        raws = [np.asarray(r.raw_signal) for r in reads]
        seqs = [Genome.to_numerical(r.sequence).astype(np.int32) for r in reads]
        keys = [np.fromiter(r.sequence_to_signal_mapping.keys(), dtype=np.int64,
                            count=len(r.sequence_to_signal_mapping)) for r in reads]
        vals = [np.fromiter(r.sequence_to_signal_mapping.values(), dtype=np.int64,
                            count=len(r.sequence_to_signal_mapping)) for r in reads]
        order = [np.argsort(k, kind='stable') for k in keys]
        cat = lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dtype=dt)
        dt = np.result_type(*[x.dtype for x in raws]) if raws else np.float64
        return cls(cat(raws, dt), _offsets([x.size for x in raws]), cat(seqs, np.int32),
                   _offsets([x.size for x in seqs]), cat([k[o] for k, o in zip(keys, order)], np.int64),
                   cat([v[o] for v, o in zip(vals, order)], np.int64), _offsets([k.size for k in keys]))


class BaseAlignmentBatch:
    """Matched bases of all reads: pair p of read j (pairs [off[j], off[j+1])) says read base
    ``read_idx[p]`` sits on reference base ``ref_idx[p]``, both in the READ's orientation (for a
    reverse-complement hit the reference index counts from the reference's end, alignment.py:128-134).
    ``reverse[j]``: the read is on the reverse strand; a read without pairs did not align."""

    def __init__(self, read_idx, ref_idx, off, reverse):
        self.read_idx = np.ascontiguousarray(read_idx, dtype=np.int32)   # a position inside one read
        self.ref_idx = np.ascontiguousarray(ref_idx, dtype=np.int64)
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        self.reverse = np.ascontiguousarray(reverse, dtype=bool)


class SignalAlignmentBatch:
    """The per-read results of ``get_signal_alignment`` (alignment.py:142-186) and ``_get_read_context``
    (estimator.py:49-57) for the ``live`` reads (those with at least one anchor), flat; torch tensors on the
    device the stage ran on (``host()`` -> the same with numpy arrays)."""
    FIELDS = ('live', 'anchors', 'anc_off', 'win_start', 'win_len', 'slice_start', 'ref_start', 'ref_end',
              'reverse', 'read_seq_start', 'read_seq_end', 'reference', 'ref_off', 'context_before',
              'cb_off', 'context_after', 'ca_off')
    __slots__ = FIELDS

    def host(self, fields=None):
        out = SignalAlignmentBatch()
        for f in SignalAlignmentBatch.FIELDS:
            v = getattr(self, f)
            setattr(out, f, v.cpu().numpy() if (fields is None or f in fields) and hasattr(v, 'cpu') else v)
        return out


def signal_alignments(rb, ba, bandwidth, reference_num, k, central, device='cpu'):
    """-> SignalAlignmentBatch.  ``reference_num``: the reference as base codes.  Per read, exactly the
    arithmetic of the reference: anchors = matched bases the basecaller placed on the signal; reference range
    from the first and last anchor (reported on the forward strand); signal window = anchors' sample span
    +- bandwidth, clipped to the read; reference part = that range, reverse-complemented for a reverse-strand
    read; contexts = the read's own bases around the aligned part.
    Written with torch tensor operations so that it runs where the data is: on the GPU (``device`` = the
    context's cuda device: a millisecond for 10 000 reads, and its outputs are already where the kernels
    want them) or on the CPU (tests).  Index plumbing only — gathers, prefix sums, comparisons."""
    import torch
    dev = torch.device(device)
    # (each table crosses to the device in the dtype it is kept in and is widened there)
    T = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dev).to(dt)
    i64 = torch.int64
    ref_num = T(reference_num, torch.int32)
    L = int(ref_num.numel())
    n = rb.n
    seq_off, sig_off_r, map_off, ba_off = T(rb.seq_off, i64), T(rb.sig_off, i64), T(rb.map_off, i64), T(ba.off, i64)
    sequence = T(rb.sequence, torch.int32)
    map_base, map_sig = T(rb.map_base, i64), T(rb.map_sig, i64)
    read_idx, ref_idx = T(ba.read_idx, i64), T(ba.ref_idx, i64)
    reverse_all = T(ba.reverse, torch.bool)

    def offsets(lengths):
        return torch.cat([torch.zeros(1, dtype=i64, device=dev), torch.cumsum(lengths, 0)])

    def seg_index(off):  # owner segment and position inside it, for flat positions 0..off[-1]
        lens = off[1:] - off[:-1]
        total = int(off[-1])
        owner = torch.repeat_interleave(torch.arange(lens.numel(), dtype=i64, device=dev), lens, output_size=total)
        inner = torch.arange(total, dtype=i64, device=dev) - off[:-1][owner]
        return owner, inner

    seq_len_all = seq_off[1:] - seq_off[:-1]
    # base -> sample look-up for all reads at once: a dense table over the concatenated sequences
    sig_of_base = torch.full((int(seq_off[-1]) + 1,), -1, dtype=i64, device=dev)
    m_owner, _ = seg_index(map_off)
    ok = (map_base >= 0) & (map_base < seq_len_all[m_owner])
    sig_of_base[(seq_off[:-1][m_owner] + map_base)[ok]] = map_sig[ok]
    p_owner, _ = seg_index(ba_off)
    inside = (read_idx >= 0) & (read_idx < seq_len_all[p_owner])
    g = torch.where(inside, seq_off[:-1][p_owner] + read_idx, seq_off[-1])
    sig = sig_of_base[g]
    keep = sig >= 0
    k_owner, k_sig, k_ref = p_owner[keep], sig[keep], ref_idx[keep]
    cnt = torch.bincount(k_owner, minlength=n)
    live = torch.nonzero(cnt > 0).reshape(-1)
    a_off_all = offsets(cnt)
    first, last = a_off_all[:-1][live], a_off_all[1:][live] - 1
    rev = reverse_all[live]

    out = SignalAlignmentBatch()
    out.live = live
    start_ref_o, end_ref_o = k_ref[first], k_ref[last] + 1          # oriented coordinates
    out.ref_start = torch.where(rev, L - end_ref_o, start_ref_o)
    out.ref_end = torch.where(rev, L - start_ref_o, end_ref_o)
    out.reverse = rev
    start_sig, end_sig = k_sig[first], k_sig[last] + 1
    sig_len = (sig_off_r[1:] - sig_off_r[:-1])[live]
    out.slice_start = torch.clamp(start_sig - bandwidth, min=0)
    slice_end = torch.minimum(sig_len, end_sig + bandwidth)
    out.win_start = sig_off_r[:-1][live] + out.slice_start
    out.win_len = slice_end - out.slice_start
    # first / last matched read base over ALL pairs of the read (alignment.py:183)
    out.read_seq_start = read_idx[ba_off[:-1][live]]
    out.read_seq_end = read_idx[ba_off[1:][live] - 1] + 1

    # anchors of the live reads, rebased to the window and to the reference part
    live_of = torch.full((n,), -1, dtype=i64, device=dev)
    live_of[live] = torch.arange(live.numel(), dtype=i64, device=dev)
    lo = live_of[k_owner]                    # every kept pair belongs to a live read
    out.anchors = torch.stack([k_sig - out.slice_start[lo], k_ref - start_ref_o[lo]], dim=1).to(torch.int32)
    out.anc_off = offsets(cnt[live])

    # reference parts: forward range, reverse-complemented for reverse-strand reads
    out.ref_off = offsets(out.ref_end - out.ref_start)
    r_owner, r_inner = seg_index(out.ref_off)
    r_rev = rev[r_owner]
    pos = torch.where(r_rev, out.ref_end[r_owner] - 1 - r_inner, out.ref_start[r_owner] + r_inner)
    bases = ref_num[pos]
    out.reference = torch.where(r_rev, 3 - bases, bases).to(torch.int32)

    # k-mer contexts from the read's own sequence: seq[start - central : start] and
    # seq[end : end + k - central - 1], with Python's slice rules (a negative start wraps)
    s_len, s_base = seq_len_all[live], seq_off[:-1][live]
    b_lo = out.read_seq_start - central
    b_lo = torch.where(b_lo < 0, torch.clamp(b_lo + s_len, min=0), b_lo)
    b_hi = torch.minimum(out.read_seq_start, s_len)
    out.cb_off = offsets(torch.clamp(b_hi - b_lo, min=0))
    o, inner = seg_index(out.cb_off)
    out.context_before = sequence[s_base[o] + b_lo[o] + inner]
    a_lo = torch.minimum(out.read_seq_end, s_len)
    a_hi = torch.minimum(out.read_seq_end + (k - central - 1), s_len)
    out.ca_off = offsets(torch.clamp(a_hi - a_lo, min=0))
    o, inner = seg_index(out.ca_off)
    out.context_after = sequence[s_base[o] + a_lo[o] + inner]
    return out


class SyntheticBatchAligner:
    """Stands in for BWA on simulated reads: returns the true base mapping of every read
    (synthetic.make_read_batch), in the contract of ``BaseAlignmentBatch``."""

    def __init__(self, reference_num, base_alignments):
        self.reference_num = np.asarray(reference_num, dtype=np.int32)
        self._ba = base_alignments

    def get_base_alignments(self, read_batch):
        return self._ba
