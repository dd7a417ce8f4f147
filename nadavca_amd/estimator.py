"""ProbabilityEstimator / Chunk (mirrors /root/reference/nadavca/estimator.py:7-236).

Same public methods and results as the reference; the difference is the execution plan: the
reference loops over reads and calls the DP once per read, here every stage is one batched GPU
launch over all reads —
    refine_alignment (tweak pre-pass)  -> expected levels -> [host: spline tweak, scipy]
    -> estimate_log_likelihoods -> normalise / strand-flip / per-position sum (consensus kernel)
    -> windowed posterior kernel.
Nothing numerical runs on the CPU except the spline FIT of ``Read.tweak_signal_normalization``
(FITPACK ``splrep``, a host step of the reference adjacent to the path, the same scipy call); its
evaluation over the signals is a kernel that restates FITPACK's ``splev`` bit for bit.
"""
import ctypes as C

import numpy

from . import _lib, dtw
from .alphabet import alphabet
from .genome import Genome


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class Chunk:
    def __init__(self, start, end, values, coverage=None):
        self.start = start
        self.end = end
        self.values = values
        self.coverage = coverage
        if coverage is None:
            self.coverage = numpy.ones(end - start, dtype=int)

    def __lt__(self, other):
        if self.start == other.start:
            return self.end < other.end
        return self.start < other.start

    @staticmethod
    def print_head(file):
        file.write('index\tbase\tcoverage\t{}\n'.format('\t'.join(alphabet)))

    def print(self, file, reference):
        for i in range(self.start, self.end):
            values_string = '\t'.join(map('{:18.16f}'.format, self.values[i - self.start]))
            file.write('{}\t{}\t{}\t{}\n'.format(i, reference[i], self.coverage[i - self.start],
                                                 values_string))


class _Prepared:
    """Per-read inputs of the DP, sliced exactly as estimator.py:59-74 / 158-170 do."""
    __slots__ = ('read', 'apx', 'reference_part', 'signal_range', 'context_before', 'context_after')


class ProbabilityEstimator:
    def __init__(self, kmer_model, aligner, config):
        self.kmer_model = kmer_model
        self.aligner = aligner
        self.bandwidth = config['bandwidth']
        self.snp_prior = config['snp_prior_probability']
        self.min_event_length = config['min_event_length']
        self.model_wobbling = config['model_wobbling']
        self.model_transitions = config['model_transitions']
        self.normalization_event_length = config['normalization_event_length']
        self.tweak_signal_normalization = config['tweak_signal_normalization']

    # ---- slicing (host) ------------------------------------------------------------------------
    def _get_read_context(self, read, read_sequence_range):
        start, end = read_sequence_range
        k = self.kmer_model.get_k()
        central = self.kmer_model.get_central_position()
        # a negative slice start wraps, as in the reference (estimator.py:53)
        before = Genome.to_numerical(read.sequence[start - central: start])
        after = Genome.to_numerical(read.sequence[end: end + k - central - 1])
        return before, after

    def _prepare(self, read, reference=None):
        apx = self.aligner.get_signal_alignment(read, self.bandwidth)
        if apx is None:
            return None
        p = _Prepared()
        p.read, p.apx = read, apx
        if reference is None:
            part = apx.reference_part
        else:  # estimator.py:64-68 re-slices the caller's reference
            s, e = apx.reference_range
            part = reference[s:e]
            if apx.reverse_complement:
                part = Genome.reverse_complement(part)
        p.reference_part = Genome.to_numerical(part)
        p.signal_range = apx.signal_range
        p.context_before, p.context_after = self._get_read_context(read, apx.read_sequence_range)
        return p

    @staticmethod
    def _dp_tuple(p, signal):
        s, e = p.signal_range
        return (signal[s:e], p.reference_part, p.context_before, p.context_after, p.apx.alignment)

    # ---- alignment -------------------------------------------------------------------------------
    def get_refined_alignments(self, reads):
        """Batched ``get_refined_alignment``: list of (approximate_alignment, (R,3) int array) or
        None per read (estimator.py:158-196)."""
        prepared = [self._prepare(r) for r in reads]
        live = [p for p in prepared if p is not None]
        events = dtw.refine_alignment_batch(
            [self._dp_tuple(p, p.read.normalized_signal) for p in live], self.bandwidth,
            self.min_event_length, self.kmer_model, self.model_transitions) if live else []
        out, it = [], iter(events)
        for p in prepared:
            if p is None:
                out.append(None)
                continue
            ev = next(it)
            if len(ev) == 0:  # no valid path in the band
                out.append(None)
                continue
            s0 = p.signal_range[0]
            start_ref, end_ref = p.apx.reference_range
            res = numpy.zeros((len(ev), 3), dtype=int)
            pos = numpy.arange(len(ev))
            res[:, 0] = (end_ref - pos - 1) if p.apx.reverse_complement else (start_ref + pos)
            res[:, 1] = ev[:, 0] + s0
            res[:, 2] = ev[:, 1] + s0
            out.append((p.apx, res))
        return out

    def get_refined_alignment(self, read):
        return self.get_refined_alignments([read])[0]

    def _alignment_rows(self, p, ev):
        s0 = p.signal_range[0]
        start_ref, end_ref = p.apx.reference_range
        res = numpy.zeros((len(ev), 3), dtype=int)
        pos = numpy.arange(len(ev))
        res[:, 0] = (end_ref - pos - 1) if p.apx.reverse_complement else (start_ref + pos)
        res[:, 1] = ev[:, 0] + s0
        res[:, 2] = ev[:, 1] + s0
        return res

    def refine_and_renormalize(self, reads, renorm_rounds):
        """``align_signal``'s per-read loop (align_signal.py:55-80) for all reads at once and without
        leaving the device between rounds: align, then alternately re-fit the normalisation linearly
        against the model's expected levels (even rounds) and align again (odd rounds) —
        ``device.refine_renorm_loop_dev``.  Every read's ``normalized_signal`` ends up rescaled as in
        the reference.  -> list of (approximate_alignment, (R,3) int array) or None per read."""
        import torch
        from .device import DeviceBatch, refine_renorm_loop_dev
        prepared = [self._prepare(r) for r in reads]
        live = [p for p in prepared if p is not None]
        if not live:
            return [None] * len(prepared)
        batch = dtw.FlatBatch([self._dp_tuple(p, p.read.normalized_signal) for p in live])
        dbatch = DeviceBatch(batch, torch.device('cuda', self.kmer_model.context.device))
        events, status, fits = refine_renorm_loop_dev(dbatch, self.bandwidth, self.min_event_length,
                                                      self.kmer_model, self.model_transitions, renorm_rounds)
        events, status = events.cpu().numpy(), status.cpu().numpy()
        fits = [f.cpu().numpy() for f in fits]
        if (status < 0).any():
            bad = numpy.nonzero(status < 0)[0]
            raise ValueError('refine_alignment: invalid input for read(s) %s' % bad[:8].tolist())
        out, j = [], 0
        for p in prepared:
            if p is None:
                out.append(None)
                continue
            if status[j] != 0:
                out.append(None)
            else:
                # the same two linear maps for the samples outside the aligned slice (the reference
                # rescales the whole read, align_signal.py:73)
                for f in fits:
                    p.read.normalized_signal = (p.read.normalized_signal - f[j, 1]) / f[j, 0]
                out.append((p.apx, self._alignment_rows(p, events[batch.ref_off[j]:batch.ref_off[j + 1]])))
            j += 1
        return out

    # ---- SNP scoring -----------------------------------------------------------------------------
    def _log_likelihood_batch(self, reference, reads):
        """Stages shared by both modes: -> (live prepared reads, FlatBatch, ll (sum R, 4), status)."""
        prepared = [self._prepare(r, reference) for r in reads]
        live = [p for p in prepared if p is not None]
        if not live:
            return [], None, None, None
        if self.tweak_signal_normalization:
            pre = dtw.refine_alignment_batch(
                [self._dp_tuple(p, p.read.normalized_signal) for p in live], self.bandwidth,
                self.min_event_length, self.kmer_model, False)
            expected = self.kmer_model.get_expected_signal_batch(
                [(p.reference_part, p.context_before, p.context_after) for p in live])
            fitted, splines = [], []
            for p, ev, exp in zip(live, pre, expected):
                if len(ev) == 0:
                    # the reference would index an empty array here and fail; keep the untweaked signal
                    p.read.tweaked_normalized_signal = p.read.normalized_signal
                    continue
                # the fit on the host (FITPACK splrep, the reference's call), the evaluation over the
                # whole signal for all reads in one kernel
                splines.append(p.read.fit_signal_tweak(numpy.asarray(ev) + p.signal_range[0], exp))
                fitted.append(p.read)
            from .read import Read
            Read.apply_signal_tweaks_device(fitted, splines, context=self.kmer_model.context)
            signals = [p.read.tweaked_normalized_signal for p in live]
        else:
            signals = [p.read.normalized_signal for p in live]
        batch = dtw.FlatBatch([self._dp_tuple(p, s) for p, s in zip(live, signals)])
        ll, status = dtw.estimate_log_likelihoods_flat(batch, self.bandwidth, self.min_event_length,
                                                       self.kmer_model, self.model_wobbling)
        return live, batch, ll, status

    def _accumulate(self, live, batch, ll, status, chunk_start, length):
        lib = _lib.load()
        alpha = self.kmer_model.alphabet_size
        acc = numpy.zeros((length, alpha), dtype=numpy.float64)
        cov = numpy.zeros(length, dtype=numpy.int64)
        reverse = numpy.array([1 if p.apx.reverse_complement else 0 for p in live], dtype=numpy.int32)
        chunk_start = numpy.ascontiguousarray(chunk_start, dtype=numpy.int64)
        status = numpy.ascontiguousarray(status, dtype=numpy.int32)
        _lib.check(lib.nvk_consensus_accumulate(
            self.kmer_model.context.handle, len(live), alpha, _ptr(ll), _ptr(batch.reference),
            _ptr(batch.ref_off), _ptr(chunk_start), _ptr(reverse), _ptr(status),
            float(self.normalization_event_length), length, _ptr(acc), _ptr(cov)),
            'nvk_consensus_accumulate')
        return acc, cov

    def _posterior(self, ll, reference_num, seg_off):
        lib = _lib.load()
        alpha = self.kmer_model.alphabet_size
        ll = numpy.ascontiguousarray(ll, dtype=numpy.float64)
        ref = numpy.ascontiguousarray(reference_num, dtype=numpy.int32)
        seg = numpy.ascontiguousarray(seg_off, dtype=numpy.int64)
        out = numpy.zeros_like(ll)
        _lib.check(lib.nvk_posterior(self.kmer_model.context.handle, ll.shape[0], seg.size - 1, _ptr(seg),
                                     alpha, self.kmer_model.get_k(), float(self.snp_prior), _ptr(ll),
                                     _ptr(ref), _ptr(out)), 'nvk_posterior')
        return out

    @staticmethod
    def group_ranges(ranges):
        """Merge sorted (start, end) chunk intervals into groups while next.start < current_end;
        touching chunks do not merge (estimator.py:205-220)."""
        ranges = sorted(ranges)
        groups, cur_s, cur_e = [], None, None
        for idx, (s, e) in enumerate(ranges):
            if cur_s is None:
                cur_s, cur_e = s, e
            cur_e = max(cur_e, e)
            if idx + 1 >= len(ranges) or ranges[idx + 1][0] >= cur_e:
                groups.append((cur_s, cur_e))
                cur_s = cur_e = None
        return groups

    def local_consensus(self, reference, reads):
        """This process's share of the consensus: per-position sums of the normalised,
        strand-corrected log-likelihoods of ``reads`` over the whole reference, the coverage, and
        the chunk intervals (estimator.py:199-231).  -> (acc (L,4) f64, cov (L,) i64, ranges)."""
        alpha = self.kmer_model.alphabet_size
        live, batch, ll, status = self._log_likelihood_batch(reference, reads)
        keep = [j for j, p in enumerate(live) if status[j] == dtw.READ_OK]
        if not keep:
            return (numpy.zeros((len(reference), alpha)), numpy.zeros(len(reference), dtype=numpy.int64), [])
        chunk_start = [p.apx.reference_range[0] for p in live]
        acc, cov = self._accumulate(live, batch, ll, status, chunk_start, len(reference))
        return acc, cov, [tuple(int(v) for v in live[j].apx.reference_range) for j in keep]

    def posterior_of_groups(self, reference, cov, groups, seg_off, ll_cat):
        """Posterior of already grouped sums (all groups in one launch, laid end to end) -> Chunk list."""
        if not groups:
            return []
        ref_cat = numpy.concatenate([Genome.to_numerical(reference[s:e]) for s, e in groups])
        post = self._posterior(ll_cat, ref_cat, seg_off)
        return [Chunk(s, e, post[seg_off[g]:seg_off[g + 1]], cov[s:e].copy())
                for g, (s, e) in enumerate(groups)]

    def posterior_groups(self, reference, acc, cov, ranges):
        """Group the chunk intervals and turn the summed log-likelihoods into posteriors
        (estimator.py:205-235)."""
        groups = self.group_ranges(ranges)
        if not groups:
            return []
        seg_off = numpy.zeros(len(groups) + 1, dtype=numpy.int64)
        numpy.cumsum([e - s for s, e in groups], out=seg_off[1:])
        ll_cat = numpy.concatenate([acc[s:e] for s, e in groups])
        return self.posterior_of_groups(reference, cov, groups, seg_off, ll_cat)

    def estimate_probabilities(self, reference, reads):
        """Consensus over all reads (estimator.py:199-236) -> list of Chunk(start, end, posterior,
        coverage), one per group of overlapping reads."""
        acc, cov, ranges = self.local_consensus(reference, reads)
        return self.posterior_groups(reference, acc, cov, ranges)

    def estimate_probabilities_independent(self, reference, reads):
        """``[estimate_probabilities(reference, [read])[0] for read in reads]`` in batched form
        (estimate_snps.py:63-68); None for a read that yields no chunk."""
        live, batch, ll, status = self._log_likelihood_batch(reference, reads)
        result = {}
        if live:
            alpha = self.kmer_model.alphabet_size
            total = int(batch.ref_off[-1])
            acc, _ = self._accumulate(live, batch, ll, status, batch.ref_off[:-1], total)
            ref_cat = numpy.zeros(total, dtype=numpy.int32)
            for j, p in enumerate(live):
                s, e = p.apx.reference_range
                ref_cat[batch.ref_off[j]:batch.ref_off[j + 1]] = Genome.to_numerical(reference[s:e])
            post = self._posterior(acc, ref_cat, batch.ref_off)
            for j, p in enumerate(live):
                if status[j] != dtw.READ_OK:
                    continue
                s, e = p.apx.reference_range
                result[id(p.read)] = Chunk(s, e, post[batch.ref_off[j]:batch.ref_off[j + 1]].reshape(-1, alpha))
        return [result.get(id(r)) for r in reads]
