"""Nanopore read container (mirrors /root/reference/nadavca/read.py:9-94).

``normalize_reads`` (global median/MAD + clip) and ``tweak_signal_normalization`` (spline through
(event mean, expected level) pairs) are host steps adjacent to the GPU path; they use the same
numpy/scipy calls as the reference so their results are identical.  ``normalize_reads_device`` is the
same normalisation as a device kernel (what the workflows call).  fast5 loading needs h5py and is
imported lazily (no fast5 data exists offline)."""
import numpy
from scipy import interpolate

from .genome import Genome


class Read:
    def __init__(self):
        self.raw_signal = None
        self.normalized_signal = None
        self.tweaked_normalized_signal = None
        self.strand = None
        self.fastq = None
        self.sequence_to_signal_mapping = None
        self.sequence = None

    @staticmethod
    def _extract_sequence_to_signal_mapping(events):
        # Events table: a base is emitted whenever move > 0 (read.py:19-27)
        mapping, index = {}, 1
        for event in events:
            index += event['move']
            if event['move'] > 0:
                mapping[index] = event['start']
        return mapping

    @staticmethod
    def _extract_sequence_to_signal_mapping_from_moves(moves, signal_start, signal_step):
        # Move table: one row per stride of signal_step samples (read.py:29-41)
        mapping, base_pos, signal_pos = {}, 0, signal_start
        for row in moves:
            if row == 1:
                mapping[base_pos] = int(signal_pos)
            base_pos += row
            signal_pos += signal_step
        return mapping

    @staticmethod
    def load_from_fast5(filename, basecall_group, segmentation_group='Analyses/Segmentation_000'):
        import h5py  # optional dependency
        read = Read()
        with h5py.File(filename, 'r') as file:
            read_group = list(file['Raw/Reads'].values())[0]
            read.raw_signal = numpy.array(read_group['Signal'][()])
            template = '{}/BaseCalled_template'.format(basecall_group)
            if template + '/Events' in file:
                read.sequence_to_signal_mapping = Read._extract_sequence_to_signal_mapping(
                    file[template + '/Events'])
            elif template + '/Move' in file:
                start = file['{}/Summary/segmentation'.format(segmentation_group)].attrs['first_sample_template']
                step = file['{}/Summary/basecall_1d_template'.format(basecall_group)].attrs['block_stride']
                read.sequence_to_signal_mapping = Read._extract_sequence_to_signal_mapping_from_moves(
                    file[template + '/Move'], start, step)
            else:
                raise KeyError('Cannot find Events or Move table from basecaller.')
            read.fastq = file[template + '/Fastq'][()].decode('ascii')
            read.sequence = Genome.create_from_fastq_string(read.fastq)[0].bases
        return read

    @staticmethod
    def normalize_reads(reads):
        """Robust z-score with ONE centre and scale for the whole list: centre = median of every
        raw sample of every read, scale = median absolute deviation from it; the result is clipped
        to [-5, 5] and stored in ``normalized_signal`` (reference behaviour: read.py:67-81, where the
        medians come from ``statistics.median`` — numerically the same as ``numpy.median``)."""
        if not reads:
            return
        pooled = numpy.concatenate([numpy.asarray(r.raw_signal, dtype=float) for r in reads])
        centre = numpy.median(pooled)
        spread = numpy.median(abs(pooled - centre))
        for r in reads:
            r.normalized_signal = numpy.clip((r.raw_signal - centre) / spread, -5, 5)

    @staticmethod
    def normalize_reads_device(reads, per_read=False, context=None):
        """``normalize_reads`` on the GPU (``nvk_normalize_groups_dev``: exact radix selection for the
        two medians, then the clip): one centre and scale for the whole list, or, with ``per_read``,
        one per read — what ``align_signal`` does by calling ``normalize_reads([read])`` per read
        (align_signal.py:54).  Results are identical to ``normalize_reads``."""
        if not reads:
            return
        import torch
        from . import _lib
        from .device import normalize_groups_dev
        context = context or _lib.default_context()
        device = torch.device('cuda', context.device)
        raws = [numpy.ascontiguousarray(r.raw_signal, dtype=float) for r in reads]
        bounds = numpy.zeros(len(raws) + 1, dtype=numpy.int64)
        numpy.cumsum([x.size for x in raws], out=bounds[1:])
        groups = bounds if per_read else bounds[[0, -1]]
        raw = torch.from_numpy(numpy.concatenate(raws)).to(device)
        out, _ = normalize_groups_dev(context, raw, torch.from_numpy(numpy.ascontiguousarray(groups)).to(device))
        host = out.cpu().numpy()
        for j, r in enumerate(reads):
            r.normalized_signal = host[bounds[j]:bounds[j + 1]].copy()

    def fit_signal_tweak(self, alignment, expected_means):
        """The fit of ``tweak_signal_normalization`` -> FITPACK spline ``(t, c, k)``.
        Re-normalise the read against the pore model (reference behaviour: read.py:83-94).
        For every aligned event (absolute sample range per row of ``alignment``) take the mean of
        ``normalized_signal``; keep the events whose mean is within 1 of the model's expected level;
        fit a smoothing spline (FITPACK ``splrep`` with s = number of points) from observed mean to
        expected level; the caller applies it to the whole signal -> ``tweaked_normalized_signal``."""
        signal = self.normalized_signal
        events = numpy.asarray(alignment)[:, :2].astype(numpy.intp).reshape(-1, 2)
        levels = numpy.asarray(expected_means, dtype=float)[:len(events)]
        events = events[:len(levels)]
        first, last = events[:, 0], events[:, 1]
        count = last - first
        # all event sums in one call: np.add.reduceat over [first_0, last_0, first_1, last_1, ...]
        # (every other result is the sum of signal[first:last]); one pad sample makes last == len
        # addressable.  The summation order differs from numpy.mean's pairwise scheme by <= 1 ulp.
        cuts = numpy.empty(2 * len(first), dtype=numpy.intp)
        cuts[0::2], cuts[1::2] = first, numpy.maximum(last, first)
        padded = numpy.append(numpy.asarray(signal, dtype=float), 0.0)
        sums = numpy.add.reduceat(padded, cuts)[0::2] if len(first) else numpy.zeros(0)
        with numpy.errstate(invalid='ignore', divide='ignore'):
            observed = numpy.where(count > 0, sums / numpy.maximum(count, 1), numpy.nan)
            keep = numpy.abs(levels - observed) <= 1  # an empty event has mean NaN and drops out
        xs, ys = observed[keep], levels[keep]
        order = numpy.lexsort((ys, xs))
        xs, ys = xs[order], ys[order]
        return interpolate.splrep(xs, ys, s=len(xs))

    def tweak_signal_normalization(self, alignment, expected_means):
        """``fit_signal_tweak`` + the spline's evaluation over the whole signal (read.py:94), both on the
        host with the reference's scipy calls; the workflows evaluate on the GPU instead
        (``apply_signal_tweaks_device``)."""
        knots = self.fit_signal_tweak(alignment, expected_means)
        self.tweaked_normalized_signal = interpolate.splev(self.normalized_signal, knots)

    @staticmethod
    def apply_signal_tweaks_device(reads, splines, context=None):
        """``tweaked_normalized_signal = splev(normalized_signal, spline)`` for many reads in one kernel
        (``nvk_splev_groups_dev``: FITPACK's evaluation restated operation for operation, results equal
        scipy's).  ``splines``: one ``(t, c, k)`` per read, as ``fit_signal_tweak`` returns them."""
        if not reads:
            return
        import torch
        from . import _lib
        from .device import splev_groups_dev
        context = context or _lib.default_context()
        device = torch.device('cuda', context.device)
        sigs = [numpy.ascontiguousarray(r.normalized_signal, dtype=float) for r in reads]
        bounds = numpy.zeros(len(sigs) + 1, dtype=numpy.int64)
        numpy.cumsum([x.size for x in sigs], out=bounds[1:])
        degree = int(splines[0][2])
        if any(int(sp[2]) != degree for sp in splines):
            raise ValueError('splines of different degrees in one batch')
        kb = numpy.zeros(len(sigs) + 1, dtype=numpy.int64)
        numpy.cumsum([len(sp[0]) for sp in splines], out=kb[1:])
        t = numpy.concatenate([numpy.asarray(sp[0], dtype=float) for sp in splines])
        c = numpy.concatenate([numpy.asarray(sp[1], dtype=float)[:len(sp[0])] for sp in splines])
        up = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(device)
        out = splev_groups_dev(context, up(numpy.concatenate(sigs)), up(bounds), up(t), up(c), up(kb), degree)
        host = out.cpu().numpy()
        for j, r in enumerate(reads):
            r.tweaked_normalized_signal = host[bounds[j]:bounds[j + 1]].copy()
