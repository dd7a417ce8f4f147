"""Approximate (anchor) alignment stage (mirrors /root/reference/nadavca/alignment.py:10-186).

Only the contract matters to the GPU path: ``get_signal_alignment(read, bandwidth)`` returns an
``ApproximateSignalAlignment`` whose ``alignment`` rows are (signal index inside the slice,
reference index inside the part) anchors, with the slice/range arithmetic of alignment.py:142-186.
The base-level mapping comes from ``_get_base_alignment``; the default implementation drives BWA
(bwapy if importable, else the ``bwa`` executable), which is not available offline — tests and
bench inject a subclass (``synthetic.SyntheticAligner``) instead.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import namedtuple

import numpy

from .genome import Genome

ApproximateSignalAlignment = namedtuple('ApproximateSignalAlignment',
                                        ['alignment', 'signal_range', 'reference_range',
                                         'read_sequence_range', 'reverse_complement',
                                         'reference_part', 'contig_name'])

_CIGAR = re.compile(r'(\d+)([A-Za-z=])')


class ApproximateAligner:
    def __init__(self, bwa_executable, reference, reference_filename, references_dict=None):
        if reference is None and references_dict is None:
            raise AssertionError('one of reference or reference_dict must be not None')
        self.bwa_executable = bwa_executable
        self.reference = reference
        self.references_dict = references_dict
        self.reference_filename = reference_filename
        if reference_filename is not None and not os.path.isfile(reference_filename + '.bwt'):
            subprocess.run([self.bwa_executable, 'index', reference_filename],
                           stderr=subprocess.PIPE, check=True)
        self.bwapy_aligner = None
        try:
            from bwapy import BwaAligner
            self.bwapy_aligner = BwaAligner(reference_filename, options='-x ont2d')
        except ImportError:
            sys.stderr.write("Could't import bwapy, will use bwa executable to align reads\n")

    # ---- pieces subclasses may replace ------------------------------------------------------
    @staticmethod
    def _parse_cigar(cigar):
        return [(int(n), op) for n, op in _CIGAR.findall(cigar)]

    def _get_reference_contig(self, contig_name):
        if self.references_dict is not None:
            return self.references_dict[contig_name]
        return self.reference

    def _bwa_hit(self, read):
        """-> (cigar, is_reverse_complement, 0-based position, contig) or None."""
        if self.bwapy_aligner:
            hits = self.bwapy_aligner.align_seq(''.join(read.sequence))
            if len(hits) == 0:
                return None
            h = hits[0]
            return h.cigar, h.orient == '-', h.pos, h.rname
        with tempfile.TemporaryDirectory(prefix='nadavca_tmp') as tmp:
            fq, sam = os.path.join(tmp, 'read.fastq'), os.path.join(tmp, 'read.sam')
            with open(fq, 'w') as f:
                f.write(read.fastq)
            subprocess.run([self.bwa_executable, 'mem', self.reference_filename, fq, '-o', sam],
                           stderr=subprocess.PIPE, check=True)
            with open(sam) as f:
                for line in f:
                    if line.startswith('@'):
                        continue
                    fields = line.rstrip('\n').split('\t')
                    flag = int(fields[1])
                    if flag & 4:
                        return None
                    return fields[5], bool(flag & 16), int(fields[3]) - 1, fields[2]
        return None

    def _get_base_alignment(self, read):
        """-> (base_mapping int (M,2) [read index, oriented reference index], is_rc, contig) or None.
        Matched bases only; for a reverse-complement hit both indices are expressed in the read's
        own orientation (reference index counted from the end), alignment.py:109-140."""
        hit = self._bwa_hit(read)
        if hit is None:
            return None
        cigar, is_rc, pos, contig = hit
        reference = self._get_reference_contig(contig)
        oriented = Genome.reverse_complement(read.sequence) if is_rc else read.sequence
        pairs, ri, gi = [], 0, pos
        for n, op in self._parse_cigar(cigar):
            if op == 'S' or op == 'I':
                ri += n
            elif op == 'D':
                gi += n
            elif op == 'M':
                same = numpy.nonzero(numpy.asarray(reference[gi:gi + n]) == numpy.asarray(oriented[ri:ri + n]))[0]
                pairs.extend((ri + int(d), gi + int(d)) for d in same)
                ri += n
                gi += n
            else:
                raise ValueError('Unknown cigar operation: {}'.format(op))
        if is_rc:
            pairs = [(len(read.sequence) - 1 - a, len(reference) - 1 - b) for a, b in reversed(pairs)]
        return numpy.array(pairs, dtype=int).reshape(-1, 2), is_rc, contig

    # ---- contract used by the estimator ------------------------------------------------------
    @staticmethod
    def convert_mapping(base_mapping, read):
        """(read base, reference base) pairs -> (signal index, reference base) for the bases the
        basecaller placed on the signal (alignment.py:54-60)."""
        m = read.sequence_to_signal_mapping
        cached = getattr(read, '_dense_signal_map', None)
        if cached is None or cached[0] is not m or cached[1] != len(m):
            keys = numpy.fromiter(m.keys(), dtype=numpy.int64, count=len(m))
            vals = numpy.fromiter(m.values(), dtype=numpy.int64, count=len(m))
            dense = numpy.full(int(keys.max()) + 1 if len(keys) else 0, -1, dtype=numpy.int64)
            dense[keys] = vals
            cached = (m, len(m), dense)
            read._dense_signal_map = cached
        dense = cached[2]
        bm = numpy.asarray(base_mapping, dtype=numpy.int64).reshape(-1, 2)
        a = bm[:, 0]
        inside = (a >= 0) & (a < dense.size)
        sig = numpy.where(inside, dense[numpy.where(inside, a, 0)] if dense.size else -1, -1)
        have = sig >= 0
        return numpy.stack([sig[have], bm[have, 1]], axis=1).astype(int).reshape(-1, 2)

    def get_signal_alignment(self, read, bandwidth):
        base_alignment = self._get_base_alignment(read)
        if base_alignment is None:
            return None
        base_mapping, is_rc, contig_name = base_alignment
        signal_mapping = self.convert_mapping(base_mapping, read)
        if len(signal_mapping) == 0:
            return None
        reference = self._get_reference_contig(contig_name)

        start_in_reference = int(signal_mapping[0][1])
        end_in_reference = int(signal_mapping[-1][1]) + 1
        signal_mapping[:, 1] -= start_in_reference
        if is_rc:  # report the range on the forward strand
            start_in_reference, end_in_reference = \
                len(reference) - end_in_reference, len(reference) - start_in_reference

        start_in_signal = int(signal_mapping[0][0])
        end_in_signal = int(signal_mapping[-1][0]) + 1
        slice_start = max(0, start_in_signal - bandwidth)
        slice_end = min(len(read.normalized_signal), end_in_signal + bandwidth)
        signal_mapping[:, 0] -= slice_start

        reference_part = reference[start_in_reference: end_in_reference]
        if is_rc:
            reference_part = Genome.reverse_complement(reference_part)
        return ApproximateSignalAlignment(
            alignment=signal_mapping,
            signal_range=(slice_start, slice_end),
            reference_range=(start_in_reference, end_in_reference),
            read_sequence_range=(int(base_mapping[0][0]), int(base_mapping[-1][0]) + 1),
            reverse_complement=is_rc,
            reference_part=reference_part,
            contig_name=contig_name)
