"""Nanopore read container (mirrors /root/reference/nadavca/read.py:9-94).

``normalize_reads`` (global median/MAD + clip) and ``tweak_signal_normalization`` (spline through
(event mean, expected level) pairs) are host steps adjacent to the GPU path; they use the same
numpy/scipy calls as the reference so their results are identical.  fast5 loading needs h5py and is
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
        """One median / median-absolute-deviation over ALL reads' samples, clip to +-5
        (read.py:67-81).  numpy.median equals statistics.median on the same values."""
        values = numpy.concatenate([numpy.asarray(r.raw_signal, dtype=float) for r in reads]) \
            if reads else numpy.zeros(0)
        shift = numpy.median(values)
        scale = numpy.median(abs(values - shift))
        for read in reads:
            read.normalized_signal = numpy.clip((read.raw_signal - shift) / scale, -5, 5)

    def tweak_signal_normalization(self, alignment, expected_means):
        """Smoothing spline from event means to expected levels, applied to the whole signal
        (read.py:83-94).  alignment: (R, 2) absolute event ranges."""
        data = []
        for event, expected_mean in zip(alignment, expected_means):
            mean = numpy.mean(self.normalized_signal[event[0]: event[1]])
            if abs(expected_mean - mean) <= 1:
                data.append((mean, expected_mean))
        data.sort()
        means = [d[0] for d in data]
        expected = [d[1] for d in data]
        spline = interpolate.splrep(means, expected, s=len(means))
        self.tweaked_normalized_signal = interpolate.splev(self.normalized_signal, spline)
