"""Sequence helpers (mirrors the parts of /root/reference/nadavca/genome.py:11-62 the estimator
uses): FASTA/FASTQ parsing into arrays of single-character bases, numerical encoding and reverse
complement."""
import numpy

from .alphabet import inv_alphabet, complement

_COMP = numpy.zeros(256, dtype=numpy.uint8)
for _a, _b in complement.items():
    _COMP[ord(_a)] = ord(_b)
_NUM = numpy.full(256, -1, dtype=numpy.int64)
for _a, _i in inv_alphabet.items():
    _NUM[ord(_a)] = _i


class Genome:
    def __init__(self, desc_line=None):
        self.description = desc_line
        self.lines = []
        self.bases = None

    def _finish(self):
        self.bases = numpy.array(list(''.join(self.lines)))

    @staticmethod
    def to_numerical(sequence):
        """Array of bases 'A','C','G','T' -> int array 0..3 (KeyError on anything else, like the
        reference's dict lookup)."""
        seq = numpy.asarray(sequence)
        if seq.size == 0:
            return numpy.zeros(0, dtype=numpy.int64)
        codes = numpy.frombuffer(''.join(seq.tolist()).encode('ascii'), dtype=numpy.uint8)
        out = _NUM[codes]
        if (out < 0).any():
            raise KeyError(str(seq[numpy.nonzero(out < 0)[0][0]]))
        return out

    @staticmethod
    def reverse_complement(sequence):
        seq = numpy.asarray(sequence)
        if seq.size == 0:
            return numpy.array([], dtype='<U1')
        codes = numpy.frombuffer(''.join(seq.tolist()).encode('ascii'), dtype=numpy.uint8)
        rc = _COMP[codes][::-1]
        if (rc == 0).any():
            raise KeyError('non-ACGT base in sequence')
        return numpy.array(list(rc.tobytes().decode('ascii')))

    @staticmethod
    def load_from_fasta(filename):
        result, current = [], None
        with open(filename, 'r') as file:
            for line in file:
                if line.startswith('>'):
                    current = Genome(line.rstrip())
                    result.append(current)
                elif current is not None:
                    current.lines.append(line.rstrip())
        for genome in result:
            genome._finish()
        return result

    @staticmethod
    def create_from_fastq_string(fastq_string):
        result, current, expect_sequence = [], None, False
        for line in fastq_string.split('\n'):
            if len(line) > 0 and line[0] == '@':
                current = Genome(line)
                result.append(current)
                expect_sequence = True
            elif expect_sequence:
                current.lines.append(line.rstrip())
                expect_sequence = False
        for genome in result:
            genome._finish()
        return result

    @staticmethod
    def load_from_fastq(filename):
        with open(filename, 'r') as file:
            return Genome.create_from_fastq_string(file.read())
