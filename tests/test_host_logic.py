"""CPU-only tests: host-side mirror of the reference interface (no GPU compute)."""
import io
import os

import numpy as np
import pytest

from est_fixture import EstimatorFixture


@pytest.fixture(scope='module')
def fx():
    return EstimatorFixture()


def test_normalize_reads_matches_reference(fx):
    """G5: one median/MAD over all reads (read.py:67-81), values from the reference's own run."""
    reads = fx.reads()
    for i, r in enumerate(reads):
        assert np.array_equal(r.normalized_signal[:64], fx.z['r%d_normalized_head' % i])
    assert np.array_equal(np.array([float(np.sum(r.normalized_signal)) for r in reads]),
                          fx.z['normalized_checksum'])
    assert all(np.all(np.abs(r.normalized_signal) <= 5) for r in reads)


def test_signal_alignment_contract(fx):
    reads = fx.reads()
    al = fx.aligner()
    for r, spec in zip(reads, fx.specs):
        a = al.get_signal_alignment(r, 150)
        s0, s1 = a.signal_range
        assert 0 <= s0 < s1 <= len(r.normalized_signal)
        assert a.alignment[0][1] == 0 and a.alignment[-1][1] == len(a.reference_part) - 1
        assert np.all(np.diff(a.alignment[:, 0]) >= 0) and np.all(np.diff(a.alignment[:, 1]) > 0)
        assert a.alignment[0][0] == min(150, a.alignment[0][0] + s0)  # slice starts bandwidth before the first anchor
        g0, g1 = a.reference_range
        part = fx.genome[g0:g1]
        from nadavca_amd.genome import Genome
        exp = Genome.reverse_complement(part) if spec['reverse'] else part
        assert ''.join(a.reference_part) == ''.join(exp)
        assert a.reverse_complement == spec['reverse']


def test_genome_helpers():
    from nadavca_amd.genome import Genome
    seq = np.array(list('ACGTTGCA'))
    assert Genome.to_numerical(seq).tolist() == [0, 1, 2, 3, 3, 2, 1, 0]
    assert ''.join(Genome.reverse_complement(np.array(list('AACGT')))) == 'ACGTT'
    g = Genome.create_from_fastq_string('@r1\nACGT\n+\n!!!!\n')
    assert ''.join(g[0].bases) == 'ACGT'
    with pytest.raises(KeyError):
        Genome.to_numerical(np.array(list('ACNT')))


def test_chunk_ordering_and_print():
    from nadavca_amd.estimator import Chunk
    a, b, c = Chunk(5, 9, None), Chunk(5, 7, None), Chunk(2, 30, None)
    assert sorted([a, b, c])[0] is c and sorted([a, b, c])[1] is b
    ch = Chunk(1, 3, np.array([[0.25, 0.25, 0.25, 0.25], [1, 0, 0, 0]], dtype=float))
    assert ch.coverage.tolist() == [1, 1]
    buf = io.StringIO()
    Chunk.print_head(buf)
    ch.print(buf, list('ACGT'))
    lines = buf.getvalue().splitlines()
    assert lines[0] == 'index\tbase\tcoverage\tA\tC\tG\tT'
    assert lines[1].startswith('1\tC\t1\t0.2500000000000000')


def test_cigar_base_mapping():
    """CIGAR -> matched-base mapping (alignment.py:109-140) without running BWA."""
    from nadavca_amd.alignment import ApproximateAligner

    class Fake(ApproximateAligner):
        def __init__(self, reference, hit):
            self.reference, self.references_dict, self._hit = reference, None, hit

        def _bwa_hit(self, read):
            return self._hit

    class R:
        pass
    ref = np.array(list('AACCGGTTAACC'))
    r = R()
    r.sequence = np.array(list('TTCCGATT'))   # 2 soft-clipped, CCG match, A mismatch (ref G), TT match
    m, rc, contig = Fake(ref, ('2S6M', False, 2, 'c'))._get_base_alignment(r)
    assert m.tolist() == [[2, 2], [3, 3], [4, 4], [6, 6], [7, 7]] and rc is False
    # reverse strand: indices are reported in the read's own orientation
    r.sequence = np.array(list('GGTTAA'))[::-1]
    from nadavca_amd.genome import Genome
    r.sequence = Genome.reverse_complement(np.array(list('CCGGTT')))
    m, rc, _ = Fake(ref, ('6M', True, 2, 'c'))._get_base_alignment(r)
    assert rc is True and m[:, 0].tolist() == [0, 1, 2, 3, 4, 5]
    assert m[:, 1].tolist() == [len(ref) - 1 - p for p in range(7, 1, -1)]


def test_c_abi_exports_every_declared_symbol():
    """Every function include/nadavca_hip.h declares is exported by the built library (no compute)."""
    import re
    from conftest import ROOT
    from nadavca_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'nadavca_hip.h')).read()
    declared = set(re.findall(r'\b(nvk_[a-z_0-9]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_compute_fails_loudly_without_gpu():
    from nadavca_amd import _lib
    lib = _lib.load()
    if lib.nvk_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.NadavcaHipError):
        _lib.Context(0)
