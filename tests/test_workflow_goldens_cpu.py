"""CPU side of tests/golden/workflows.npz (outputs of the reference's own Python, oracle/make_golden_workflows.py):
the approximate-anchor stage — CIGAR strings with S / M / I / D operations on both strands through
``_get_base_alignment`` and ``get_signal_alignment`` (/root/reference/nadavca/alignment.py:69-186), per read and in
the struct-of-arrays form of readbatch.py — and the TSV writer of ``Chunk.print`` (estimator.py:22-31).  No GPU:
index arithmetic and text formatting only."""
import io

import numpy as np
import pytest

from est_fixture import EstimatorFixture


@pytest.fixture(scope='module')
def wf():
    return EstimatorFixture('workflows.npz')


class _Read:
    pass


def _cigar_read(z, i):
    pre = 'cig%d_' % i
    r = _Read()
    r.sequence = np.array(list(str(z[pre + 'sequence'])))
    r.sequence_to_signal_mapping = {int(a): int(b) for a, b in zip(z[pre + 'map_keys'], z[pre + 'map_vals'])}
    r.normalized_signal = np.zeros(int(z[pre + 'n_samples']))
    r.hit = (str(z[pre + 'cigar']), str(z[pre + 'orient']) == '-', int(z[pre + 'pos']), 'contig1')
    return r


def _aligner(genome):
    from nadavca_amd.alignment import ApproximateAligner
    al = object.__new__(ApproximateAligner)       # (the constructor would run `bwa index`)
    al.bwa_executable, al.reference, al.reference_filename, al.bwapy_aligner = 'bwa', None, None, None
    al.references_dict = {'contig1': genome}
    al._bwa_hit = lambda read: read.hit           # the mapper's answer: CIGAR in, everything after it is ours
    return al


def test_cigar_to_base_mapping_and_anchors_equal_the_reference(wf):
    z = wf.z
    al = _aligner(wf.genome)
    ops = set()
    for i in range(int(z['cig_n'])):
        pre = 'cig%d_' % i
        read = _cigar_read(z, i)
        ops |= set(ch for ch in str(z[pre + 'cigar']) if ch.isalpha())
        bm, is_rc, contig = al._get_base_alignment(read)
        assert np.array_equal(np.asarray(bm).reshape(-1, 2), z[pre + 'base_mapping']), i
        assert bool(is_rc) == bool(z[pre + 'is_rc']) and contig == 'contig1'
        apx = al.get_signal_alignment(read, 150)
        assert np.array_equal(np.asarray(apx.alignment).reshape(-1, 2), z[pre + 'anchors']), i
        assert [*apx.signal_range, *apx.reference_range, *apx.read_sequence_range] == z[pre + 'ranges'].tolist()
        assert ''.join(np.asarray(apx.reference_part).tolist()) == str(z[pre + 'reference_part'])
        assert apx.reverse_complement == bool(z[pre + 'is_rc'])
    assert ops == set('SMID')                      # the fixture really exercises every operation
    read.hit = None                                # an unmapped read
    assert al._get_base_alignment(read) is None and al.get_signal_alignment(read, 150) is None


def test_batched_anchor_stage_equals_the_reference(wf):
    """The same stage for all reads at once (readbatch.signal_alignments, torch on the CPU here): anchors, signal
    windows, reference ranges and parts from the reference's base mappings."""
    from nadavca_amd import readbatch
    from nadavca_amd.genome import Genome
    z = wf.z
    n = int(z['cig_n'])
    reads = [_cigar_read(z, i) for i in range(n)]
    for r in reads:
        r.raw_signal = r.normalized_signal
    rb = readbatch.ReadBatch.from_reads(reads)
    bms = [z['cig%d_base_mapping' % i] for i in range(n)]
    ba = readbatch.BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]),
                                      np.concatenate([[0], np.cumsum([len(b) for b in bms])]),
                                      [bool(z['cig%d_is_rc' % i]) for i in range(n)])
    sa = readbatch.signal_alignments(rb, ba, 150, Genome.to_numerical(wf.genome), 6, 2, device='cpu').host()
    assert sa.live.tolist() == list(range(n))
    for i in range(n):
        pre = 'cig%d_' % i
        rng = z[pre + 'ranges']
        assert np.array_equal(sa.anchors[sa.anc_off[i]:sa.anc_off[i + 1]], z[pre + 'anchors'])
        assert [int(sa.slice_start[i]), int(sa.slice_start[i] + sa.win_len[i])] == rng[:2].tolist()
        assert [int(sa.ref_start[i]), int(sa.ref_end[i])] == rng[2:4].tolist()
        assert [int(sa.read_seq_start[i]), int(sa.read_seq_end[i])] == rng[4:6].tolist()
        part = np.array(list('ACGT'))[sa.reference[sa.ref_off[i]:sa.ref_off[i + 1]]]
        assert ''.join(part.tolist()) == str(z[pre + 'reference_part'])


def test_chunk_print_writes_the_reference_tsv(wf):
    from nadavca_amd.estimator import Chunk
    z = wf.z
    out = io.StringIO()
    Chunk.print_head(out)
    for ci in range(int(z['snps_n_chunks'])):
        s, e = z['snps_c%d_range' % ci].tolist()
        Chunk(s, e, z['snps_c%d_values' % ci], z['snps_c%d_coverage' % ci]).print(out, wf.genome)
    assert out.getvalue() == str(z['snps_tsv'])
