"""GPU side of tests/golden/workflows.npz — outputs of the reference's own Python layer
(oracle/make_golden_workflows.py) — for the rows of SURVEY.md 8 that used to be tested against themselves:

* BASELINE config 1 at its stated size: ``align_signal`` on 16 reads of ~4 000 int16 samples against a 1 kb
  reference (/root/reference/nadavca/align_signal.py:43-81), per read and through ``align_signal_batch``;
* ``detect_meth`` CSV rows (detect_meth.py:21-120);
* the per-read arrays ``align_signal_command`` saves (align_signal.py:83-147);
* ``estimate_snps`` on the same 16 reads: chunk ranges, coverage, posteriors (1e-5), and the TSV text.
Integer results exact; floating-point scores to 1e-9 relative; posteriors to the north star's 1e-5."""
import csv
import io
import json
import os

import numpy as np
import pytest

from est_fixture import EstimatorFixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def wf():
    return EstimatorFixture('workflows.npz')


@pytest.fixture(scope='module')
def km():
    from nadavca_amd.kmer_model import KmerModel
    from nadavca_amd import defaults
    return KmerModel.load_from_hdf5(defaults.KMER_MODEL_FILE)


def _named_reads(wf, subset=None):
    reads = wf.reads(normalize=False, subset=subset)
    for i, r in zip(range(wf.n) if subset is None else subset, reads):
        r.name = 'read%02d.fast5' % i
    return reads


def test_config1_align_signal_16_reads_equal_the_reference(wf, km):
    import nadavca_amd
    reads = _named_reads(wf)
    assert len(reads) == 16 and reads[0].raw_signal.dtype == np.int16
    assert 3500 < np.mean([len(r.raw_signal) for r in reads]) < 4700      # ~4 000 samples each
    out = list(nadavca_amd.align_signal(None, reads, config=dict(wf.config), kmer_model=km, aligner=wf.aligner()))
    assert len(out) == 16
    for i, (read, (apx, alignment)) in enumerate(out):
        assert np.array_equal(alignment, wf.z['as_r%d_alignment' % i]), i
        assert np.allclose(read.normalized_signal[:64], wf.z['as_r%d_norm_head' % i], rtol=1e-10, atol=1e-12)
        got = [float(np.sum(read.normalized_signal)), float(np.sum(np.abs(read.normalized_signal)))]
        assert np.allclose(got, wf.z['as_r%d_norm_checksum' % i], rtol=1e-9)


def test_config1_align_signal_batch_equals_the_reference(wf, km):
    """The struct-of-arrays workflow (int16 raw signals across PCIe once, device normalisation, device anchor
    stage) on the same 16 reads against the same reference rows."""
    from nadavca_amd.align_signal import align_signal_batch
    from nadavca_amd.genome import Genome
    from nadavca_amd.readbatch import ReadBatch, BaseAlignmentBatch, SyntheticBatchAligner
    rb = ReadBatch.from_reads(_named_reads(wf))
    assert rb.raw_signal.dtype == np.int16
    bms = [np.asarray(s['base_mapping'], dtype=np.int64).reshape(-1, 2) for s in wf.specs]
    ba = BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]),
                            np.concatenate([[0], np.cumsum([len(b) for b in bms])]), [s['reverse'] for s in wf.specs])
    out = align_signal_batch(None, rb, config=dict(wf.config), kmer_model=km,
                             aligner=SyntheticBatchAligner(Genome.to_numerical(wf.genome), ba))
    assert out.n_aligned == 16
    for i in range(16):
        assert np.array_equal(out.alignment_of(i), wf.z['as_r%d_alignment' % i]), i
        norm = out.normalized_signal(i)
        assert np.allclose(norm[:64], wf.z['as_r%d_norm_head' % i], rtol=1e-10, atol=1e-12)


def test_detect_meth_rows_equal_the_reference(wf, km, tmp_path):
    from nadavca_amd.detect_meth import detect_meth
    n = int(wf.z['meth_n_reads'])
    path = str(tmp_path / 'meth.csv')
    detect_meth(None, _named_reads(wf, subset=range(n)), str(wf.z['pattern']), path, config=dict(wf.config),
                kmer_model=km, aligner=wf.aligner())
    got = list(csv.reader(open(path, newline='')))
    exp = list(csv.reader(io.StringIO(str(wf.z['meth_csv']))))
    assert got[0] == exp[0] and len(got) == len(exp) and len(exp) > 100
    for g, e in zip(got[1:], exp[1:]):
        assert g[:3] == e[:3]                                  # file, position, sequence context
        gs, es = np.array(g[3].split(','), dtype=float), np.array(e[3].split(','), dtype=float)
        assert gs.shape == es.shape == (11,) and np.allclose(gs, es, rtol=1e-9, atol=1e-12)
        assert np.isclose(float(g[4]), float(e[4]), rtol=1e-9)


def test_alignment_npz_arrays_equal_the_reference(wf, km, tmp_path):
    import nadavca_amd
    from nadavca_amd.writers import write_alignment_npz
    n = int(wf.z['npz_n_reads'])
    reads = _named_reads(wf, subset=range(n))
    out = list(nadavca_amd.align_signal(None, reads, config=dict(wf.config), kmer_model=km, aligner=wf.aligner()))
    for i, (read, (apx, alignment)) in enumerate(out):
        base = str(tmp_path / ('read%02d' % i))
        assert write_alignment_npz(base, read, apx, alignment)
        z = np.load(base + '.npz')
        assert np.array_equal(z['arr_0'], wf.z['npz_r%d_raw_cut' % i]) and z['arr_0'].dtype == np.int16
        assert ''.join(z['arr_1'].tolist()) == str(wf.z['npz_r%d_labels' % i])
        assert [str(x) for x in z['arr_2'].tolist()] == json.loads(str(wf.z['npz_r%d_info' % i]))


def test_estimate_snps_16_reads_and_tsv(wf, km):
    import nadavca_amd
    from nadavca_amd.writers import write_chunks
    chunks = nadavca_amd.estimate_snps(None, _named_reads(wf), reference=wf.genome, config=dict(wf.config),
                                       kmer_model=km, independent=False, aligner=wf.aligner())
    assert len(chunks) == int(wf.z['snps_n_chunks'])
    for ci, c in enumerate(chunks):
        assert [c.start, c.end] == wf.z['snps_c%d_range' % ci].tolist()
        assert np.array_equal(c.coverage, wf.z['snps_c%d_coverage' % ci])
        assert np.max(np.abs(c.values - wf.z['snps_c%d_values' % ci])) < 1e-5
    # the TSV: same lines, the 16-decimal probabilities within 1e-5 of the reference's
    text = io.StringIO()
    write_chunks(chunks, wf.genome, text)
    got, exp = text.getvalue().splitlines(), str(wf.z['snps_tsv']).splitlines()
    assert got[0] == exp[0] and len(got) == len(exp)
    for g, e in zip(got[1:], exp[1:]):
        gf, ef = g.split('\t'), e.split('\t')
        assert gf[:3] == ef[:3]
        assert np.allclose(np.array(gf[3:], dtype=float), np.array(ef[3:], dtype=float), atol=1e-5)
