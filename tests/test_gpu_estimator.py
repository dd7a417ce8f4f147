"""GPU parity at the estimator level against tests/golden/estimator.npz — outputs of the
reference's own Python layer (ProbabilityEstimator, estimate_snps, align_signal) on simulated
reads.  Integer alignments exact; SNP posteriors within 1e-5 (north-star tolerance; observed far
tighter)."""
import numpy as np
import pytest

from est_fixture import EstimatorFixture

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-5


@pytest.fixture(scope='module')
def fx():
    return EstimatorFixture()


@pytest.fixture(scope='module')
def km():
    from nadavca_amd.kmer_model import KmerModel
    from nadavca_amd import defaults
    return KmerModel.load_from_hdf5(defaults.KMER_MODEL_FILE)


def test_get_refined_alignment(fx, km):
    from nadavca_amd.estimator import ProbabilityEstimator
    est = ProbabilityEstimator(km, fx.aligner(), fx.config)
    reads = fx.reads()
    res = est.get_refined_alignments(reads)
    for i, r in enumerate(res):
        assert r is not None
        assert np.array_equal(r[1], fx.z['r%d_refined' % i])
    one = est.get_refined_alignment(reads[1])   # a reverse-strand read through the per-read entry
    assert np.array_equal(one[1], fx.z['r1_refined'])
    assert one[1][0][0] > one[1][-1][0]         # reference positions descend on the reverse strand


@pytest.mark.parametrize('tweak', [1, 0])
def test_estimate_probabilities_consensus(fx, km, tweak):
    from nadavca_amd.estimator import ProbabilityEstimator
    cfg = dict(fx.config, tweak_signal_normalization=bool(tweak))
    est = ProbabilityEstimator(km, fx.aligner(), cfg)
    reads = fx.reads()
    chunks = est.estimate_probabilities(fx.genome, reads)
    assert len(chunks) == int(fx.z['cons_t%d_n' % tweak])
    for c_i, c in enumerate(chunks):
        assert [c.start, c.end] == fx.z['cons_t%d_c%d_range' % (tweak, c_i)].tolist()
        assert np.array_equal(c.coverage, fx.z['cons_t%d_c%d_coverage' % (tweak, c_i)])
        exp = fx.z['cons_t%d_c%d_values' % (tweak, c_i)]
        assert c.values.shape == exp.shape
        assert np.max(np.abs(c.values - exp)) < PROB_TOL
        assert np.allclose(c.values.sum(axis=1), 1.0)
    if tweak:
        for i, r in enumerate(reads):
            got = np.array([float(np.sum(r.tweaked_normalized_signal)),
                            float(np.sum(np.abs(r.tweaked_normalized_signal)))])
            assert np.allclose(got, fx.z['r%d_tweaked_checksum' % i], rtol=1e-9)
            assert np.allclose(r.tweaked_normalized_signal[:64], fx.z['r%d_tweaked_head' % i], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('tweak', [1, 0])
def test_estimate_probabilities_independent(fx, km, tweak):
    from nadavca_amd.estimator import ProbabilityEstimator
    cfg = dict(fx.config, tweak_signal_normalization=bool(tweak))
    est = ProbabilityEstimator(km, fx.aligner(), cfg)
    chunks = est.estimate_probabilities_independent(fx.genome, fx.reads())
    for i, c in enumerate(chunks):
        assert [c.start, c.end] == fx.z['ind_t%d_r%d_range' % (tweak, i)].tolist()
        exp = fx.z['ind_t%d_r%d_values' % (tweak, i)]
        assert np.max(np.abs(c.values - exp)) < PROB_TOL
        assert np.all(c.coverage == 1)


def test_estimate_snps_entry_point(fx, km):
    import nadavca_amd
    reads = fx.reads(normalize=False)
    chunks = nadavca_amd.estimate_snps(None, reads, reference=fx.genome, config=dict(fx.config),
                                       kmer_model=km, independent=False, aligner=fx.aligner())
    assert len(chunks) == int(fx.z['cons_t1_n'])
    assert np.max(np.abs(chunks[0].values - fx.z['cons_t1_c0_values'])) < PROB_TOL
    ind = nadavca_amd.estimate_snps(None, fx.reads(normalize=False), reference=fx.genome,
                                    config=dict(fx.config), kmer_model=km, independent=True,
                                    aligner=fx.aligner())
    assert len(ind) == fx.n
    # the called base (argmax posterior) is the reference base on clean simulated reads
    from nadavca_amd.genome import Genome
    c = chunks[0]
    called = np.argmax(c.values, axis=1)
    assert np.mean(called == Genome.to_numerical(fx.genome[c.start:c.end])) > 0.97


def test_align_signal_renorm_loop(fx, km):
    """G7: per-read normalisation, align, linear re-fit, re-align, re-fit (align_signal.py:52-81)."""
    import nadavca_amd
    n = int(fx.z['as_n'])
    reads = fx.reads(normalize=False, subset=range(n))
    out = list(nadavca_amd.align_signal(None, reads, config=dict(fx.config), kmer_model=km,
                                        aligner=fx.aligner()))
    assert len(out) == n
    for i, (read, (apx, alignment)) in enumerate(out):
        assert np.array_equal(alignment, fx.z['as_r%d_alignment' % i])
        assert np.allclose(read.normalized_signal[:64], fx.z['as_r%d_norm_head' % i], rtol=1e-10, atol=1e-12)
        assert np.allclose(np.sum(read.normalized_signal), fx.z['as_r%d_norm_checksum' % i], rtol=1e-9)


def test_align_signal_batch_equals_the_per_read_workflow(km):
    """``align_signal_batch`` (struct-of-arrays in, no per-read Python, windows cut on the device) against
    ``align_signal`` (the reference-shaped generator, pinned to the reference's Python by the G7 fixture
    above): the same (R, 3) rows, the same rescaled normalised signals."""
    from nadavca_amd import synthetic
    from nadavca_amd.alignment import ApproximateAligner
    from nadavca_amd.align_signal import align_signal, align_signal_batch
    from nadavca_amd.readbatch import ReadBatch, BaseAlignmentBatch, SyntheticBatchAligner
    model = synthetic.load_model_arrays()
    genome = np.random.default_rng(21).integers(0, 4, 4000).astype(np.int32)
    specs = [synthetic.make_read_spec(np.random.default_rng([22, i]), genome, model, i, length=220, spread=40,
                                      substitution_rate=0.05) for i in range(40)]
    reads = synthetic.reads_from_specs(specs)
    aligner = synthetic.make_synthetic_aligner(ApproximateAligner, np.array(list('ACGT'))[genome])
    per_read = list(align_signal(None, reads, kmer_model=km, aligner=aligner))
    rb = ReadBatch.from_reads(synthetic.reads_from_specs(specs))
    bms = [np.asarray(s['base_mapping'], dtype=np.int64).reshape(-1, 2) for s in specs]
    ba = BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]),
                            np.concatenate([[0], np.cumsum([len(b) for b in bms])]), [s['reverse'] for s in specs])
    out = align_signal_batch(None, rb, kmer_model=km, aligner=SyntheticBatchAligner(genome, ba))
    assert out.n_aligned == len(per_read) == 40 and out.live.tolist() == list(range(40))
    for j, (read, (apx, rows)) in enumerate(per_read):
        assert np.array_equal(out.alignment_of(j), rows)
        assert np.array_equal(out.normalized_signal(j), read.normalized_signal)
