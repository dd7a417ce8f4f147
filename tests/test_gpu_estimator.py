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


def test_device_resident_consensus_path_reproduces_estimate_probabilities(fx, km):
    """The data path bench.py's cfg4_consensus workload times at N = 1 — log-likelihoods, scatter-add into the
    per-position sums, posterior, all on device-resident buffers (no host round trip of the sums) — against
    ``estimate_probabilities`` (pinned to the reference's Python by the consensus fixture above)."""
    import torch
    from nadavca_amd.estimator import ProbabilityEstimator
    from nadavca_amd.genome import Genome
    from nadavca_amd.device import (DeviceBatch, estimate_log_likelihoods_dev, consensus_accumulate_dev,
                                    posterior_segments_dev)
    cfg = dict(fx.config, tweak_signal_normalization=False)
    est = ProbabilityEstimator(km, fx.aligner(), cfg)
    reads = fx.reads()
    want = est.estimate_probabilities(fx.genome, reads)
    live, batch, _, _ = est._log_likelihood_batch(fx.genome, fx.reads())
    dev = torch.device('cuda', km.context.device)
    db = DeviceBatch(batch, dev)
    ll, status = estimate_log_likelihoods_dev(db, est.bandwidth, est.min_event_length, km, est.model_wobbling)
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    start = up([p.apx.reference_range[0] for p in live], np.int64)
    rev = up([1 if p.apx.reverse_complement else 0 for p in live], np.int32)
    acc, cov = consensus_accumulate_dev(km.context, db, ll, start, rev, status, est.normalization_event_length,
                                        len(fx.genome))
    groups = est.group_ranges([tuple(int(v) for v in p.apx.reference_range) for p in live])
    seg = np.concatenate([[0], np.cumsum([e - s for s, e in groups])]).astype(np.int64)
    pos = up(np.concatenate([np.arange(s, e) for s, e in groups]), np.int64)
    refnum = up(Genome.to_numerical(fx.genome), np.int32)[pos]
    post = posterior_segments_dev(km.context, acc[pos], refnum, up(seg, np.int64), km.get_k(), est.snp_prior).cpu().numpy()
    assert len(want) == len(groups)
    for g, c in enumerate(want):
        assert (c.start, c.end) == groups[g]
        assert np.array_equal(c.coverage, cov.cpu().numpy()[c.start:c.end])
        assert np.max(np.abs(post[seg[g]:seg[g + 1]] - c.values)) < 1e-12


@pytest.mark.parametrize('tweak', [True, False])
def test_estimate_snps_batch_equals_the_per_read_workflow(km, tweak):
    """``estimate_snps_batch`` (struct-of-arrays in, sums and signals device-resident) against ``estimate_snps``
    (the reference-shaped workflow, pinned to the reference's Python by the fixtures above): consensus chunks
    and per-read chunks, with and without the spline tweak."""
    from nadavca_amd import synthetic, defaults
    from nadavca_amd.align_signal import _load_config
    from nadavca_amd.alignment import ApproximateAligner
    from nadavca_amd.estimate_snps import estimate_snps, estimate_snps_batch
    from nadavca_amd.readbatch import ReadBatch, BaseAlignmentBatch, SyntheticBatchAligner
    model = synthetic.load_model_arrays()
    genome = np.random.default_rng(41).integers(0, 4, 1500).astype(np.int32)
    specs = [synthetic.make_read_spec(np.random.default_rng([42, i]), genome, model, i, length=160, spread=30,
                                      substitution_rate=0.03) for i in range(30)]
    bases = np.array(list('ACGT'))[genome]
    cfg = dict(_load_config(defaults.CONFIG_FILE), tweak_signal_normalization=tweak)
    aligner = synthetic.make_synthetic_aligner(ApproximateAligner, bases)
    bms = [np.asarray(s['base_mapping'], dtype=np.int64).reshape(-1, 2) for s in specs]
    ba = BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]),
                            np.concatenate([[0], np.cumsum([len(b) for b in bms])]), [s['reverse'] for s in specs])
    for independent in (False, True):
        want = estimate_snps(None, synthetic.reads_from_specs(specs), reference=bases, config=cfg, kmer_model=km,
                             independent=independent, aligner=aligner)
        rb = ReadBatch.from_reads(synthetic.reads_from_specs(specs))
        got = estimate_snps_batch(genome, rb, config=cfg, kmer_model=km, independent=independent,
                                  aligner=SyntheticBatchAligner(genome, ba))
        if independent:
            assert len(got) == len(want) == 30
            got = [got.chunk(j) for j in range(len(got))]
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert (g.start, g.end) == (w.start, w.end)
            assert np.array_equal(g.coverage, w.coverage)
            assert np.max(np.abs(g.values - w.values)) < 1e-9


def _simulated(n, seed, **kw):
    from nadavca_amd import synthetic
    model = synthetic.load_model_arrays()
    genome = np.random.default_rng(seed).integers(0, 4, 3000).astype(np.int32)
    specs = [synthetic.make_read_spec(np.random.default_rng([seed + 1, i]), genome, model, i, length=200,
                                      spread=30, **kw) for i in range(n)]
    return genome, np.array(list('ACGT'))[genome], specs


def test_cigar_anchor_stage_feeds_the_kernels(km):
    """§8 f3: the approximate-alignment stage as the reference runs it — aligner hit (CIGAR, strand, position)
    -> matched bases -> anchors -> signal window (alignment.py:62-186) — in front of the GPU refinement.  BWA
    itself is replaced by a stub that reports each simulated read's true hit as a CIGAR string; everything
    after ``_bwa_hit`` is the package's own code, and the refined alignments must equal those obtained from
    the simulated base mapping directly."""
    from nadavca_amd import synthetic
    from nadavca_amd.alignment import ApproximateAligner
    from nadavca_amd.estimator import ProbabilityEstimator
    from nadavca_amd.read import Read
    genome, bases, specs = _simulated(24, 31, substitution_rate=0.06)

    class CigarAligner(ApproximateAligner):
        def __init__(self, reference):
            self.reference, self.references_dict, self.bwapy_aligner = reference, None, None

        def _bwa_hit(self, read):
            s = read._spec
            if s is specs[5]:
                return None                                        # an unmapped read
            clip = 30 if s is specs[2] else 0                       # one hit with soft-clipped ends
            cigar = '%dM' % s['length'] if not clip else '%dS%dM%dS' % (clip, s['length'] - 2 * clip, clip)
            return cigar, s['reverse'], s['g0'] + clip, 'synthetic'

    from nadavca_amd.align_signal import _load_config
    from nadavca_amd import defaults
    config = dict(_load_config(defaults.CONFIG_FILE), tweak_signal_normalization=False)
    reads_a, reads_b = synthetic.reads_from_specs(specs), synthetic.reads_from_specs(specs)
    Read.normalize_reads_device(reads_a, context=km.context)
    Read.normalize_reads_device(reads_b, context=km.context)
    via_cigar = ProbabilityEstimator(km, CigarAligner(bases), config).get_refined_alignments(reads_a)
    direct = ProbabilityEstimator(km, synthetic.make_synthetic_aligner(ApproximateAligner, bases),
                                  config).get_refined_alignments(reads_b)
    assert via_cigar[5] is None and direct[5] is not None
    same = 0
    for j, (a, b) in enumerate(zip(via_cigar, direct)):
        if j in (2, 5):
            continue
        assert a[0].reference_range == b[0].reference_range and a[0].reverse_complement == b[0].reverse_complement
        assert np.array_equal(a[1], b[1])
        same += 1
    assert same == 22
    # the clipped hit covers a shorter reference range, inside the unclipped one
    (lo, hi), (LO, HI) = via_cigar[2][0].reference_range, direct[2][0].reference_range
    assert LO <= lo < hi <= HI and hi - lo < HI - LO
    assert via_cigar[2][1][0][0] == lo and via_cigar[2][1][-1][0] == hi - 1


def test_detect_meth_rows(km, tmp_path):
    """§8 f4: ``detect_meth`` = align_signal's renormalisation loop on the GPU + per-event scores; the CSV rows
    against the same scores computed event by event from align_signal's own output."""
    import csv
    from scipy.stats import norm
    from nadavca_amd import synthetic
    from nadavca_amd.alignment import ApproximateAligner
    from nadavca_amd.align_signal import align_signal
    from nadavca_amd.detect_meth import detect_meth
    from nadavca_amd.genome import Genome
    _, bases, specs = _simulated(6, 41)
    aligner = synthetic.make_synthetic_aligner(ApproximateAligner, bases)
    out = tmp_path / 'meth.csv'
    detect_meth(None, synthetic.reads_from_specs(specs), 'CG', str(out), kmer_model=km, aligner=aligner)
    with open(out, newline='') as f:
        rows = list(csv.reader(f))
    assert rows[0] == ['Filename', 'Position', 'Sequence context', 'Position scores', 'Aggregated score']
    want = []
    for i, (read, (apx, al)) in enumerate(align_signal(None, synthetic.reads_from_specs(specs), kmer_model=km,
                                                       aligner=aligner)):
        seq = ''.join(apx.reference_part)
        exp = km.get_expected_signal(Genome.to_numerical(apx.reference_part), [], [])
        pos = seq.find('CG')
        while pos != -1:
            if pos >= 5 and pos + 6 <= len(al) and all(al[p][2] > al[p][1] for p in range(pos - 5, pos + 6)):
                sc = [-np.log(max(1e-50, 2 * norm.cdf(-abs(np.mean(read.normalized_signal[al[p][1]:al[p][2]])
                                                            - exp[p]) / 0.35287208)))
                      for p in range(pos - 5, pos + 6)]
                want.append(('read%d' % i, pos, seq[pos - 5:pos + 6], sc))
            pos = seq.find('CG', pos + 1)
    assert len(rows) - 1 == len(want) > 20
    for row, (name, pos, ctx, sc) in zip(rows[1:], want):
        assert row[0] == name and int(row[1]) == pos and row[2] == ctx
        got = np.array(row[3].split(','), dtype=float)
        assert np.allclose(got, sc, rtol=1e-12, atol=0)
        assert float(row[4]) == pytest.approx(max(sum(sc[i:i + 3]) for i in range(9)), rel=1e-12)
