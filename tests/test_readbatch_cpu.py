"""Host logic of the batched workflows (no GPU): the struct-of-arrays approximate-alignment stage
(nadavca_amd/readbatch.py) against the per-read restatement of the reference's arithmetic
(/root/reference/nadavca/alignment.py:54-60,142-186, /root/reference/nadavca/estimator.py:49-57), which the
estimator fixtures pin to the reference's own Python."""
import numpy as np


def test_signal_alignments_equal_the_per_read_path():
    from nadavca_amd import synthetic, readbatch
    from nadavca_amd.alignment import ApproximateAligner
    from nadavca_amd.estimator import ProbabilityEstimator
    from nadavca_amd.genome import Genome
    from nadavca_amd.readbatch import ReadBatch, BaseAlignmentBatch

    model = synthetic.synth_model_arrays(3, k=6, central=2)
    genome = np.random.default_rng(11).integers(0, 4, 3000).astype(np.int32)
    specs = []
    for i in range(120):
        rng = np.random.default_rng([12, i])
        s = synthetic.make_read_spec(rng, genome, model, i, length=150, spread=40,
                                     substitution_rate=0.15 if i % 3 else 0.0, trim=0 if i % 5 == 0 else 3)
        if i % 17 == 0:                       # a read the aligner does not place
            s['base_mapping'] = np.zeros((0, 2), dtype=int)
        if i % 19 == 0:                       # a read whose matched bases have no sample position
            s['sequence_to_signal_mapping'] = {}
        specs.append(s)
    reads = synthetic.reads_from_specs(specs)
    for r in reads:
        r.normalized_signal = np.zeros(len(r.raw_signal))
    rb = ReadBatch.from_reads(reads)
    bms = [np.asarray(s['base_mapping'], dtype=np.int64).reshape(-1, 2) for s in specs]
    off = np.concatenate([[0], np.cumsum([len(b) for b in bms])])
    ba = BaseAlignmentBatch(np.concatenate([b[:, 0] for b in bms]), np.concatenate([b[:, 1] for b in bms]), off,
                            [s['reverse'] for s in specs])
    k, central = model[0], model[1]
    sa = readbatch.signal_alignments(rb, ba, 60, genome, k, central).host()   # torch on the CPU here

    class _Model:
        def get_k(self): return k
        def get_central_position(self): return central
    aligner = synthetic.make_synthetic_aligner(ApproximateAligner, np.array(list('ACGT'))[genome])
    est = ProbabilityEstimator.__new__(ProbabilityEstimator)
    est.kmer_model = _Model()
    live = []
    for j, r in enumerate(reads):
        apx = aligner.get_signal_alignment(r, 60)
        if apx is None:
            continue
        jj = len(live)
        live.append(j)
        assert np.array_equal(sa.anchors[sa.anc_off[jj]:sa.anc_off[jj + 1]], apx.alignment)
        assert (sa.slice_start[jj], sa.slice_start[jj] + sa.win_len[jj]) == tuple(apx.signal_range)
        assert sa.win_start[jj] == rb.sig_off[j] + apx.signal_range[0]
        assert (sa.ref_start[jj], sa.ref_end[jj]) == tuple(apx.reference_range)
        assert (sa.read_seq_start[jj], sa.read_seq_end[jj]) == tuple(apx.read_sequence_range)
        assert bool(sa.reverse[jj]) == apx.reverse_complement
        assert np.array_equal(sa.reference[sa.ref_off[jj]:sa.ref_off[jj + 1]], Genome.to_numerical(apx.reference_part))
        before, after = est._get_read_context(r, apx.read_sequence_range)
        assert np.array_equal(sa.context_before[sa.cb_off[jj]:sa.cb_off[jj + 1]], before)
        assert np.array_equal(sa.context_after[sa.ca_off[jj]:sa.ca_off[jj + 1]], after)
    assert sa.live.tolist() == live and 90 < len(live) < 120
    assert any(sa.cb_off[j + 1] > sa.cb_off[j] for j in range(len(live)))   # some reads do have contexts


def test_read_batch_from_reads_layout():
    from nadavca_amd import synthetic
    from nadavca_amd.readbatch import ReadBatch
    model = synthetic.synth_model_arrays(4, k=4, central=1)
    genome = np.random.default_rng(1).integers(0, 4, 500).astype(np.int32)
    specs = [synthetic.make_read_spec(np.random.default_rng([2, i]), genome, model, i, length=60, spread=10)
             for i in range(5)]
    reads = synthetic.reads_from_specs(specs)
    rb = ReadBatch.from_reads(reads)
    assert rb.n == 5 and rb.sig_off[-1] == sum(len(r.raw_signal) for r in reads)
    for j, r in enumerate(reads):
        assert np.array_equal(rb.raw_signal[rb.sig_off[j]:rb.sig_off[j + 1]], r.raw_signal)
        m = dict(zip(rb.map_base[rb.map_off[j]:rb.map_off[j + 1]].tolist(),
                     rb.map_sig[rb.map_off[j]:rb.map_off[j + 1]].tolist()))
        assert m == r.sequence_to_signal_mapping
