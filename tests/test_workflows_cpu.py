"""Host logic of the consumers either side of the hot path: methylation scoring, the .npz /
TSV writers and the command line.  No GPU: the alignments are hand-made."""
import io
import types

import numpy as np
import pytest
from scipy.stats import norm


class _Apx:
    def __init__(self, bases, start=100, rc=False):
        self.reference_part = np.array(list(bases))
        self.reference_range = (start, start + len(bases))
        self.reverse_complement = rc
        self.contig_name = 'contig1'


class _LevelModel:
    """Expected level = a fixed number per base (stands in for KmerModel.get_expected_signal)."""
    LEVELS = np.array([-1.0, -0.3, 0.4, 1.1])

    def get_expected_signal(self, reference, context_before, context_after):
        return self.LEVELS[np.asarray(reference)]


def _loop_scores(signal, alignment, bases, pattern, model):
    """The scoring written the slow way (one event at a time), as the check."""
    from nadavca_amd.genome import Genome
    exp = model.get_expected_signal(Genome.to_numerical(np.array(list(bases))), [], [])
    base = alignment[0][1]
    out, pos = [], bases.find(pattern)
    while pos != -1:
        row = []
        for i in range(-5, 6):
            if not 0 <= pos + i < len(alignment):
                continue
            ev = signal[alignment[pos + i][1] - base:alignment[pos + i][2] - base]
            if len(ev) == 0:
                break
            p = norm.cdf(-abs(np.mean(ev) - exp[pos + i]) / 0.35287208) * 2
            row.append(-np.log(max(1e-50, p)))
        if len(row) == 11:
            out.append((pos, bases[pos - 5:pos + 6], row))
        pos = bases.find(pattern, pos + 1)
    return out


def _case(seed, n=60, empty_at=()):
    rng = np.random.default_rng(seed)
    bases = ''.join(rng.choice(list('ACGT'), n))
    lens = rng.integers(1, 9, n)
    lens[list(empty_at)] = 0
    starts = 37 + np.concatenate([[0], np.cumsum(lens)])
    alignment = np.stack([np.arange(n) + 100, starts[:-1], starts[1:]], axis=1)
    signal = rng.normal(0.0, 1.0, int(lens.sum()))
    return bases, alignment, signal


@pytest.mark.parametrize('seed,empty', [(0, ()), (1, (20,)), (2, (0, 59)), (3, (7, 8, 30))])
def test_meth_scores_equal_the_event_loop(seed, empty):
    from nadavca_amd.detect_meth import calculate_meth_scores, maxs3
    bases, alignment, signal = _case(seed, empty_at=empty)
    model = _LevelModel()
    for pattern in ('A', 'CG', bases[5:8], bases[:2], bases[-3:]):
        got = calculate_meth_scores(signal, alignment, _Apx(bases), pattern, model)
        want = _loop_scores(signal, alignment, bases, pattern, model)
        assert [(p, c) for p, c, _ in got] == [(p, c) for p, c, _ in want]
        for (_, _, a), (_, _, b) in zip(got, want):
            assert np.allclose(a, b, rtol=1e-13, atol=0)
            assert maxs3(a) == pytest.approx(max(sum(b[i:i + 3]) for i in range(9)), rel=1e-13)


def test_cdf_scoring_values():
    from nadavca_amd.detect_meth import cdf_scoring
    assert cdf_scoring([0.5, 0.5], 0.5) == pytest.approx(0.0, abs=1e-15)
    assert cdf_scoring([1.0], 0.0) == pytest.approx(-np.log(2 * norm.cdf(-1 / 0.35287208)))
    assert cdf_scoring([100.0], 0.0) == pytest.approx(-np.log(1e-50))     # floor of the p-value


def _read(raw, seq='ACGTAC'):
    r = types.SimpleNamespace()
    r.raw_signal = np.asarray(raw)
    r.sequence = np.array(list(seq))
    return r


def test_alignment_npz_layout(tmp_path):
    from nadavca_amd.writers import labelled_raw_cut, write_alignment_npz
    read = _read(np.arange(100, 160, dtype=np.int16))
    alignment = np.array([[7, 10, 14], [8, 14, 15], [9, 15, 21], [10, 21, 30]])
    apx = _Apx('GATC', start=7)
    cut, labels = labelled_raw_cut(read, apx, alignment)
    assert cut.tolist() == list(range(110, 121))              # first event start .. last event start
    assert ''.join(labels) == 'GNNNATNNNNN'                   # bases at the starts of all events but the last
    assert write_alignment_npz(str(tmp_path / 'r'), read, apx, alignment)
    z = np.load(tmp_path / 'r.npz')
    assert z['arr_0'].tolist() == cut.tolist() and ''.join(z['arr_1']) == 'GNNNATNNNNN'
    assert z['arr_2'].tolist() == ['7', '+', 'contig1', 'ACGTAC']
    with pytest.raises(ValueError):
        labelled_raw_cut(read, apx, np.array([[7, 10, 14], [8, 14, 14], [9, 14, 21], [10, 21, 30]]))
    assert not write_alignment_npz(str(tmp_path / 'e'), read, apx, np.array([[7, 10, 10]]))


def test_chunk_tsv(tmp_path):
    from nadavca_amd.estimator import Chunk
    from nadavca_amd.writers import write_chunks
    chunks = [Chunk(1, 3, np.array([[.25, .25, .25, .25], [.7, .1, .1, .1]])),
              Chunk(5, 6, np.array([[0., 0., 1., 0.]]), coverage=np.array([4]))]
    buf = io.StringIO()
    write_chunks(chunks, list('ACGTACGT'), buf)
    lines = buf.getvalue().splitlines()
    assert lines[0].split('\t') == ['index', 'base', 'coverage', 'A', 'C', 'G', 'T'] and len(lines) == 4
    assert lines[2].split('\t')[:3] == ['2', 'G', '1'] and lines[3].split('\t')[:3] == ['5', 'C', '4']


def test_command_line_surface():
    from nadavca_amd import defaults
    from nadavca_amd.cli import build_parser
    p = build_parser()
    a = p.parse_args(['-k', 'm.hdf5', 'ref.fa', 'reads', 'snp', '-i', '-o', 'out'])
    assert (a.function, a.independent, a.output, a.kmer_model) == ('snp', True, 'out', 'm.hdf5')
    a = p.parse_args(['ref.fa', 'reads', 'align'])
    assert a.function == 'align' and a.output is None and a.group_name == defaults.GROUP_NAME
    a = p.parse_args(['ref.fa', 'reads', 'meth', '-p', 'CCWGG'])
    assert (a.pattern, a.renorm_rounds, a.bwa_executable) == ('CCWGG', defaults.RENORM_ROUNDS,
                                                              defaults.BWA_EXECUTABLE)
    with pytest.raises(SystemExit):
        p.parse_args(['ref.fa', 'reads', 'meth'])
