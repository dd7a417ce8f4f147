"""``detect_meth`` — per-event deviation scores around every occurrence of a sequence pattern (mirrors
/root/reference/nadavca/detect_meth.py:21-120; a consumer of the alignments).

The alignment itself — normalise, align, linear re-fit, re-align, linear re-fit — is ``align_signal`` (the
reference repeats that loop inline, detect_meth.py:75-106; it is the same arithmetic, and here the same
kernels).  What this module adds is the scoring: for an occurrence of ``pattern`` at reference-part position
p, the 11 events p-5 .. p+5 are each scored by how far their mean level is from the model's expected level,
    z = |mean(event) - expected| / 0.35287208,   score = -log(max(1e-50, 2 * Phi(-z))),
an occurrence counts only when all 11 events exist and are non-empty, and its aggregate is the largest sum of
three consecutive scores.  All events of a read are scored at once (segment sums instead of a Python loop per
event)."""
import csv
import os
import sys

import numpy as np
from scipy.special import ndtr

from . import defaults
from .align_signal import align_signal
from .genome import Genome

SMALLEST_PVAL = 1e-50
LEVEL_SD = 0.35287208     # detect_meth.py:24
FLANK = 5                 # events on each side of the pattern's first base


def cdf_scoring(raw, exp):
    """Score of one event (detect_meth.py:23-26)."""
    z = np.abs(np.mean(raw) - exp) / LEVEL_SD
    return -np.log(max(SMALLEST_PVAL, ndtr(-z) * 2.0))


def event_scores(signal_cut, alignment, expected):
    """Scores of all events of one read: ``alignment`` (R, 3) rows (position, start, end) in read coordinates,
    ``signal_cut`` = normalized_signal[alignment[0][1] : alignment[-1][2]].  An empty event scores NaN."""
    alignment = np.asarray(alignment)
    start = alignment[:, 1] - alignment[0][1]
    end = alignment[:, 2] - alignment[0][1]
    padded = np.append(np.asarray(signal_cut, dtype=float), 0.0)
    cuts = np.empty(2 * len(start), dtype=np.intp)
    cuts[0::2], cuts[1::2] = start, np.maximum(end, start)
    sums = np.add.reduceat(padded, cuts)[0::2]
    n = end - start
    with np.errstate(invalid='ignore', divide='ignore'):
        means = np.where(n > 0, sums / np.maximum(n, 1), np.nan)
        z = np.abs(means - np.asarray(expected, dtype=float)[:len(means)]) / LEVEL_SD
        return -np.log(np.maximum(SMALLEST_PVAL, ndtr(-z) * 2.0))


def calculate_meth_scores(signal_cut, alignment, apx_alignment, pattern, kmer_model):
    """-> [(position, sequence context, [11 scores])] for every scorable occurrence of ``pattern`` in the
    aligned reference part (detect_meth.py:28-60)."""
    bases = apx_alignment.reference_part
    expected = np.asarray(kmer_model.get_expected_signal(Genome.to_numerical(bases), [], []))
    seq = ''.join(np.asarray(bases).tolist())
    scores = event_scores(signal_cut, alignment, expected)
    n = len(alignment)
    features, pos = [], seq.find(pattern)
    while pos != -1:
        lo, hi = pos - FLANK, pos + FLANK + 1
        if lo >= 0 and hi <= n:
            window = scores[lo:hi]
            if not np.isnan(window).any():
                features.append((pos, seq[lo:hi], window.tolist()))
        pos = seq.find(pattern, pos + 1)
    return features


def maxs3(values):
    """Largest sum of three consecutive scores (detect_meth.py:63-65)."""
    v = np.asarray(values, dtype=float)
    return float(np.max(v[:-2] + v[1:-1] + v[2:]))


def detect_meth(reference_filename, reads, pattern, output, config=defaults.CONFIG_FILE,
                kmer_model=defaults.KMER_MODEL_FILE, bwa_executable=defaults.BWA_EXECUTABLE,
                group_name=defaults.GROUP_NAME, renorm_rounds=defaults.RENORM_ROUNDS, aligner=None):
    """CSV of (Filename, Position, Sequence context, Position scores, Aggregated score) rows, one per scorable
    pattern occurrence per read, to ``output`` (a path) or stdout.  ``reads``: fast5 paths or ``Read``
    objects; ``aligner``: optional approximate aligner (extension, as in ``align_signal``)."""
    from .align_signal import load_model_and_estimator
    loaded = load_model_and_estimator(reference_filename, config, kmer_model, bwa_executable, aligner)
    if loaded is None:
        return
    model = loaded[0]
    out = open(output, 'w', newline='') if output is not None else sys.stdout
    try:
        writer = csv.writer(out)
        writer.writerow(('Filename', 'Position', 'Sequence context', 'Position scores', 'Aggregated score'))
        names = [r if isinstance(r, str) else getattr(r, 'name', 'read%d' % i) for i, r in enumerate(reads)]
        results = align_signal(reference_filename, reads, config=config, kmer_model=model,
                               bwa_executable=bwa_executable, group_name=group_name,
                               renorm_rounds=renorm_rounds, aligner=aligner)
        for name, (read, (apx, alignment)) in zip(names, results):
            cut = read.normalized_signal[alignment[0][1]:alignment[-1][2]]
            for pos, context, scores in calculate_meth_scores(cut, alignment, apx, pattern, model):
                writer.writerow((name, pos, context, ','.join(map(str, scores)), maxs3(scores)))
    finally:
        if output is not None:
            out.close()


def detect_meth_command(args):
    reads = [os.path.join(args.read_basedir, fn) for fn in os.listdir(args.read_basedir) if fn.endswith('.fast5')]
    detect_meth(args.reference, reads, args.pattern, args.output, args.configuration, args.kmer_model,
                args.bwa_executable, args.group_name, args.renorm_rounds)
