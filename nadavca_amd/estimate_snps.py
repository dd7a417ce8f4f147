"""``estimate_snps`` (mirrors /root/reference/nadavca/estimate_snps.py:13-70): SNP posteriors per
reference position, either a consensus over all reads or one Chunk per read (independent=True).
``aligner`` (optional, not in the reference signature) injects an approximate aligner."""
import os
import sys

import yaml

from . import defaults
from .alignment import ApproximateAligner
from .estimator import ProbabilityEstimator, Chunk  # noqa: F401
from .genome import Genome
from .kmer_model import KmerModel
from .read import Read


def estimate_snps(reference_filename, reads, reference=None, config=defaults.CONFIG_FILE,
                  kmer_model=defaults.KMER_MODEL_FILE, bwa_executable=defaults.BWA_EXECUTABLE,
                  independent=False, group_name=defaults.GROUP_NAME, aligner=None):
    if isinstance(config, str):
        try:
            with open(config, 'r') as file:
                config = yaml.safe_load(file)
        except FileNotFoundError:
            sys.stderr.write('failed to load config: {} not found\n'.format(config))
            return None
    if isinstance(kmer_model, str):
        try:
            kmer_model = KmerModel.load_from_hdf5(kmer_model)
        except FileNotFoundError:
            sys.stderr.write('failed to load k-mer model: {} not found\n'.format(kmer_model))
            return None
    if reference is None:
        try:
            reference = Genome.load_from_fasta(reference_filename)[0].bases
        except FileNotFoundError:
            sys.stderr.write("failed to process: reference {} doesn't exist\n".format(reference_filename))
            return None
    if aligner is None:
        aligner = ApproximateAligner(bwa_executable, reference, reference_filename)
    estimator = ProbabilityEstimator(kmer_model, aligner, config)

    if isinstance(reads, str):
        base = reads
        reads = [os.path.join(base, f) for f in os.listdir(base)
                 if f.endswith('.fast5') and not os.path.isdir(os.path.join(base, f))]
    reads = [Read.load_from_fast5(r, group_name) if isinstance(r, str) else r for r in reads]
    # ONE median/MAD over all reads (estimate_snps.py:61)
    Read.normalize_reads_device(reads, context=getattr(kmer_model, 'context', None))

    if independent:
        chunks = estimator.estimate_probabilities_independent(reference, reads)
        if any(c is None for c in chunks):
            # the reference indexes chunks[0] of an empty list here (estimate_snps.py:66-67)
            raise IndexError('a read produced no chunk (not aligned, or no valid path in the band)')
        return chunks
    return estimator.estimate_probabilities(reference, reads)
