"""``align_signal`` (mirrors /root/reference/nadavca/align_signal.py:13-81).

Generator of ``(read, (approximate_alignment, alignment))`` with ``alignment`` the (R, 3) int
array (reference position, event start, event end).  The per-read renormalise / re-align loop of
the reference (renorm_rounds, default 3: linear re-fit, re-align, linear re-fit) is kept, but every
round processes all reads in one batched GPU launch, and the signal stays on the device between the
rounds (per-event means, least-squares re-fit and rescale are kernels: include/nadavca_hip.h).
Extensions over the reference signature, both optional: ``reads`` may hold ``Read`` objects (not only
fast5 paths), and ``aligner`` injects an approximate aligner (BWA is not available offline)."""
import sys

import yaml

from . import defaults
from .alignment import ApproximateAligner
from .estimator import ProbabilityEstimator
from .genome import Genome
from .kmer_model import KmerModel
from .read import Read


def _load_config(config):
    if isinstance(config, str):
        with open(config, 'r') as file:
            return yaml.safe_load(file)
    return config


def load_model_and_estimator(reference_filename, config=defaults.CONFIG_FILE, kmer_model=None,
                             bwa_executable=defaults.BWA_EXECUTABLE, aligner=None):
    if kmer_model is None:
        kmer_model = defaults.KMER_MODEL_FILE
    try:
        config = _load_config(config)
    except FileNotFoundError:
        sys.stderr.write('failed to load config: {} not found\n'.format(config))
        return None
    if isinstance(kmer_model, str):
        kmer_model = KmerModel.load_from_hdf5(kmer_model)
    if aligner is None:
        try:
            references = Genome.load_from_fasta(reference_filename)
        except FileNotFoundError:
            sys.stderr.write("failed to process: reference {} doesn't exist\n".format(reference_filename))
            return None
        references_dict = {r.description[1:]: r.bases for r in references}
        aligner = ApproximateAligner(bwa_executable, None, reference_filename, references_dict)
    return kmer_model, ProbabilityEstimator(kmer_model, aligner, config)


def align_signal(reference_filename, reads, config=defaults.CONFIG_FILE,
                 kmer_model=defaults.KMER_MODEL_FILE, bwa_executable=defaults.BWA_EXECUTABLE,
                 group_name=defaults.GROUP_NAME, renorm_rounds=defaults.RENORM_ROUNDS, aligner=None):
    loaded = load_model_and_estimator(reference_filename, config, kmer_model, bwa_executable, aligner)
    if loaded is None:
        return
    kmer_model, estimator = loaded
    loaded_reads = [Read.load_from_fast5(item, group_name) if isinstance(item, str) else item
                    for item in reads]
    # per-read normalisation (align_signal.py:54), all reads in one kernel
    Read.normalize_reads_device(loaded_reads, per_read=True, context=kmer_model.context)
    # align / re-fit / re-align / re-fit (align_signal.py:55-80), device-resident
    results = estimator.refine_and_renormalize(loaded_reads, renorm_rounds)
    for read, res in zip(loaded_reads, results):
        if res is None:
            # the reference raises TypeError here (it unpacks None, align_signal.py:55)
            raise TypeError('read could not be aligned (no approximate alignment or no path in the band)')
        yield read, res
