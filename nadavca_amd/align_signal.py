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


class AlignedBatch:
    """What ``align_signal_batch`` returns: for the reads that aligned (``live``: their indices in the
    ReadBatch), the (reference position, event start, event end) rows of all reads end to end — read
    ``live[j]`` at rows [ref_off[j], ref_off[j+1]) — the approximate alignment they started from
    (``approximate``: a readbatch.SignalAlignmentBatch) and the per-round linear fits.  The reads' rescaled
    ``normalized_signal`` stays on the device (``normalized``) until ``normalized_signal(j)`` asks for it."""

    def __init__(self, live, status, alignment, ref_off, approximate, fits, normalized, sig_off):
        self.live, self.status, self.alignment, self.ref_off = live, status, alignment, ref_off
        self.approximate, self.fits, self.normalized, self.sig_off = approximate, fits, normalized, sig_off
        self.n_aligned = int((status == 0).sum())

    def alignment_of(self, j):
        """(R, 3) int64 rows of live read j, or None where the band held no path."""
        if self.status[j] != 0:
            return None
        return self.alignment[self.ref_off[j]:self.ref_off[j + 1]]

    def normalized_signal(self, read_index):
        lo, hi = int(self.sig_off[read_index]), int(self.sig_off[read_index + 1])
        return self.normalized[lo:hi].cpu().numpy()


def align_signal_batch(reference_filename, read_batch, config=defaults.CONFIG_FILE,
                       kmer_model=defaults.KMER_MODEL_FILE, renorm_rounds=defaults.RENORM_ROUNDS, aligner=None):
    """``align_signal`` for a struct-of-arrays ``ReadBatch`` (nadavca_amd/readbatch.py): the same steps per read
    as the reference's loop (align_signal.py:52-81) — per-read median/MAD normalisation, approximate
    alignment, banded alignment, linear re-fit, re-alignment, linear re-fit — with no per-read Python: the raw
    signals and the flat tables cross PCIe once in their native dtypes, the approximate-alignment stage and
    the window cutting are tensor operations on the device, and everything between stays there.  ``aligner``: an object with
    ``get_base_alignments(read_batch) -> BaseAlignmentBatch`` and ``reference_num`` (the reference as base
    codes).  -> AlignedBatch."""
    import numpy
    import torch
    from . import readbatch
    from .device import DeviceBatch, normalize_groups_dev, refine_renorm_loop_dev, to_host
    config = _load_config(config)
    if isinstance(kmer_model, str):
        kmer_model = KmerModel.load_from_hdf5(kmer_model)
    if aligner is None:
        raise ValueError('align_signal_batch needs a batch aligner (BWA has no batch adapter offline)')
    context = kmer_model.context
    device = torch.device('cuda', context.device)
    rb = read_batch
    raw = torch.from_numpy(rb.raw_signal).to(device)
    if raw.dtype != torch.float64:
        raw = raw.to(torch.float64)
    sig_off_dev = torch.from_numpy(rb.sig_off).to(device)
    norm, _ = normalize_groups_dev(context, raw, sig_off_dev, out=raw)   # per read (align_signal.py:54)
    ba = aligner.get_base_alignments(rb)
    sa = readbatch.signal_alignments(rb, ba, config['bandwidth'], aligner.reference_num, kmer_model.get_k(),
                                     kmer_model.get_central_position(), device=device)
    n_live = int(sa.live.numel())
    if n_live == 0:
        return AlignedBatch(numpy.zeros(0, dtype=numpy.int64), numpy.zeros(0, dtype=numpy.int32),
                            numpy.zeros((0, 3), dtype=numpy.int64), numpy.zeros(1, dtype=numpy.int64), sa, [],
                            norm, rb.sig_off)
    dbatch = DeviceBatch.from_windows(norm, sa, device)
    events, status, fits = refine_renorm_loop_dev(dbatch, config['bandwidth'], config['min_event_length'],
                                                  kmer_model, config['model_transitions'], renorm_rounds)
    from .estimate_snps import _check_status
    _check_status('refine_alignment', status, sa.live)   # (too-wide reads stay in `status`, like reads without a path)
    # the same linear maps for the samples outside the windows (the reference rescales the whole read,
    # align_signal.py:73): per read (x - intercept) / slope, fit after fit, on the device
    total = int(rb.sig_off[-1])
    lens = sig_off_dev[1:] - sig_off_dev[:-1]
    okay = (status == 0)
    for f in fits:
        slope = torch.ones(rb.n, dtype=torch.float64, device=device)
        icpt = torch.zeros(rb.n, dtype=torch.float64, device=device)
        slope[sa.live] = torch.where(okay, f[:, 0], torch.ones_like(f[:, 0]))
        icpt[sa.live] = torch.where(okay, f[:, 1], torch.zeros_like(f[:, 1]))
        norm -= torch.repeat_interleave(icpt, lens, output_size=total)
        norm /= torch.repeat_interleave(slope, lens, output_size=total)
    # (R, 3) rows: reference position (descending on the reverse strand), events in read coordinates
    rlen = sa.ref_off[1:] - sa.ref_off[:-1]
    n_rows = int(sa.ref_off[-1])
    owner = torch.repeat_interleave(torch.arange(n_live, dtype=torch.int64, device=device), rlen, output_size=n_rows)
    inner = torch.arange(n_rows, dtype=torch.int64, device=device) - sa.ref_off[:-1][owner]
    rows = torch.empty((n_rows, 3), dtype=torch.int64, device=device)
    rows[:, 0] = torch.where(sa.reverse[owner], sa.ref_end[owner] - inner - 1, sa.ref_start[owner] + inner)
    start = sa.slice_start[owner]
    rows[:, 1] = events[:, 0] + start
    rows[:, 2] = events[:, 1] + start
    rb.normalized = norm
    return AlignedBatch(sa.live.cpu().numpy(), status.cpu().numpy(), to_host(rows), sa.ref_off.cpu().numpy(),
                        sa, [f.cpu().numpy() for f in fits], norm, rb.sig_off)
