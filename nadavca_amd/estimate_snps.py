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


def _check_status(what, status, live):
    """Per-read failures of a batch kernel (device tensor of NVK_READ_* codes): invalid input raises ValueError
    with the reads' indices in the ReadBatch; a band wider than the compiled kernels serve (READ_TOO_WIDE, a
    capability limit of this build, not bad input) only drops that read — it stays out of the sums like a read
    without a path — with a note on stderr."""
    from ._lib import READ_TOO_WIDE
    if not bool((status < 0).any()):
        return
    import torch
    wide = status == READ_TOO_WIDE
    bad = torch.nonzero((status < 0) & ~wide).reshape(-1)[:8]
    if bad.numel():
        raise ValueError('%s: invalid input for read(s) %s (status %s)'
                         % (what, live[bad].tolist(), status[bad].tolist()))
    sys.stderr.write('%s: %d read(s) skipped, band wider than the compiled kernels serve (first: %s)\n'
                     % (what, int(wide.sum()), live[torch.nonzero(wide).reshape(-1)[:8]].tolist()))


def _apply_splines(context, dbatch, fits):
    """``signal = splev(signal, spline of its read)`` in place for a (sub-)batch (read.py:94; the kernel restates
    FITPACK's evaluation).  Reads without a fit keep their signal (the reference would fail on them): a placeholder
    spline for the kernel, their samples put back afterwards."""
    import numpy
    import torch
    from .device import splev_groups_dev
    t, c, knot_off, fitted = fits
    device = dbatch.device
    ident_t = numpy.array([-5.0] * 4 + [5.0] * 4)
    ident_c = numpy.array([-5.0, -5.0 / 3, 5.0 / 3, 5.0, 0.0, 0.0, 0.0, 0.0])
    lens = numpy.diff(knot_off)
    lens2 = numpy.where(fitted, lens, ident_t.size)
    koff2 = numpy.concatenate([[0], numpy.cumsum(lens2)]).astype(numpy.int64)
    t2 = numpy.empty(int(koff2[-1]))
    c2 = numpy.empty(int(koff2[-1]))
    src = numpy.repeat(fitted, lens2)
    t2[src], c2[src] = t, c
    t2[~src] = numpy.tile(ident_t, int((~fitted).sum()))
    c2[~src] = numpy.tile(ident_c, int((~fitted).sum()))
    up = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(device)
    keep = saved = None
    if not fitted.all():
        keep = torch.repeat_interleave(up(~fitted), dbatch.sig_off[1:] - dbatch.sig_off[:-1],
                                       output_size=dbatch.total_signal)
        saved = dbatch.signal[keep]
    splev_groups_dev(context, dbatch.signal, dbatch.sig_off, up(t2), up(c2), up(koff2), 3, out=dbatch.signal)
    if keep is not None:
        dbatch.signal[keep] = saved


def _apply_device_fits(context, dbatch, t, c, fit):
    """``signal = splev(signal, spline of its read)`` in place with the splines as nvk_spline_fit_dev left them on
    the device (8 knots + 8 coefficients per read; reads without a fit carry a placeholder and get their samples
    back)."""
    import torch
    from .device import splev_groups_dev
    n = dbatch.n
    knot_off = torch.arange(n + 1, dtype=torch.int64, device=dbatch.device) * 8
    unfit = fit != 0
    keep = saved = None
    if bool(unfit.any()):
        keep = torch.repeat_interleave(unfit, dbatch.sig_off[1:] - dbatch.sig_off[:-1],
                                       output_size=dbatch.total_signal)
        saved = dbatch.signal[keep]
    splev_groups_dev(context, dbatch.signal, dbatch.sig_off, t.reshape(-1), c.reshape(-1), knot_off, 3,
                     out=dbatch.signal)
    if keep is not None:
        dbatch.signal[keep] = saved


def _tweak_signal_normalization(context, kmer_model, dbatch, config, fit_workers=0, spline_fit='device'):
    """``Read.tweak_signal_normalization`` (read.py:83-94) for the batch, in place on its signal: pre-alignment
    without transition rows, expected levels, per-event means, the fit, the evaluation — five kernels, nothing on
    the host (``spline_fit='device'``: nvk_spline_fit_dev restates the pass of FITPACK's ``curfit`` that decides
    these fits and checks per read that it does).  A read the kernel reports as outside that case (fit == 2: NaN
    levels, all means equal) is fitted by FITPACK itself, as is everything with ``spline_fit='host'`` (scipy in
    ``fit_workers`` processes — the path of rounds 1-2, kept as the cross-check of the kernel).
    -> number of reads fitted."""
    from . import splinefit
    from .device import refine_alignment_dev, expected_levels_dev, event_means_dev, spline_fit_dev
    bw, mel = config['bandwidth'], config['min_event_length']
    ev0, st0 = refine_alignment_dev(dbatch, bw, mel, kmer_model, False)
    expected = expected_levels_dev(dbatch, kmer_model, with_contexts=True)
    means = event_means_dev(dbatch, context, ev0, st0)
    if spline_fit == 'device':
        t, c, fit = spline_fit_dev(context, means, expected, dbatch.ref_off, st0)
        beyond = fit == 2
        if not bool(beyond.any()):
            _apply_device_fits(context, dbatch, t, c, fit)
            return int((fit == 0).sum())
        fits = splinefit.merge_host_fits(means.cpu().numpy(), expected.cpu().numpy(),
                                         dbatch.ref_off.cpu().numpy(), t.cpu().numpy(), c.cpu().numpy(),
                                         fit.cpu().numpy())
    elif spline_fit == 'host':
        fits = splinefit.fit_splines(means.cpu().numpy(), expected.cpu().numpy(), dbatch.ref_off.cpu().numpy(),
                                     st0.cpu().numpy() == 0, workers=fit_workers)
    else:
        raise ValueError("spline_fit: 'device' or 'host'")
    _apply_splines(context, dbatch, fits)
    return int(fits[3].sum())


# what the last estimate_snps_batch call saw (bench.py reports it): reads in the batch, reads with an approximate
# alignment, reads whose log-likelihoods came back with status 0, reads the spline tweak fitted
last_batch_counts = {}


class IndependentChunks:
    """``estimate_snps(independent=True)`` for a batch: one chunk per read that produced one, as arrays —
    read ``reads[j]`` covers reference positions [start[j], end[j]) and has posterior rows
    values[row_off[j]:row_off[j+1]] (coverage is 1 everywhere, as in the reference's per-read call)."""

    def __init__(self, reads, start, end, values, row_off):
        self.reads, self.start, self.end, self.values, self.row_off = reads, start, end, values, row_off

    def __len__(self):
        return len(self.reads)

    def chunk(self, j):
        return Chunk(int(self.start[j]), int(self.end[j]), self.values[self.row_off[j]:self.row_off[j + 1]])


def estimate_snps_batch(reference_num, read_batch, config=defaults.CONFIG_FILE,
                        kmer_model=defaults.KMER_MODEL_FILE, independent=False, aligner=None, fit_workers=0,
                        group=None, distributed=None, dst=0, spline_fit='device'):
    """``estimate_snps`` for a struct-of-arrays ``ReadBatch`` (nadavca_amd/readbatch.py) without per-read
    Python: the steps of estimate_snps.py:57-70 and estimator.py:59-121,199-236 — ONE median/MAD over all
    reads, approximate alignment, the spline tweak (pre-alignment without transition rows, expected levels,
    per-event means, fit — nvk_spline_fit_dev; ``spline_fit='host'`` sends it to scipy in ``fit_workers``
    processes instead, nadavca_amd/splinefit.py — and evaluation, all on the device), log-likelihoods, normalise / strand-flip / per-position sum, grouping, posterior — with the
    signals, the sums and everything between them resident on the device.
    ``reference_num``: the reference as base codes; ``aligner``: as for ``align_signal_batch``.
    -> list of Chunk (consensus) or IndependentChunks.

    Several GPUs (``distributed=True``, or a ``group``; default: whenever torch.distributed is initialised with
    more than one rank): every rank passes ITS shard of the reads as ``read_batch`` and the two exchange steps of
    the path run over torch.distributed (RCCL over xGMI with the nccl backend) — the pooled median / MAD of
    estimate_snps.py:61 as an exact distributed selection (256 counts per pass cross ranks, distributed.py:
    pooled_centre_scale), and for ``independent=False`` ONE reduce(sum) of the packed per-position sums plus a
    small all-gather of the chunk intervals.  The consensus Chunk list is returned on rank ``dst`` (None
    elsewhere); ``independent=True`` returns every rank's own IndependentChunks."""
    import numpy
    import torch
    from . import readbatch
    from .device import (DeviceBatch, normalize_groups_dev, refine_alignment_dev, expected_levels_dev,
                         event_means_dev, estimate_log_likelihoods_dev, consensus_accumulate_dev,
                         posterior_segments_dev)
    from .estimator import ProbabilityEstimator
    if isinstance(config, str):
        with open(config, 'r') as file:
            config = yaml.safe_load(file)
    if isinstance(kmer_model, str):
        kmer_model = KmerModel.load_from_hdf5(kmer_model)
    if aligner is None:
        raise ValueError('estimate_snps_batch needs a batch aligner (BWA has no batch adapter offline)')
    context = kmer_model.context
    device = torch.device('cuda', context.device)
    rb = read_batch
    bw, mel = config['bandwidth'], config['min_event_length']
    reference_num = numpy.ascontiguousarray(reference_num, dtype=numpy.int32)
    L = reference_num.size
    raw = torch.from_numpy(rb.raw_signal).to(device)
    if raw.dtype != torch.float64:
        raw = raw.to(torch.float64)
    total = int(rb.sig_off[-1])
    if distributed is None:
        distributed = group is not None
        if not distributed:
            import torch.distributed as tdist
            distributed = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    if distributed:
        from . import distributed as D
        from .device import select_hist_dev, normalize_apply_dev
        # all reads of ALL ranks pooled (estimate_snps.py:61): exact distributed median and MAD
        centre, scale = D.pooled_centre_scale(select_hist_dev(context, raw), total, device=device, group=group)
        norm = normalize_apply_dev(context, raw, centre, scale, out=raw)
    else:
        one_group = torch.tensor([0, total], dtype=torch.int64, device=device)
        norm, _ = normalize_groups_dev(context, raw, one_group, out=raw)   # all reads pooled (estimate_snps.py:61)
    ba = aligner.get_base_alignments(rb)
    sa = readbatch.signal_alignments(rb, ba, bw, reference_num, kmer_model.get_k(),
                                     kmer_model.get_central_position(), device=device)
    n_live = int(sa.live.numel())
    last_batch_counts.clear()
    last_batch_counts.update(reads=int(rb.n), reads_aligned=n_live, reads_ok=0, reads_fitted=None)
    if n_live == 0:
        if independent:
            return IndependentChunks(numpy.zeros(0, dtype=numpy.int64), *[numpy.zeros(0)] * 3,
                                     numpy.zeros(1, dtype=numpy.int64))
        if not distributed:
            return []
        # (a rank whose shard aligned nowhere still takes part in the exchange, with empty sums)
        acc = torch.zeros((L, kmer_model.alphabet_size), dtype=torch.float64, device=device)
        cov = torch.zeros(L, dtype=torch.int64, device=device)
        return _consensus_chunks(context, kmer_model, config, reference_num, acc, cov, [], device, True, group, dst)
    dbatch = DeviceBatch.from_windows(norm, sa, device)
    if config['tweak_signal_normalization']:
        last_batch_counts['reads_fitted'] = _tweak_signal_normalization(context, kmer_model, dbatch, config,
                                                                        fit_workers, spline_fit)
    ll, status = estimate_log_likelihoods_dev(dbatch, bw, mel, kmer_model, config['model_wobbling'])
    _check_status('estimate_log_likelihoods', status, sa.live)
    rev32 = sa.reverse.to(torch.int32)
    nel = config['normalization_event_length']
    k, prior = kmer_model.get_k(), config['snp_prior_probability']
    ref_dev = torch.from_numpy(reference_num).to(device)
    ok = (status == 0)
    last_batch_counts['reads_ok'] = int(ok.sum())
    if independent:
        # every read a segment of its own, laid end to end (estimate_snps.py:63-68)
        acc, _ = consensus_accumulate_dev(context, dbatch, ll, sa.ref_off[:-1].contiguous(), rev32, status, nel,
                                          dbatch.total_ref)
        rlen = sa.ref_off[1:] - sa.ref_off[:-1]
        owner = torch.repeat_interleave(torch.arange(n_live, dtype=torch.int64, device=device), rlen,
                                        output_size=dbatch.total_ref)
        pos = sa.ref_start[owner] + (torch.arange(dbatch.total_ref, dtype=torch.int64, device=device)
                                     - sa.ref_off[:-1][owner])
        post = posterior_segments_dev(context, acc, ref_dev[pos], sa.ref_off.contiguous(), k, prior)
        okh = ok.cpu().numpy()
        off = sa.ref_off.cpu().numpy()
        return _independent_result(sa, okh, off, post.cpu().numpy())
    acc, cov = consensus_accumulate_dev(context, dbatch, ll, sa.ref_start.contiguous(), rev32, status, nel, L)
    starts, ends = sa.ref_start[ok].cpu().numpy(), sa.ref_end[ok].cpu().numpy()
    return _consensus_chunks(context, kmer_model, config, reference_num, acc, cov,
                             list(zip(starts.tolist(), ends.tolist())), device, distributed, group, dst)


def _consensus_chunks(context, kmer_model, config, reference_num, acc, cov, ranges, device, distributed, group, dst):
    """Per-position sums -> grouped posteriors (estimator.py:205-236).  Distributed: the sums of all ranks meet in
    ONE reduce of the packed device buffer, the intervals in a small all-gather; the posterior runs on ``dst``."""
    import numpy
    import torch
    from .device import posterior_segments_dev
    from .estimator import ProbabilityEstimator
    if distributed:
        from . import distributed as D
        ranges = D.gather_ranges(ranges, device=device, group=group)
        tot = D.reduce_consensus_tensors(acc, cov, dst=dst, group=group)
        if tot is None:
            return None
        acc, cov = tot
    groups = ProbabilityEstimator.group_ranges(ranges)
    if not groups:
        return []
    k, prior = kmer_model.get_k(), config['snp_prior_probability']
    ref_dev = torch.from_numpy(numpy.ascontiguousarray(reference_num, dtype=numpy.int32)).to(device)
    seg = numpy.concatenate([[0], numpy.cumsum([e - s for s, e in groups])]).astype(numpy.int64)
    pos = torch.from_numpy(numpy.concatenate([numpy.arange(s, e) for s, e in groups])).to(device)
    post = posterior_segments_dev(context, acc[pos], ref_dev[pos], torch.from_numpy(seg).to(device), k, prior)
    post_h, cov_h = post.cpu().numpy(), cov.cpu().numpy()
    return [Chunk(s, e, post_h[seg[g]:seg[g + 1]], cov_h[s:e].copy()) for g, (s, e) in enumerate(groups)]


def _independent_result(sa, okh, off, post):
    import numpy
    lens = (off[1:] - off[:-1])[okh]
    row_off = numpy.concatenate([[0], numpy.cumsum(lens)]).astype(numpy.int64)
    rows = numpy.concatenate([numpy.arange(off[j], off[j + 1]) for j in numpy.nonzero(okh)[0]]) if okh.any() \
        else numpy.zeros(0, dtype=numpy.int64)
    return IndependentChunks(sa.live.cpu().numpy()[okh], sa.ref_start.cpu().numpy()[okh],
                             sa.ref_end.cpu().numpy()[okh], post[rows], row_off)
