"""Device-resident batches: torch tensors used purely as HBM buffers for the ``*_dev`` entry
points of the C-ABI (inputs already in HBM when the call starts; see include/nadavca_hip.h)."""
import ctypes as C

import numpy as np

from . import _lib


def _dp(t):
    return C.c_void_p(t.data_ptr())


def to_host(t):
    """A device tensor as a numpy array through page-locked host memory (torch keeps a cache of such blocks, so a
    caller that drops the previous batch's result gets the block back: no page faults on fresh memory, the copy at
    PCIe speed).  Meant for the large results of the batch workflows (96 MB of rows per 10 000 reads); small ones
    go ``.cpu()``."""
    import torch
    if t.numel() < (1 << 18):
        return t.cpu().numpy()
    try:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:
        return t.cpu().numpy()
    h.copy_(t)
    return h.numpy()


class DeviceBatch:
    """A flat batch (nadavca_amd.dtw.FlatBatch / synthetic.Batch layout) copied to one GPU."""

    def __init__(self, batch, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.n = int(batch.n)
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        self.signal = up(batch.signal, np.float64)
        self.sig_off = up(batch.sig_off, np.int64)
        self.reference = up(batch.reference, np.int32)
        self.ref_off = up(batch.ref_off, np.int64)
        self.context_before = up(batch.context_before, np.int32)
        self.cb_off = up(batch.cb_off, np.int64)
        self.context_after = up(batch.context_after, np.int32)
        self.ca_off = up(batch.ca_off, np.int64)
        self.anchors = up(batch.anchors, np.int32)
        self.anc_off = up(batch.anc_off, np.int64)
        self.total_signal = int(batch.sig_off[-1])
        self.total_ref = int(batch.ref_off[-1])
        self.total_anchors = int(batch.anc_off[-1])
        torch.cuda.synchronize(self.device)

    @classmethod
    def from_windows(cls, signal_dev, sa, device):
        """The DP inputs of ``readbatch.signal_alignments()`` (a SignalAlignmentBatch of tensors on ``device``)
        with every read's signal window gathered ON THE DEVICE out of ``signal_dev`` (the normalised signals of
        all reads end to end, f64 device tensor): no per-read array crosses PCIe."""
        import torch
        self = cls.__new__(cls)
        self.torch = torch
        self.device = torch.device(device)
        dev = self.device
        self.n = int(sa.live.numel())
        win_len = sa.win_len.to(dev)
        sig_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(win_len, 0)])
        self.sig_off = sig_off
        self.total_signal = int(sig_off[-1])
        # source index of every window sample: its position in the batch + (window start - batch offset)
        idx = torch.repeat_interleave(sa.win_start.to(dev) - sig_off[:-1], win_len, output_size=self.total_signal)
        idx += torch.arange(self.total_signal, dtype=torch.int64, device=dev)
        self.signal = signal_dev[idx]
        del idx
        nz = lambda t, m: t.to(dev).contiguous() if t.numel() else torch.zeros(m, dtype=t.dtype, device=dev)
        self.reference = nz(sa.reference, 1)
        self.ref_off = sa.ref_off.to(dev).contiguous()
        self.context_before = nz(sa.context_before, 1)
        self.cb_off = sa.cb_off.to(dev).contiguous()
        self.context_after = nz(sa.context_after, 1)
        self.ca_off = sa.ca_off.to(dev).contiguous()
        self.anchors = nz(sa.anchors.reshape(-1), 2)
        self.anc_off = sa.anc_off.to(dev).contiguous()
        self.total_ref = int(sa.ref_off[-1])
        self.total_anchors = int(sa.anc_off[-1])
        return self

    def pointers(self):
        return [_dp(self.signal), _dp(self.sig_off), _dp(self.reference), _dp(self.ref_off),
                _dp(self.context_before), _dp(self.cb_off), _dp(self.context_after), _dp(self.ca_off),
                _dp(self.anchors), _dp(self.anc_off)]

    def algorithmic_bytes_align(self, band_cells):
        """B_align summed over the batch: 20*C + 8*N + 8*A + 4*(R+ctx) + 24*R."""
        ctx = int(self.context_before.numel() + self.context_after.numel())
        return (20 * int(band_cells) + 8 * self.total_signal + 8 * self.total_anchors
                + 4 * (self.total_ref + ctx) + 24 * self.total_ref)

    def algorithmic_bytes_snp(self, band_cells):
        """B_snp: 32*C' + 8*N + 8*A + 4*(R+ctx) + 32*R."""
        ctx = int(self.context_before.numel() + self.context_after.numel())
        return (32 * int(band_cells) + 8 * self.total_signal + 8 * self.total_anchors
                + 4 * (self.total_ref + ctx) + 32 * self.total_ref)


def refine_alignment_dev(dbatch, bandwidth, min_event_length, kmer_model, model_transitions,
                         events=None, status=None):
    """Device in, device out: -> (events int32 (sum R, 2), status int32 (n,)) torch tensors."""
    torch = dbatch.torch
    lib = _lib.load()
    if events is None:
        events = torch.zeros((dbatch.total_ref, 2), dtype=torch.int32, device=dbatch.device)
    if status is None:
        status = torch.zeros(dbatch.n, dtype=torch.int32, device=dbatch.device)
    _lib.check(lib.nvk_refine_alignment_batch_dev(
        kmer_model.handle, dbatch.n, dbatch.total_signal, dbatch.total_ref, dbatch.total_anchors,
        *dbatch.pointers(), int(bandwidth), int(min_event_length), int(bool(model_transitions)),
        _dp(events), _dp(status)), 'nvk_refine_alignment_batch_dev')
    return events, status


def estimate_log_likelihoods_dev(dbatch, bandwidth, min_event_length, kmer_model, model_wobbling,
                                 ll=None, status=None):
    torch = dbatch.torch
    lib = _lib.load()
    if ll is None:
        ll = torch.zeros((dbatch.total_ref, kmer_model.alphabet_size), dtype=torch.float64, device=dbatch.device)
    if status is None:
        status = torch.zeros(dbatch.n, dtype=torch.int32, device=dbatch.device)
    _lib.check(lib.nvk_estimate_log_likelihoods_batch_dev(
        kmer_model.handle, dbatch.n, dbatch.total_signal, dbatch.total_ref, dbatch.total_anchors,
        *dbatch.pointers(), int(bandwidth), int(min_event_length), int(bool(model_wobbling)),
        _dp(ll), _dp(status)), 'nvk_estimate_log_likelihoods_batch_dev')
    return ll, status


# ---- host steps adjacent to the path, on the device (include/nadavca_hip.h) --------
def normalize_groups_dev(context, raw, grp_off, out=None):
    """``Read.normalize_reads`` for groups of samples laid end to end (torch f64 / int64 tensors on the
    context's device): -> (normalised signal, (n_groups, 2) tensor of (centre, scale))."""
    import torch
    lib = _lib.load()
    n_groups = int(grp_off.numel()) - 1
    if out is None:
        out = torch.empty_like(raw)
    cs = torch.zeros((max(n_groups, 0), 2), dtype=torch.float64, device=raw.device)
    _lib.check(lib.nvk_normalize_groups_dev(context.handle, n_groups, _dp(raw), _dp(grp_off), _dp(out),
                                            _dp(cs)), 'nvk_normalize_groups_dev')
    return out, cs


def select_hist_dev(context, x):
    """-> local_hist(mode, centre, prefix, pass) over a device tensor of samples, for distributed.pooled_median:
    256 counts per call (int64 cuda tensor, ready for the all-reduce), nvk_select_hist_dev."""
    import torch
    lib = _lib.load()

    def local_hist(mode, centre, prefix, p):
        h = torch.zeros(256, dtype=torch.int64, device=x.device)
        _lib.check(lib.nvk_select_hist_dev(context.handle, _dp(x), int(x.numel()), int(mode), float(centre),
                                           C.c_uint64(int(prefix)), int(p), _dp(h)), 'nvk_select_hist_dev')
        return h
    return local_hist


def normalize_apply_dev(context, x, centre, scale, out=None):
    """clip((x - centre) / scale, -5, 5) on the device (read.py:80-81 with a given shift and scale)."""
    import torch
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.nvk_normalize_apply_dev(context.handle, _dp(x), int(x.numel()), float(centre), float(scale),
                                           _dp(out)), 'nvk_normalize_apply_dev')
    return out


def expected_levels_dev(dbatch, kmer_model, with_contexts=True):
    """``KmerModel.get_expected_signal`` for every read of the batch -> f64 (sum R,).  Without contexts
    the k-mers at the ends are padded with base 0, as ``get_expected_signal(bases, [], [])`` does
    (align_signal.py:63)."""
    torch = dbatch.torch
    lib = _lib.load()
    out = torch.zeros(dbatch.total_ref, dtype=torch.float64, device=dbatch.device)
    if with_contexts:
        cb, cbo, ca, cao = dbatch.context_before, dbatch.cb_off, dbatch.context_after, dbatch.ca_off
    else:
        cb = ca = torch.zeros(1, dtype=torch.int32, device=dbatch.device)
        cbo = cao = torch.zeros(dbatch.n + 1, dtype=torch.int64, device=dbatch.device)
    _lib.check(lib.nvk_expected_signal_batch_dev(
        kmer_model.handle, dbatch.n, dbatch.total_ref, _dp(dbatch.reference), _dp(dbatch.ref_off),
        _dp(cb), _dp(cbo), _dp(ca), _dp(cao), _dp(out)), 'nvk_expected_signal_batch_dev')
    return out


def event_means_dev(dbatch, context, events, status=None):
    """Mean of the batch's signal over every event of ``events`` (as returned by
    ``refine_alignment_dev``) -> f64 (sum R,); equals ``numpy.mean`` of the same samples bit for bit."""
    torch = dbatch.torch
    lib = _lib.load()
    out = torch.zeros(dbatch.total_ref, dtype=torch.float64, device=dbatch.device)
    _lib.check(lib.nvk_event_means_dev(
        context.handle, dbatch.n, dbatch.total_ref, _dp(dbatch.signal), _dp(dbatch.sig_off), _dp(events),
        _dp(dbatch.ref_off), _dp(status) if status is not None else C.c_void_p(0), _dp(out)),
        'nvk_event_means_dev')
    return out


def linfit_rescale_dev(dbatch, context, expected, means, status=None):
    """Per read: least-squares line of ``means`` on ``expected`` and ``signal = (signal - intercept) /
    slope`` in place on the batch's signal -> (n, 2) tensor of (slope, intercept)."""
    torch = dbatch.torch
    lib = _lib.load()
    fit = torch.zeros((dbatch.n, 2), dtype=torch.float64, device=dbatch.device)
    _lib.check(lib.nvk_linfit_rescale_dev(
        context.handle, dbatch.n, _dp(expected), _dp(means), _dp(dbatch.ref_off),
        _dp(status) if status is not None else C.c_void_p(0), _dp(dbatch.signal), _dp(dbatch.sig_off),
        _dp(fit)), 'nvk_linfit_rescale_dev')
    return fit


def spline_fit_dev(context, means, expected, ref_off, status=None):
    """The fit of ``Read.tweak_signal_normalization`` (read.py:83-93: keep |expected - mean| <= 1, sort,
    ``splrep(..., s=len)``) for every read on the device.  means / expected: f64 device tensors, read j at
    [ref_off[j], ref_off[j+1]); status: int32 per read or None (reads with status != 0 are not fitted).
    -> (t (n, 8), c (n, 8), fit int32 (n,)): fit 0 = fitted (coefficients equal scipy's), 1 = fewer than 4 usable
    events, 2 = outside the polynomial case (include/nadavca_hip.h: nvk_spline_fit_dev)."""
    import torch
    lib = _lib.load()
    n = int(ref_off.numel()) - 1
    t = torch.empty((n, 8), dtype=torch.float64, device=means.device)
    c = torch.empty((n, 8), dtype=torch.float64, device=means.device)
    fit = torch.empty(n, dtype=torch.int32, device=means.device)
    _lib.check(lib.nvk_spline_fit_dev(
        context.handle, n, int(means.numel()), _dp(means), _dp(expected), _dp(ref_off),
        _dp(status) if status is not None else C.c_void_p(0), _dp(t), _dp(c), _dp(fit)), 'nvk_spline_fit_dev')
    return t, c, fit


def splev_groups_dev(context, x, grp_off, t, c, knot_off, degree, out=None):
    """``scipy.interpolate.splev`` for groups of samples laid end to end, one spline (its knots ``t`` and
    coefficients ``c`` between ``knot_off[g]`` and ``knot_off[g+1]``) per group -> values (torch f64)."""
    import torch
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.nvk_splev_groups_dev(context.handle, int(grp_off.numel()) - 1, _dp(x), _dp(grp_off), _dp(t),
                                        _dp(c), _dp(knot_off), int(degree), _dp(out)), 'nvk_splev_groups_dev')
    return out


def refine_renorm_loop_dev(dbatch, bandwidth, min_event_length, kmer_model, model_transitions,
                           renorm_rounds):
    """The renormalise / re-align loop of ``align_signal`` (align_signal.py:55-80) for a whole batch
    without leaving the device: align; then for round r = 0 .. renorm_rounds-1: even r — per-event
    means, linear re-fit against the model's expected levels (empty contexts), rescale the signal;
    odd r — align again.  -> (events, status, [fit tensors of the even rounds]).  ``dbatch.signal`` is
    rescaled in place."""
    context = kmer_model.context
    events, status = refine_alignment_dev(dbatch, bandwidth, min_event_length, kmer_model, model_transitions)
    expected, fits = None, []
    for r in range(int(renorm_rounds)):
        if r % 2 == 0:
            if expected is None:
                expected = expected_levels_dev(dbatch, kmer_model, with_contexts=False)
            means = event_means_dev(dbatch, context, events, status)
            fits.append(linfit_rescale_dev(dbatch, context, expected, means, status))
        else:
            # a read that lost its path stays lost (the reference raises on it at this point)
            new_events, new_status = refine_alignment_dev(dbatch, bandwidth, min_event_length, kmer_model,
                                                          model_transitions)
            keep = status != 0
            new_status[keep] = status[keep]
            events, status = new_events, new_status
    return events, status, fits


# ---- Chunk score accumulation and posterior, device-resident (estimator.py:199-236) ---------------------
def consensus_accumulate_dev(context, dbatch, ll, chunk_start, reverse, status, normalization_event_length,
                             ref_len, acc=None, cov=None):
    """Normalise, strand-correct and scatter-add the per-read log-likelihood rows ``ll`` (as written by
    ``estimate_log_likelihoods_dev``) into the per-position sums ``acc`` (ref_len, alphabet) f64 and the
    coverage ``cov`` (ref_len,) i64 — torch tensors on the batch's device, created zeroed when not given,
    accumulated into otherwise.  chunk_start i64 (n,), reverse i32 (n,), status i32 (n,) device tensors."""
    torch = dbatch.torch
    lib = _lib.load()
    alpha = int(ll.shape[1])
    if acc is None:
        acc = torch.zeros((int(ref_len), alpha), dtype=torch.float64, device=dbatch.device)
    if cov is None:
        cov = torch.zeros(int(ref_len), dtype=torch.int64, device=dbatch.device)
    _lib.check(lib.nvk_consensus_accumulate_dev(
        context.handle, dbatch.n, dbatch.total_ref, alpha, _dp(ll), _dp(dbatch.reference), _dp(dbatch.ref_off),
        _dp(chunk_start), _dp(reverse), _dp(status), float(normalization_event_length), int(ref_len),
        _dp(acc), _dp(cov)), 'nvk_consensus_accumulate_dev')
    return acc, cov


def posterior_segments_dev(context, ll, reference_num, seg_off, k, snp_prior, out=None):
    """``_compute_posterior`` for groups of positions laid end to end (device tensors: ll (len, alphabet) f64,
    reference_num (len,) i32, seg_off (n_segments + 1,) i64) -> posterior (len, alphabet) f64."""
    import torch
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(ll)
    _lib.check(lib.nvk_posterior_segments_dev(
        context.handle, int(ll.shape[0]), int(seg_off.numel()) - 1, _dp(seg_off), int(ll.shape[1]), int(k),
        float(snp_prior), _dp(ll), _dp(reference_num), _dp(out)), 'nvk_posterior_segments_dev')
    return out
