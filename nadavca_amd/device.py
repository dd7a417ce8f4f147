"""Device-resident batches: torch tensors used purely as HBM buffers for the ``*_dev`` entry
points of the C-ABI (inputs already in HBM when the call starts; see include/nadavca_hip.h)."""
import ctypes as C

import numpy as np

from . import _lib


def _dp(t):
    return C.c_void_p(t.data_ptr())


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
