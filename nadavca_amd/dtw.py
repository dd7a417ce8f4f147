"""``nadavca_amd.dtw`` — the drop-in for the reference's pybind11 module ``nadavca.dtw``
(/root/reference/nadavca/dtw/dtwmodule.cpp:10-29), backed by HIP kernels for gfx950.

Per-read surface (same names, keywords and return shapes as the reference):
    KmerModel(k, central_position, alphabet_size, mean, sigma)
        .get_k() .get_central_position() .get_expected_signal(reference, context_before, context_after)
    refine_alignment(signal, reference, context_before, context_after, approximate_alignment,
                     bandwidth, min_event_length, kmer_model, model_transitions) -> R x 2 ints, or []
    estimate_log_likelihoods(..., model_wobbling) -> R x alphabet floats

Batched surface (what the estimator uses; one launch for many reads):
    refine_alignment_batch(reads, ...)            reads = list of per-read argument tuples
    estimate_log_likelihoods_batch(reads, ...)

All compute goes through libnadavca_hip.so; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import READ_OK, READ_NO_PATH, READ_BAD_INPUT, READ_BAD_BAND, READ_TOO_WIDE  # noqa: F401


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _i32(a):
    a = np.asarray(a)
    if a.size == 0:
        return np.zeros(0, dtype=np.int32)
    return np.ascontiguousarray(a.reshape(-1), dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    return off


class FlatBatch:
    """Host-side flat batch (layout of include/nadavca_hip.h)."""

    def __init__(self, reads):
        """reads: iterable of (signal, reference, context_before, context_after, approximate_alignment)."""
        sig, ref, cb, ca, anc = [], [], [], [], []
        for s, r, b, a, al in reads:
            sig.append(_f64(s))
            ref.append(_i32(r))
            cb.append(_i32(b))
            ca.append(_i32(a))
            al = _i32(al)
            if al.size % 2:
                raise ValueError('approximate_alignment must have shape (A, 2)')
            anc.append(al)
        self.n = len(sig)
        cat = lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dtype=dt)
        self.signal = cat(sig, np.float64)
        self.reference = cat(ref, np.int32)
        self.context_before = cat(cb, np.int32)
        self.context_after = cat(ca, np.int32)
        self.anchors = cat(anc, np.int32)
        self.sig_off = _offsets([x.size for x in sig])
        self.ref_off = _offsets([x.size for x in ref])
        self.cb_off = _offsets([x.size for x in cb])
        self.ca_off = _offsets([x.size for x in ca])
        self.anc_off = _offsets([x.size // 2 for x in anc])

    @classmethod
    def from_arrays(cls, signal, sig_off, reference, ref_off, context_before, cb_off, context_after,
                    ca_off, anchors, anc_off):
        self = cls.__new__(cls)
        self.n = len(sig_off) - 1
        self.signal = _f64(signal)
        self.reference = _i32(reference)
        self.context_before = _i32(context_before)
        self.context_after = _i32(context_after)
        self.anchors = _i32(anchors)
        for name, v in (('sig_off', sig_off), ('ref_off', ref_off), ('cb_off', cb_off),
                        ('ca_off', ca_off), ('anc_off', anc_off)):
            setattr(self, name, np.ascontiguousarray(v, dtype=np.int64))
        return self

    def pointers(self):
        return [_ptr(self.signal), _ptr(self.sig_off), _ptr(self.reference), _ptr(self.ref_off),
                _ptr(self.context_before), _ptr(self.cb_off), _ptr(self.context_after), _ptr(self.ca_off),
                _ptr(self.anchors), _ptr(self.anc_off)]


class KmerModel:
    """Device-resident k-mer table (reference: dtw.KmerModel, kmer_model.cpp:6-14)."""

    def __init__(self, k, central_position, alphabet_size, mean, sigma, context=None):
        self._lib = _lib.load()
        self.context = context or _lib.default_context()
        mean, sigma = _f64(mean), _f64(sigma)
        if mean.size != sigma.size:
            raise ValueError('mean and sigma differ in length')
        self.k, self.central_position, self.alphabet_size = int(k), int(central_position), int(alphabet_size)
        h = C.c_void_p()
        _lib.check(self._lib.nvk_model_create(self.context.handle, self.k, self.central_position,
                                              self.alphabet_size, _ptr(mean), _ptr(sigma), mean.size,
                                              C.byref(h)), 'nvk_model_create')
        self.handle = h
        self.mean, self.sigma = mean, sigma

    def get_k(self):
        return self.k

    def get_central_position(self):
        return self.central_position

    def get_alphabet_size(self):
        return self.alphabet_size

    def get_expected_signal(self, reference, context_before, context_after):
        out = self.get_expected_signal_batch([(reference, context_before, context_after)])
        return out[0]

    def get_expected_signal_batch(self, items):
        """items: list of (reference, context_before, context_after) -> list of f64 arrays."""
        ref = [_i32(r) for r, _, _ in items]
        cb = [_i32(b) for _, b, _ in items]
        ca = [_i32(a) for _, _, a in items]
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int32)
        r, b, a = cat(ref), cat(cb), cat(ca)
        for name, v in (('reference', r), ('context_before', b), ('context_after', a)):
            if v.size and (int(v.min()) < 0 or int(v.max()) >= self.alphabet_size):
                raise ValueError('get_expected_signal: %s holds a base code outside 0..%d'
                                 % (name, self.alphabet_size - 1))
        ro, bo, ao = _offsets([x.size for x in ref]), _offsets([x.size for x in cb]), _offsets([x.size for x in ca])
        out = np.zeros(r.size, dtype=np.float64)
        _lib.check(self._lib.nvk_expected_signal_batch(self.handle, len(items), _ptr(r), _ptr(ro), _ptr(b),
                                                       _ptr(bo), _ptr(a), _ptr(ao), _ptr(out)),
                   'nvk_expected_signal_batch')
        return [out[ro[i]:ro[i + 1]] for i in range(len(items))]

    @staticmethod
    def load_from_npz(filename, context=None):
        z = np.load(filename)
        return KmerModel(int(z['k']), int(z['central_pos']), int(z['alphabet_size']), z['mean'], z['sigma'],
                         context=context)

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.nvk_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# batched operators
# ------------------------------------------------------------------------------------------------
def _raise_on_status(what, status):
    """Per-read failures as exceptions (the per-read operators of the reference raise too): invalid input
    -> ValueError; a band wider than the compiled kernels serve -> NadavcaHipError."""
    bad = np.nonzero((status < 0) & (status != READ_TOO_WIDE))[0]
    if bad.size:
        raise ValueError('%s: invalid input for read(s) %s (status %s)'
                         % (what, bad[:8].tolist(), status[bad[:8]].tolist()))
    wide = np.nonzero(status == READ_TOO_WIDE)[0]
    if wide.size:
        raise _lib.NadavcaHipError('%s: the band of read(s) %s is wider than the compiled kernels serve '
                                   '(INTEGRATION.md, limits)' % (what, wide[:8].tolist()))


def refine_alignment_flat(batch, bandwidth, min_event_length, kmer_model, model_transitions, on_error='raise'):
    """-> (events int32 (sum R, 2), status int32 (n,)) for a FlatBatch.  on_error='status': per-read
    failures (status < 0) are left to the caller instead of raised."""
    lib = _lib.load()
    events = np.zeros((int(batch.ref_off[-1]), 2), dtype=np.int32)
    status = np.zeros(batch.n, dtype=np.int32)
    _lib.check(lib.nvk_refine_alignment_batch(kmer_model.handle, batch.n, *batch.pointers(), int(bandwidth),
                                              int(min_event_length), int(bool(model_transitions)),
                                              _ptr(events), _ptr(status)), 'nvk_refine_alignment_batch')
    if on_error == 'raise':
        _raise_on_status('refine_alignment', status)
    return events, status


class RefineStream:
    """``refine_alignment`` over a stream of FlatBatches with the PCIe copies behind the kernels
    (nvk_refine_alignment_submit / _wait, include/nadavca_hip.h): ``submit`` uploads a batch and returns at once
    while the previous batch's kernels still run, ``wait`` hands back (events, status, tie flags) of a ticket.
    Keep at most a few batches in flight (one lane each; the library has three)."""

    def __init__(self, kmer_model, bandwidth, min_event_length, model_transitions):
        self._lib = _lib.load()
        self.kmer_model = kmer_model
        self.args = (int(bandwidth), int(min_event_length), int(bool(model_transitions)))
        self._pending = {}

    def submit(self, batch, out=None):
        """``out``: (events int32 (sum R, 2), status int32 (n,), tie flags int32 (n,)) arrays to reuse — fresh
        arrays are page-faulted in while the results arrive, which shows in a tight loop."""
        if out is None:
            out = (np.zeros((int(batch.ref_off[-1]), 2), dtype=np.int32), np.zeros(batch.n, dtype=np.int32),
                   np.zeros(batch.n, dtype=np.int32))
        events, status, ties = out
        assert events.size == 2 * int(batch.ref_off[-1]) and status.size == batch.n and ties.size == batch.n
        t = C.c_int64(-1)
        _lib.check(self._lib.nvk_refine_alignment_submit(self.kmer_model.handle, batch.n, *batch.pointers(),
                                                         *self.args, _ptr(events), _ptr(status), _ptr(ties),
                                                         C.byref(t)), 'nvk_refine_alignment_submit')
        self._pending[t.value] = (batch, events, status, ties)   # (keeps the host arrays alive until wait)
        return t.value

    def wait(self, ticket):
        batch, events, status, ties = self._pending.pop(ticket)
        _lib.check(self._lib.nvk_refine_alignment_wait(self.kmer_model.handle, int(ticket)),
                   'nvk_refine_alignment_wait')
        return events, status, ties


def estimate_log_likelihoods_flat(batch, bandwidth, min_event_length, kmer_model, model_wobbling,
                                  on_error='raise'):
    """-> (ll f64 (sum R, alphabet), status int32 (n,)) for a FlatBatch."""
    lib = _lib.load()
    alpha = kmer_model.alphabet_size
    ll = np.zeros((int(batch.ref_off[-1]), alpha), dtype=np.float64)
    status = np.zeros(batch.n, dtype=np.int32)
    _lib.check(lib.nvk_estimate_log_likelihoods_batch(kmer_model.handle, batch.n, *batch.pointers(),
                                                      int(bandwidth), int(min_event_length),
                                                      int(bool(model_wobbling)), _ptr(ll), _ptr(status)),
               'nvk_estimate_log_likelihoods_batch')
    if on_error == 'raise':
        _raise_on_status('estimate_log_likelihoods', status)
    return ll, status


def refine_alignment_batch(reads, bandwidth, min_event_length, kmer_model, model_transitions,
                           on_error='raise', return_status=False):
    """reads: list of (signal, reference, context_before, context_after, approximate_alignment).
    -> list of (R, 2) int arrays; an empty (0, 2) array where the band holds no valid path (and, with
    on_error='status', where the read was refused: return_status=True also returns the status array)."""
    batch = reads if isinstance(reads, FlatBatch) else FlatBatch(reads)
    events, status = refine_alignment_flat(batch, bandwidth, min_event_length, kmer_model, model_transitions,
                                           on_error=on_error)
    out = []
    for j in range(batch.n):
        if status[j] == READ_OK:
            out.append(events[batch.ref_off[j]:batch.ref_off[j + 1]])
        else:
            out.append(np.zeros((0, 2), dtype=np.int32))
    return (out, status) if return_status else out


def estimate_log_likelihoods_batch(reads, bandwidth, min_event_length, kmer_model, model_wobbling):
    batch = reads if isinstance(reads, FlatBatch) else FlatBatch(reads)
    ll, _ = estimate_log_likelihoods_flat(batch, bandwidth, min_event_length, kmer_model, model_wobbling)
    return [ll[batch.ref_off[j]:batch.ref_off[j + 1]] for j in range(batch.n)]


# ------------------------------------------------------------------------------------------------
# per-read operators (reference signatures)
# ------------------------------------------------------------------------------------------------
def refine_alignment(signal, reference, context_before, context_after, approximate_alignment, bandwidth,
                     min_event_length, kmer_model, model_transitions):
    return refine_alignment_batch([(signal, reference, context_before, context_after, approximate_alignment)],
                                  bandwidth, min_event_length, kmer_model, model_transitions)[0]


def estimate_log_likelihoods(signal, reference, context_before, context_after, approximate_alignment,
                             bandwidth, min_event_length, kmer_model, model_wobbling):
    return estimate_log_likelihoods_batch(
        [(signal, reference, context_before, context_after, approximate_alignment)],
        bandwidth, min_event_length, kmer_model, model_wobbling)[0]
