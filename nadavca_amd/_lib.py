"""ctypes binding of nadavca_amd/csrc/libnadavca_hip.so (ABI: include/nadavca_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C nadavca_amd/csrc``.
There is no fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libnadavca_hip.so')

NVK_OK = 0
NVK_ERR_NO_DEVICE = -1
NVK_ERR_INVALID = -2
NVK_ERR_HIP = -3
NVK_ERR_UNSUPPORTED = -4
NVK_ERR_NOMEM = -5

READ_OK = 0
READ_NO_PATH = 1
READ_BAD_INPUT = -1
READ_BAD_BAND = -2
READ_TOO_WIDE = -3

TIE_EXACT = 1   # nvk_last_tie_flags: some path comparison of the read met two exactly equal scores
TIE_NEAR = 2    # ... two scores closer than 2^-24 relative (beyond the class below)
TIE_PLATEAU = 8  # structural mark: two adjacent bases with the same k-mer level (boundary unidentifiable)
TIE_ULP = 4     # ... two scores within 64 ulps of the reference's log value, not equal (where its rounding may decide)

K_PLAN, K_ALIGN, K_ELL_SWEEP, K_ELL_HYP, K_EXPECTED, K_CONSENSUS, K_POSTERIOR, K_RENORM = range(8)
KERNEL_NAMES = ['plan', 'align', 'ell_sweep', 'ell_hyp', 'expected', 'consensus', 'posterior', 'renorm']

_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_dbl = C.c_double

# every symbol include/nadavca_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'nvk_last_error': (C.c_char_p, []),
    'nvk_device_count': (_int, []),
    'nvk_ctx_create': (_int, [_int, C.POINTER(_vp)]),
    'nvk_ctx_destroy': (None, [_vp]),
    'nvk_ctx_synchronize': (_int, [_vp]),
    'nvk_ctx_stream': (_vp, [_vp]),
    'nvk_ctx_set_slots': (_int, [_vp, _int]),
    'nvk_timing_enable': (_int, [_vp, _int]),
    'nvk_timing_reset': (_int, [_vp]),
    'nvk_timing_read': (_int, [_vp, _int, C.POINTER(_dbl), C.POINTER(_i64)]),
    'nvk_last_batch_stats': (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'nvk_last_retry_count': (_int, [_vp, C.POINTER(_i64)]),
    'nvk_last_tie_count': (_int, [_vp, C.POINTER(_i64)]),
    'nvk_last_tie_counts': (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'nvk_last_tie_flags': (_int, [_vp, _i64, _vp]),
    'nvk_ctx_set_workspace_limit': (_int, [_vp, _i64]),
    'nvk_model_create': (_int, [_vp, _int, _int, _int, _vp, _vp, _i64, C.POINTER(_vp)]),
    'nvk_model_destroy': (None, [_vp]),
    'nvk_model_info': (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    'nvk_expected_signal_batch': (_int, [_vp, _i64] + [_vp] * 7),
    'nvk_expected_signal_batch_dev': (_int, [_vp, _i64, _i64] + [_vp] * 7),
    'nvk_refine_alignment_batch': (_int, [_vp, _i64] + [_vp] * 10 + [_int, _int, _int, _vp, _vp]),
    'nvk_refine_alignment_submit': (_int, [_vp, _i64] + [_vp] * 10 + [_int, _int, _int, _vp, _vp, _vp, C.POINTER(_i64)]),
    'nvk_refine_alignment_wait': (_int, [_vp, _i64]),
    'nvk_refine_alignment_batch_dev': (_int, [_vp, _i64, _i64, _i64, _i64] + [_vp] * 10 + [_int, _int, _int, _vp, _vp]),
    'nvk_estimate_log_likelihoods_batch': (_int, [_vp, _i64] + [_vp] * 10 + [_int, _int, _int, _vp, _vp]),
    'nvk_estimate_log_likelihoods_batch_dev': (_int, [_vp, _i64, _i64, _i64, _i64] + [_vp] * 10 + [_int, _int, _int, _vp, _vp]),
    'nvk_consensus_accumulate_dev': (_int, [_vp, _i64, _i64, _int] + [_vp] * 6 + [_dbl, _i64, _vp, _vp]),
    'nvk_posterior_dev': (_int, [_vp, _i64, _int, _int, _dbl, _vp, _vp, _vp]),
    'nvk_consensus_accumulate': (_int, [_vp, _i64, _int] + [_vp] * 6 + [_dbl, _i64, _vp, _vp]),
    'nvk_posterior_segments_dev': (_int, [_vp, _i64, _i64, _vp, _int, _int, _dbl, _vp, _vp, _vp]),
    'nvk_posterior': (_int, [_vp, _i64, _i64, _vp, _int, _int, _dbl, _vp, _vp, _vp]),
    'nvk_normalize_groups_dev': (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    'nvk_select_hist_dev': (_int, [_vp, _vp, _i64, _int, _dbl, C.c_uint64, _int, _vp]),
    'nvk_normalize_apply_dev': (_int, [_vp, _vp, _i64, _dbl, _dbl, _vp]),
    'nvk_event_means_dev': (_int, [_vp, _i64, _i64] + [_vp] * 6),
    'nvk_linfit_rescale_dev': (_int, [_vp, _i64] + [_vp] * 7),
    'nvk_splev_groups_dev': (_int, [_vp, _i64] + [_vp] * 5 + [_int, _vp]),
    'nvk_spline_fit_dev': (_int, [_vp, _i64, _i64] + [_vp] * 7),
}

_lib = None
_lock = threading.Lock()


class NadavcaHipError(RuntimeError):
    pass


def _torch_runtime_first():
    """PyTorch-ROCm ships its own copy of the HIP runtime; this library links the system one.  Both can
    live in one process only if torch's touches the device first — otherwise the first
    ``tensor.to('cuda')`` after a kernel call of this library fails with "No HIP GPUs are available".
    So torch's runtime is initialised before this library makes its first HIP call (torch is the
    package's plumbing for device memory anyway, nadavca_amd/device.py)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch, or no device: this library reports the latter itself
        pass


def load():
    """Load libnadavca_hip.so (once) and attach the prototypes.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        _torch_runtime_first()
        if not os.path.isfile(LIB_PATH):
            raise NadavcaHipError(
                'HIP library not built: %s is missing (run __graft_entry__.build() or '
                '`make -C nadavca_amd/csrc`). There is no CPU fallback.' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc, what):
    if rc != NVK_OK:
        lib = load()
        msg = lib.nvk_last_error()
        msg = msg.decode('utf-8', 'replace') if msg else ''
        if rc == NVK_ERR_INVALID:
            raise ValueError('%s: %s' % (what, msg))
        raise NadavcaHipError('%s failed (%d): %s' % (what, rc, msg))


class Context:
    """One per process/GPU: owns the HIP stream, the workspaces and the timers."""

    def __init__(self, device=0):
        lib = load()
        self._lib = lib
        h = _vp()
        check(lib.nvk_ctx_create(int(device), C.byref(h)), 'nvk_ctx_create')
        self.handle = h
        self.device = int(device)

    def synchronize(self):
        check(self._lib.nvk_ctx_synchronize(self.handle), 'nvk_ctx_synchronize')

    def set_slots(self, slots):
        check(self._lib.nvk_ctx_set_slots(self.handle, int(slots)), 'nvk_ctx_set_slots')

    def timing_enable(self, on=True):
        check(self._lib.nvk_timing_enable(self.handle, int(bool(on))), 'nvk_timing_enable')

    def timing_reset(self):
        check(self._lib.nvk_timing_reset(self.handle), 'nvk_timing_reset')

    def timing_read(self):
        out = {}
        for kid, name in enumerate(KERNEL_NAMES):
            ms, n = _dbl(), _i64()
            check(self._lib.nvk_timing_read(self.handle, kid, C.byref(ms), C.byref(n)), 'nvk_timing_read')
            out[name] = (ms.value, n.value)
        return out

    def last_batch_stats(self):
        a, b, c = _i64(), _i64(), _i64()
        check(self._lib.nvk_last_batch_stats(self.handle, C.byref(a), C.byref(b), C.byref(c)),
              'nvk_last_batch_stats')
        d = _i64()
        check(self._lib.nvk_last_retry_count(self.handle, C.byref(d)), 'nvk_last_retry_count')
        e = _i64()
        check(self._lib.nvk_last_tie_count(self.handle, C.byref(e)), 'nvk_last_tie_count')
        f, g, h = _i64(), _i64(), _i64()
        check(self._lib.nvk_last_tie_counts(self.handle, C.byref(f), C.byref(g), C.byref(h)), 'nvk_last_tie_counts')
        return dict(band_cells=a.value, wave_steps=b.value, spill_bytes=c.value, reads_redone_exact=d.value,
                    reads_tie_ambiguous=e.value, reads_tie_exact=f.value, reads_tie_near=g.value,
                    reads_tie_ulp=h.value)

    def last_tie_flags(self, n_reads):
        """Per read of the last refine_alignment batch: TIE_EXACT | TIE_NEAR | TIE_ULP where a path comparison fell
        inside the tie margin (include/nadavca_hip.h, parity contract)."""
        import numpy as np
        out = np.zeros(int(n_reads), dtype=np.int32)
        check(self._lib.nvk_last_tie_flags(self.handle, int(n_reads), _vp(out.ctypes.data)), 'nvk_last_tie_flags')
        return out

    def set_workspace_limit(self, nbytes):
        check(self._lib.nvk_ctx_set_workspace_limit(self.handle, int(nbytes)), 'nvk_ctx_set_workspace_limit')

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.nvk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    """Process-wide context for ``device`` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
        lib = load()
        n = lib.nvk_device_count()
        if n > 0:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
