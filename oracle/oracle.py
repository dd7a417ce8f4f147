"""TEST INFRASTRUCTURE ONLY — ctypes front-ends for the CPU oracle.

Two back-ends with one Python surface (mirroring the reference's ``nadavca.dtw``
module, /root/reference/nadavca/dtw/dtwmodule.cpp:10-29):

* ``Oracle("port")``      -> oracle/liboracle.so      (C restatement, nadavca_oracle.c)
* ``Oracle("reference")`` -> oracle/_ref/libnadavca_ref.so (the reference's own
  sources compiled in place by ``make -C oracle ref``; present only where that
  build was possible)

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  The product package (nadavca_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libnadavca_ref.so")

_p_f64 = C.POINTER(C.c_double)
_p_i32 = C.POINTER(C.c_int32)


def build(with_reference=True):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    if with_reference and os.path.isdir("/root/reference/nadavca/dtw"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


def have_reference():
    return os.path.isfile(REF_LIB)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a).reshape(-1) if np.size(a) else np.zeros(0), dtype=np.int32)


class Oracle:
    def __init__(self, kind="port"):
        self.kind = kind
        if kind == "port":
            if not os.path.isfile(PORT_LIB):
                build(with_reference=False)
            self.lib = C.CDLL(PORT_LIB)
            self.pfx = "orc_"
        elif kind == "reference":
            if not os.path.isfile(REF_LIB):
                raise FileNotFoundError(REF_LIB)
            self.lib = C.CDLL(REF_LIB)
            self.pfx = "ref_"
        else:
            raise ValueError(kind)
        L, p = self.lib, self.pfx
        self._create = getattr(L, p + "model_create")
        self._create.restype = C.c_void_p
        self._create.argtypes = [C.c_int, C.c_int, C.c_int, _p_f64, _p_f64, C.c_int64]
        self._destroy = getattr(L, p + "model_destroy")
        self._destroy.restype = None
        self._destroy.argtypes = [C.c_void_p]
        self._expected = getattr(L, p + "expected_signal")
        self._expected.restype = None
        self._expected.argtypes = [C.c_void_p, _p_i32, C.c_int64, _p_i32, C.c_int64, _p_i32, C.c_int64, _p_f64]
        common = [C.c_void_p, _p_f64, C.c_int64, _p_i32, C.c_int64, _p_i32, C.c_int64, _p_i32, C.c_int64,
                  _p_i32, C.c_int64, C.c_int, C.c_int, C.c_int]
        self._refine = getattr(L, p + "refine_alignment")
        self._refine.restype = C.c_int
        self._refine.argtypes = common + [_p_i32]
        self._ell = getattr(L, p + "estimate_log_likelihoods")
        self._ell.restype = None
        self._ell.argtypes = common + [_p_f64]
        if kind == "port":
            self._bands = L.orc_bands
            self._bands.restype = None
            self._bands.argtypes = [_p_i32, C.c_int64, C.c_int64, C.c_int64, C.c_int, _p_i32, _p_i32]

    # -- model ---------------------------------------------------------------
    def KmerModel(self, k, central_position, alphabet_size, mean, sigma):
        return OracleKmerModel(self, k, central_position, alphabet_size, mean, sigma)

    # -- ops -----------------------------------------------------------------
    def _args(self, signal, reference, context_before, context_after, approximate_alignment):
        sig = _f64(signal)
        ref = _i32(reference)
        cb = _i32(context_before)
        ca = _i32(context_after)
        anc = _i32(approximate_alignment)
        keep = (sig, ref, cb, ca, anc)
        return keep, [sig.ctypes.data_as(_p_f64), sig.size, ref.ctypes.data_as(_p_i32), ref.size,
                      cb.ctypes.data_as(_p_i32), cb.size, ca.ctypes.data_as(_p_i32), ca.size,
                      anc.ctypes.data_as(_p_i32), anc.size // 2]

    def refine_alignment(self, signal, reference, context_before, context_after, approximate_alignment,
                         bandwidth, min_event_length, kmer_model, model_transitions):
        keep, a = self._args(signal, reference, context_before, context_after, approximate_alignment)
        R = keep[1].size
        out = np.zeros((R, 2), dtype=np.int32)
        st = self._refine(kmer_model.handle, *a, int(bandwidth), int(min_event_length),
                          int(bool(model_transitions)), out.ctypes.data_as(_p_i32))
        if st != 0:
            return np.zeros((0, 2), dtype=np.int32)
        return out

    def estimate_log_likelihoods(self, signal, reference, context_before, context_after,
                                 approximate_alignment, bandwidth, min_event_length, kmer_model,
                                 model_wobbling):
        keep, a = self._args(signal, reference, context_before, context_after, approximate_alignment)
        R = keep[1].size
        out = np.zeros((R, kmer_model.alphabet_size), dtype=np.float64)
        self._ell(kmer_model.handle, *a, int(bandwidth), int(min_event_length),
                  int(bool(model_wobbling)), out.ctypes.data_as(_p_f64))
        return out

    def bands(self, approximate_alignment, R, N, bandwidth):
        anc = _i32(approximate_alignment)
        bs = np.zeros(R + 1, dtype=np.int32)
        be = np.zeros(R + 1, dtype=np.int32)
        self._bands(anc.ctypes.data_as(_p_i32), anc.size // 2, R, N, int(bandwidth),
                    bs.ctypes.data_as(_p_i32), be.ctypes.data_as(_p_i32))
        return bs, be


class OracleKmerModel:
    def __init__(self, oracle, k, central_position, alphabet_size, mean, sigma):
        self.oracle = oracle
        self.k, self.central_position, self.alphabet_size = int(k), int(central_position), int(alphabet_size)
        mean, sigma = _f64(mean), _f64(sigma)
        assert mean.size == sigma.size
        self.handle = oracle._create(self.k, self.central_position, self.alphabet_size,
                                     mean.ctypes.data_as(_p_f64), sigma.ctypes.data_as(_p_f64), mean.size)

    def get_k(self):
        return self.k

    def get_central_position(self):
        return self.central_position

    def get_expected_signal(self, reference, context_before, context_after):
        ref, cb, ca = _i32(reference), _i32(context_before), _i32(context_after)
        out = np.zeros(ref.size, dtype=np.float64)
        self.oracle._expected(self.handle, ref.ctypes.data_as(_p_i32), ref.size, cb.ctypes.data_as(_p_i32),
                              cb.size, ca.ctypes.data_as(_p_i32), ca.size, out.ctypes.data_as(_p_f64))
        return out

    def __del__(self):
        try:
            self.oracle._destroy(self.handle)
        except Exception:
            pass


LD_LIB = os.path.join(_HERE, "liboracle_ld.so")


class LongDoubleReferee:
    """refine_alignment of the C restatement compiled with every ``double`` an 80-bit ``long double``
    (``make -C oracle liboracle_ld.so``).  Used where the engine and the double-precision reference
    disagree on a row: the reference decides such rows by the rounding of its log-domain doubles
    (DESIGN.md 2.1); the same algorithm with 11 more mantissa bits says which answer the mathematics
    supports."""

    def __init__(self, k, central_position, alphabet_size, mean, sigma):
        if not os.path.isfile(LD_LIB):
            subprocess.run(["make", "-C", _HERE, "liboracle_ld.so"], check=True, capture_output=True)
        self.lib = C.CDLL(LD_LIB)
        self._pld = C.POINTER(C.c_longdouble)
        self.lib.orc_model_create.restype = C.c_void_p
        self.lib.orc_model_create.argtypes = [C.c_int, C.c_int, C.c_int, self._pld, self._pld, C.c_int64]
        self.lib.orc_refine_alignment.restype = C.c_int
        self.lib.orc_refine_alignment.argtypes = [C.c_void_p, self._pld, C.c_int64, _p_i32, C.c_int64, _p_i32,
                                                  C.c_int64, _p_i32, C.c_int64, _p_i32, C.c_int64, C.c_int,
                                                  C.c_int, C.c_int, _p_i32]
        mean = np.ascontiguousarray(mean, dtype=np.longdouble)
        sigma = np.ascontiguousarray(sigma, dtype=np.longdouble)
        self.handle = self.lib.orc_model_create(int(k), int(central_position), int(alphabet_size),
                                                mean.ctypes.data_as(self._pld), sigma.ctypes.data_as(self._pld),
                                                mean.size)

    def refine_alignment(self, signal, reference, context_before, context_after, approximate_alignment,
                         bandwidth, min_event_length, model_transitions):
        sig = np.ascontiguousarray(signal, dtype=np.longdouble)
        ref, cb, ca, anc = _i32(reference), _i32(context_before), _i32(context_after), _i32(approximate_alignment)
        out = np.zeros((ref.size, 2), dtype=np.int32)
        st = self.lib.orc_refine_alignment(self.handle, sig.ctypes.data_as(self._pld), sig.size,
                                           ref.ctypes.data_as(_p_i32), ref.size, cb.ctypes.data_as(_p_i32), cb.size,
                                           ca.ctypes.data_as(_p_i32), ca.size, anc.ctypes.data_as(_p_i32),
                                           anc.size // 2, int(bandwidth), int(min_event_length),
                                           int(bool(model_transitions)), out.ctypes.data_as(_p_i32))
        return out if st == 0 else np.zeros((0, 2), dtype=np.int32)
