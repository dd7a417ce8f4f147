"""nadavca_amd — MI355X-native engine for Nadavca's signal-to-reference alignment path.

Public surface mirrors the reference package (/root/reference/nadavca/__init__.py:1-2):
``align_signal`` and ``estimate_snps``, plus the ``dtw`` operator module.  Importing the package
does not touch the GPU; every compute entry point goes through the HIP library
(nadavca_amd/csrc/libnadavca_hip.so) and fails loudly if it or the device is missing.
"""
from . import dtw  # noqa: F401
from .estimate_snps import estimate_snps, estimate_snps_batch  # noqa: F401
from .align_signal import align_signal, align_signal_batch  # noqa: F401

__all__ = ['align_signal', 'align_signal_batch', 'estimate_snps', 'estimate_snps_batch', 'dtw']
