"""nadavca_amd — MI355X-native engine for Nadavca's signal-to-reference alignment path.

Public surface mirrors the reference package (/root/reference/nadavca/__init__.py:1-2):
``align_signal`` and ``estimate_snps``, plus the ``dtw`` operator module.  Imports are
lazy so that host-only helpers (synthetic data, alphabet, config) work without a GPU;
every compute entry point goes through the HIP library and fails loudly if it or the
device is missing.
"""

__all__ = ['align_signal', 'estimate_snps', 'dtw']


def __getattr__(name):
    if name == 'align_signal':
        from .align_signal import align_signal
        return align_signal
    if name == 'estimate_snps':
        from .estimate_snps import estimate_snps
        return estimate_snps
    if name == 'dtw':
        import importlib
        return importlib.import_module('.dtw', __name__)
    raise AttributeError(name)
