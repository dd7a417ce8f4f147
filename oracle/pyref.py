"""TEST INFRASTRUCTURE ONLY (build container only) — import the reference's Python layer from
/root/reference to generate estimator-level golden vectors (oracle/make_golden_estimator.py).

The reference package does not import as-is here: h5py and simplesam are not installed (neither
is used on the path we exercise), it uses the removed aliases numpy.int / numpy.float, and its
compiled extension is not built in its tree.  This loader (recipe recorded in SURVEY.md §8c):
  * restores numpy.int / numpy.float,
  * registers empty placeholder modules for h5py / simplesam (never called),
  * creates a package object `nadavca` whose __path__ is the reference directory (its __init__
    is skipped) and registers oracle/_ref/dtw*.so — the reference's own pybind11 module compiled
    unmodified by `make -C oracle ref` — as `nadavca.dtw`.
Nothing from the reference is copied; the modules are executed where they lie.
"""
import glob
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy

REF = '/root/reference'
_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    if 'nadavca.estimator' in sys.modules and getattr(sys.modules['nadavca'], '_is_reference', False):
        return sys.modules['nadavca']
    if not os.path.isdir(os.path.join(REF, 'nadavca')):
        raise RuntimeError('reference not present at %s' % REF)
    so = glob.glob(os.path.join(_HERE, '_ref', 'dtw*.so'))
    if not so:
        raise RuntimeError('oracle/_ref/dtw*.so missing: run `make -C oracle ref`')
    if not hasattr(numpy, 'int'):
        numpy.int = int
    if not hasattr(numpy, 'float'):
        numpy.float = float
    for name in ('h5py', 'simplesam'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    pkg = types.ModuleType('nadavca')
    pkg.__path__ = [os.path.join(REF, 'nadavca')]
    pkg.__file__ = os.path.join(REF, 'nadavca', '__init__.py')  # defaults.py derives paths from it
    pkg._is_reference = True
    sys.modules['nadavca'] = pkg
    spec = importlib.util.spec_from_file_location('nadavca.dtw', so[0])
    dtw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dtw)
    sys.modules['nadavca.dtw'] = dtw
    pkg.dtw = dtw
    for sub in ('alphabet', 'genome', 'read', 'alignment', 'estimator'):
        importlib.import_module('nadavca.' + sub)
    return pkg
