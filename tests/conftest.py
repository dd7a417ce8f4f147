import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class GoldenFile:
    """tests/golden/dp_*.npz written by oracle/make_golden.py (reference outputs)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.model = (int(z['model_k']), int(z['model_central']), int(z['model_alphabet']),
                      np.array(z['model_mean']), np.array(z['model_sigma']))
        self.note = json.loads(str(z['note']))
        n = int(z['n_cases'])
        self.cases = []
        for i in range(n):
            pre = 'c%d_' % i
            self.cases.append({k[len(pre):]: np.array(z[k]) for k in z.files if k.startswith(pre)})


@pytest.fixture(scope='session')
def golden_tiny():
    return GoldenFile('dp_tiny.npz')


@pytest.fixture(scope='session')
def golden_config():
    return GoldenFile('dp_config.npz')


@pytest.fixture(scope='session')
def golden_nopath():
    return GoldenFile('dp_nopath.npz')


@pytest.fixture(scope='session')
def oracle_port():
    from oracle.oracle import Oracle, build
    build(with_reference=False)
    return Oracle('port')


def dp_args(case):
    return (case['signal'], case['reference'], case['context_before'], case['context_after'],
            case['approximate_alignment'], int(case['bandwidth']), int(case['min_event_length']))
