import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


def golden_cases(prefix='case_'):
    return sorted(f[:-3] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith('.pt'))


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
