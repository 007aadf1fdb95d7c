import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def golden_names(sequence=False):
    """Fixtures recorded from the live reference: BilinearNet runs (oracle/make_golden.py) or,
    with sequence=True, ImplicitSequenceModel/PoolNet runs (oracle/make_golden_seq.py)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith('.npz') and f.startswith('seq_') == bool(sequence))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
