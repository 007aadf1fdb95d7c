import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built library: on a GPU-less machine a plain `pytest tests`
    skips them instead of erroring in fixture setup (the product itself has no CPU fallback)."""
    import torch
    lib = os.path.join(ROOT, 'spotlight_amd', 'csrc', 'libspotlight_hip.so')
    if torch.cuda.is_available() and os.path.exists(lib):
        return
    why = 'needs an MI355X + libspotlight_hip.so (run with gpurun)'
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(pytest.mark.skip(reason=why))


GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def golden_names(kind='bilinear'):
    """Fixtures recorded from the live reference: 'bilinear' = plain BilinearNet runs
    (oracle/make_golden.py), 'seq' = ImplicitSequenceModel/PoolNet (oracle/make_golden_seq.py),
    'bloom' = BilinearNet with BloomEmbedding layers (oracle/make_golden_bloom.py), 'host' = outputs of
    the host-side callers (oracle/make_golden_host.py), 'explicit' = ExplicitFactorizationModel runs
    (oracle/make_golden_explicit.py), 'enc' = ImplicitSequenceModel with the LSTM / CNN / mixture encoders
    (oracle/make_golden_encoders.py)."""
    def kind_of(f):
        if f.startswith('host_'):
            return 'host'
        if f.startswith('enc_'):
            return 'enc'
        if f.startswith('explicit_'):
            return 'explicit'
        return 'seq' if f.startswith('seq_') else 'bloom' if f.startswith('bloom_') else 'bilinear'
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith('.npz') and kind_of(f) == kind)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
