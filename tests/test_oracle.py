"""The CPU oracle against the committed golden vectors recorded from the live reference
(oracle/make_golden.py) and against numpy / scikit-learn for the third-party streams."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_names
from oracle.oracle import Rng, bloom_indices, murmur3_32
from oracle.replay import (case_from_rec, replay_bloom_with_oracle, replay_explicit_with_oracle, replay_seq_with_oracle,
                           replay_with_oracle)


# Open-loop drift of the final tables after the recorded multi-epoch runs, as a NORM (||oracle - reference||_2 / ||reference||_2;
# oracle/replay.py::frac_outside): no outlier quota.  Bounds = the engine tests' (tests/engine_checks.py); measured over the
# fixtures: embedding tables <= 2.6e-4, bias tables <= 6.7e-2 (a PoolNet bias table of a few hundred near-zero elements).
DRIFT_ROWS, DRIFT_BIAS = 5e-3, 0.2


def assert_drift(drift, bias_keys=('final_2', 'final_3')):
    for k, v in drift.items():
        assert v <= (DRIFT_BIAS if k in bias_keys else DRIFT_ROWS), (k, v, drift)


@pytest.mark.parametrize('name', golden_names())
def test_oracle_replays_reference_run(name):
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    if case.get('no_oracle'):
        pytest.skip('recorded for the autograd route (an optimizer the C oracle does not restate): tests/test_host_model.py')
    errs, frac = replay_with_oracle(case, rec)  # asserts bit-exact shuffles/negatives/rng state
    step = max(v for k, v in errs.items() if k.startswith('grad0') or k == 'loss')
    assert step < 1e-5, errs          # north-star tolerance on identical minibatches
    assert errs['loss'] < 1e-4
    assert_drift(frac)


@pytest.mark.parametrize('name', golden_names('seq'))
def test_oracle_replays_reference_sequence_run(name):
    """PoolNet / ImplicitSequenceModel (sequence/implicit.py:193-340) fixtures."""
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    errs, frac = replay_seq_with_oracle(case, rec)  # asserts bit-exact shuffles/negatives/rng state
    step = max(v for k, v in errs.items() if k.startswith('grad0') or k == 'loss0')
    assert step < 1e-5, errs
    assert errs['loss'] < 1e-3 and errs['predict_all'] < 1e-5 and errs['predict_some'] < 1e-5
    assert_drift(frac, bias_keys=('final_1',))


@pytest.mark.parametrize('name', golden_names('bloom'))
def test_oracle_replays_reference_bloom_run(name):
    """BilinearNet with BloomEmbedding user/item layers (layers.py:74-244) fixtures."""
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    errs, frac = replay_bloom_with_oracle(case, rec)
    step = max(v for k, v in errs.items() if k.startswith('grad0') or k == 'loss0')
    assert step < 1e-5, errs
    assert errs['loss'] < 1e-3 and errs['predict_all'] < 1e-5 and errs['predict_pairs'] < 1e-5
    assert_drift(frac)


@pytest.mark.parametrize('seed', [0, 42, 2 ** 31 + 5])
def test_rng_stream_matches_numpy(seed):
    rs, r = np.random.RandomState(seed), Rng(seed=seed)
    for num_items in (1, 2, 100, 1682, 10 ** 6, 10 ** 9, 2 ** 31, 2 ** 32):
        assert (rs.randint(0, num_items, 3000, dtype=np.int64) == r.randint(num_items, 3000)).all()
    for n in (1, 2, 17, 5000):
        idx = np.arange(n)
        rs.shuffle(idx)
        assert (idx == r.shuffle_perm(n)).all()
    a, b = rs.get_state(), r.get_state()
    assert (a[1] == b[1]).all() and a[2] == b[2]


def test_first_model_seed_draw():
    # RandomState(42).randint(-10**8, 10**8) == 99900595; the oracle stream
    # reproduces it as low + bounded(2e8 - 1).
    r = Rng(seed=42)
    assert int(r.randint(2 * 10 ** 8, 1)[0]) - 10 ** 8 == 99900595


def test_murmur_and_bloom_indices_match_sklearn():
    from sklearn.utils import murmurhash3_32
    ids = np.arange(0, 20000, dtype=np.int32)
    seeds = [179424941, 179425457, 179425907, 179426369]
    for s in seeds[:2]:
        h = murmurhash3_32(ids, seed=s)
        for i in range(0, 20000, 97):
            assert int(h[i]) == murmur3_32(int(ids[i]), s)
    comp = 4000
    want = np.stack([murmurhash3_32(ids, seed=s).astype(np.int64) % comp for s in seeds], axis=1)
    want[0] = 0  # padding_idx
    got = bloom_indices(ids.astype(np.int64), seeds, comp, padding_idx=0)
    assert (got == want).all()


@pytest.mark.parametrize('name', golden_names('explicit'))
def test_oracle_replays_reference_explicit_run(name):
    """ExplicitFactorizationModel (factorization/explicit.py:173-284, losses.py:169-244) fixtures."""
    rec = np.load(os.path.join(GOLDEN, name + '.npz'))
    case = case_from_rec(rec)
    errs, frac = replay_explicit_with_oracle(case, rec)  # asserts bit-exact shuffles / rng state
    step = max(v for k, v in errs.items() if k.startswith('grad0') or k == 'loss0')
    assert step < 1e-5, errs
    assert errs['loss'] < 1e-3 and errs['predict_all'] < 1e-5 and errs['predict_pairs'] < 1e-5
    assert_drift(frac)
