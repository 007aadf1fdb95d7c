"""BloomEmbedding user/item layers in the fused BilinearNet path (in-kernel murmur3 hashing,
hashed-row owner passes), run by the fiber emulator against the CPU oracle and the fixtures
recorded from the live reference.  Same checks on the real library: tests/test_gpu_engine.py."""
import pytest

import engine_checks as ec
from conftest import GOLDEN
from emu_backend import EmuBackend


@pytest.fixture(scope='module')
def be():
    b = EmuBackend()
    yield b
    b.close()


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ['adagrad', 'adam_dense', 'adagrad_dense', 'sparse_adam'])
def test_item_bloom_train_matches_oracle(be, loss, opt):
    ec.check_bloom_train_matches_oracle(be, loss, opt, 8, user_bloom=0, item_bloom=4)


@pytest.mark.parametrize('ub,ib', [(4, 0), (4, 4), (2, 3), (1, 8)])
@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('adaptive_hinge', 'adam_dense')])
def test_user_and_both_bloom_train_matches_oracle(be, ub, ib, loss, opt):
    ec.check_bloom_train_matches_oracle(be, loss, opt, 8, user_bloom=ub, item_bloom=ib)


@pytest.mark.parametrize('D', [32, 128, 6])
def test_bloom_other_layouts(be, D):
    ec.check_bloom_train_matches_oracle(be, 'bpr', 'adagrad', D, user_bloom=2, item_bloom=4, N=80, B=50, epochs=1)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('ub,ib', [(0, 4), (4, 4)])
def test_bloom_single_step_loss_and_gradients(be, loss, ub, ib):
    ec.check_bloom_single_step_gradients(be, loss, 16, user_bloom=ub, item_bloom=ib)


@pytest.mark.parametrize('name', ec.BLOOM_FIXTURES)
def test_bloom_replays_reference_fixture(be, name):
    ec.check_bloom_replays_reference_fixture(be, GOLDEN, name)
