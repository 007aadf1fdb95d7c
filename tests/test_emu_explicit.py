"""Explicit-feedback route of the BilinearNet kernels (slk_bilinear_train_explicit: regression / poisson /
logistic losses of spotlight/losses.py:169-244) on the fiber emulator against the CPU oracle and the
fixtures recorded from the live reference's ExplicitFactorizationModel."""
import pytest

import engine_checks as ec
from conftest import GOLDEN
from emu_backend import EmuBackend


@pytest.fixture(scope='module')
def be():
    b = EmuBackend()
    yield b
    b.close()


@pytest.mark.parametrize('loss', ec.EXPLICIT_LOSSES)
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_explicit_train_matches_oracle(be, loss, opt):
    ec.check_explicit_train_matches_oracle(be, loss, opt, 8)


@pytest.mark.parametrize('loss', ec.EXPLICIT_LOSSES)
def test_explicit_single_step_loss_and_gradients(be, loss):
    ec.check_explicit_single_step_gradients(be, loss, 16)


def test_explicit_other_layouts(be):
    ec.check_explicit_train_matches_oracle(be, 'regression', 'adagrad', 64, U=23, I=31, N=77, B=70, epochs=1)
    ec.check_explicit_train_matches_oracle(be, 'logistic', 'adagrad', 3, U=5, I=4, N=130, B=64, epochs=1)
    ec.check_explicit_train_matches_oracle(be, 'poisson', 'sparse_adam', 128, U=9, I=7, N=40, B=16, epochs=1)


@pytest.mark.parametrize('name', ec.EXPLICIT_FIXTURES)
def test_explicit_replays_reference_fixture(be, name):
    ec.check_explicit_replays_reference_fixture(be, GOLDEN, name)


@pytest.mark.parametrize('loss', ec.EXPLICIT_LOSSES)
def test_explicit_fused_and_staged_routes_agree(be, loss):
    ec.check_explicit_routes_agree(be, loss, 'adagrad', 8)
    ec.check_explicit_routes_agree(be, loss, 'adam_dense', 16, U=11, I=60, N=200, B=48)
