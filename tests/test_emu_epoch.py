"""The persistent epoch kernel (csrc/slk_epoch.hip) through the emulator build: the cooperative grid runs as
concurrently scheduled fiber blocks (tests/emu), so the grid barrier, the phase structure, the gap sweeps of the dense
optimizers and the bit-identity with the per-minibatch launches are exercised without a GPU.  (What the emulator
cannot show -- cross-XCD visibility of the sc1 accesses -- is what the same checks assert under `-m gpu`.)"""
import pytest

import engine_checks as ec
from emu_backend import EmuBackend


@pytest.fixture(scope='module')
def be():
    b = EmuBackend()
    yield b
    b.close()


@pytest.mark.parametrize('loss', ['pointwise', 'bpr', 'hinge'])
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_bit_identical_to_launch_path(be, loss, opt):
    ec.check_epoch_kernel_is_bit_identical(be, loss, opt, 8)


@pytest.mark.parametrize('loss', ['regression', 'poisson', 'logistic'])
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_explicit_feedback_bit_identical_to_launch_path(be, loss, opt):
    """ExplicitFactorizationModel's minibatch loop (factorization/explicit.py:213-236) inside the persistent launch"""
    ec.check_epoch_kernel_is_bit_identical(be, loss, opt, 8)


@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_adaptive_hinge_bit_identical_to_launch_path(be, opt):
    """adaptive hinge (implicit.py:266-275; 5 draws per interaction, the reference's default) inside the persistent launch:
    score phase, the view(n, B) selection inside the user phase, the item phase over all 1 + n occurrences"""
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', opt, 8)


def test_epoch_kernel_adaptive_hinge_layouts_duplicates_chunks(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 32, U=23, I=31, N=307, B=100, nn=3)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adam_dense', 3, U=23, I=31, N=199, B=64, epochs=1, nn=1)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'sparse_adam', 8, U=1, I=2, N=130, B=64, nn=7)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 64, U=40, I=30, N=300, B=70, nn=20)  # more pairs than lanes
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'sgd', 16, N=1500, B=128, chunk=256, nn=5)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 8, U=40, I=30, N=300, B=64, max_grid=1, nn=4)


def test_epoch_kernel_explicit_feedback_layouts_duplicates_chunks(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'regression', 'adagrad', 32, U=23, I=31, N=307, B=100)
    ec.check_epoch_kernel_is_bit_identical(be, 'logistic', 'adam_dense', 3, U=23, I=31, N=199, B=64, epochs=1)
    ec.check_epoch_kernel_is_bit_identical(be, 'poisson', 'sparse_adam', 8, U=1, I=2, N=130, B=64)
    ec.check_epoch_kernel_is_bit_identical(be, 'regression', 'adagrad_dense', 16, N=1500, B=128, chunk=256)


@pytest.mark.parametrize('D,B', [(32, 100), (64, 70), (20, 64), (3, 64)])
def test_epoch_kernel_other_layouts(be, D, B):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', D, U=23, I=31, N=3 * B + 7, B=B)
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adam_dense', D, U=23, I=31, N=3 * B + 7, B=B, epochs=1)


def test_epoch_kernel_heavy_duplicates_tiny_tables_and_one_workgroup(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', 8, U=1, I=2, N=130, B=64)
    ec.check_epoch_kernel_is_bit_identical(be, 'hinge', 'adam_dense', 8, U=3, I=1, N=100, B=100)
    ec.check_epoch_kernel_is_bit_identical(be, 'pointwise', 'sparse_adam', 8, U=40, I=30, N=300, B=64, max_grid=1)


def test_epoch_kernel_cooperative_launch_form(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', 16, cooperative=1, epochs=1)


def test_epoch_kernel_several_chunks(be):
    """chunks of 2 minibatches: the launch is repeated per chunk, the step count and RNG stream carry over"""
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'sparse_adam', 16, N=1500, B=128, chunk=256)
    ec.check_epoch_kernel_is_bit_identical(be, 'pointwise', 'adagrad_dense', 16, N=1500, B=128, chunk=256)


def test_epoch_kernel_wider_cooperative_grid(monkeypatch):
    """A context that believes in 12 CUs: grids of up to 12 concurrently resident workgroups and strided positions."""
    monkeypatch.setenv('SLK_EMU_CUS', '12')
    b = EmuBackend()
    try:
        ec.check_epoch_kernel_is_bit_identical(b, 'bpr', 'adagrad', 32, U=500, I=300, N=1300, B=512, epochs=1)
        ec.check_epoch_kernel_is_bit_identical(b, 'pointwise', 'adam_dense', 16, U=90, I=70, N=700, B=200, epochs=1, max_grid=5)
        # two-level barrier: 12 workgroups in 8 groups of 1-2, and 5 workgroups (fewer than groups)
        ec.check_epoch_kernel_is_bit_identical(b, 'bpr', 'sparse_adam', 32, U=500, I=300, N=1300, B=512, epochs=1, barrier=1)
        ec.check_epoch_kernel_is_bit_identical(b, 'hinge', 'adagrad', 16, U=90, I=70, N=700, B=200, epochs=1, max_grid=5, barrier=1)
        ec.check_epoch_kernel_is_bit_identical(b, 'adaptive_hinge', 'adagrad', 32, U=500, I=300, N=1300, B=512, epochs=1)
        ec.check_epoch_kernel_is_bit_identical(b, 'adaptive_hinge', 'adam_dense', 16, U=90, I=70, N=700, B=200, epochs=1, max_grid=5, barrier=1, nn=3)
    finally:
        b.close()
