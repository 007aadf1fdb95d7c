"""PoolNet / ImplicitSequenceModel kernels (spotlight_amd/csrc/slk_seq.hip, unmodified) run by
the fiber emulator against the CPU oracle and the sequence fixtures recorded from the live
reference.  The same checks run on the real gfx950 library in tests/test_gpu_engine.py."""
import numpy as np
import pytest

import engine_checks as ec
from conftest import GOLDEN
from emu_backend import EmuBackend
from spotlight_amd import _native


@pytest.fixture(scope='module')
def be():
    b = EmuBackend()
    yield b
    b.close()


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_seq_train_matches_oracle(be, loss, opt):
    ec.check_seq_train_matches_oracle(be, loss, opt, 8)


@pytest.mark.parametrize('loss', ['bpr', 'pointwise'])
def test_seq_popular_items_long_runs(be, loss):
    """3 items besides the padding row: every item row collects ~1600 occurrences of the minibatch (25 tiles of the item pass
    at dim 64), summed through per-tile partials + k_item_stitch in the PoolNet (SEQ) mode; the oracle sums sequentially, so
    the comparison allows what thousands of cancelling fp32 terms allow"""
    ec.check_seq_single_step_gradients(be, loss, 64, I=4, B=60, L=40, tol=1e-4)
    ec.check_seq_single_step_gradients(be, loss, 8, I=3, B=80, L=30, seed=5, tol=1e-4)


@pytest.mark.parametrize('D,L,B', [(32, 20, 10), (64, 33, 6), (128, 5, 8), (6, 300, 3), (3, 1, 16), (16, 130, 4)])
def test_seq_other_layouts_and_lengths(be, D, L, B):
    # L > number of row groups (several timesteps per chunk), L == 1, odd dims
    ec.check_seq_train_matches_oracle(be, 'bpr', 'adagrad', D, I=47, N=B + 3, L=L, B=B, epochs=1)


def test_seq_no_padding_and_heavy_duplicates(be):
    ec.check_seq_train_matches_oracle(be, 'pointwise', 'adagrad', 8, I=5, N=20, L=12, B=8, pad_frac=0.0, tol=1e-4)
    ec.check_seq_train_matches_oracle(be, 'adaptive_hinge', 'sparse_adam', 8, I=60, N=12, L=6, B=12, nn=5, epochs=1)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
def test_seq_single_step_loss_and_gradients(be, loss):
    ec.check_seq_single_step_gradients(be, loss, 16)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ['adagrad', 'sparse_adam', 'adam_dense'])
def test_seq_bloom_item_layer_matches_oracle(be, loss, opt):
    """PoolNet over a BloomEmbedding item layer (sequence/representations.py:62-68 + layers.py:74-244):
    in-kernel hashing, sums of hashed rows, hashed-row owner pass + plain bias pass."""
    ec.check_seq_train_matches_oracle(be, loss, opt, 8, I=60, bloom=2)
    ec.check_seq_single_step_gradients(be, loss, 16, I=80, bloom=4, ratio=0.25)


def test_seq_bloom_other_layouts(be):
    ec.check_seq_train_matches_oracle(be, 'bpr', 'adagrad', 64, I=300, N=20, L=33, B=8, epochs=1, bloom=4, ratio=0.2)
    ec.check_seq_train_matches_oracle(be, 'pointwise', 'adagrad', 128, I=90, N=11, L=5, B=8, epochs=1, bloom=3)
    ec.check_seq_train_matches_oracle(be, 'hinge', 'adagrad', 6, I=50, N=6, L=300, B=3, epochs=1, bloom=2)


@pytest.mark.parametrize('name', ec.SEQ_FIXTURES)
def test_seq_replays_reference_fixture(be, name):
    ec.check_seq_replays_reference_fixture(be, GOLDEN, name)


def test_seq_argument_errors(be):
    eng = be.engine
    dev = be.seq_model([np.zeros((5, 8)), np.zeros(5)])
    seqs = np.zeros((2, 4), dtype=np.int64)
    loss = np.zeros(1, dtype=np.float32)
    with pytest.raises(_native.SlkError) as e:
        eng.poolnet_train(dev.tables, dev.optim, 0, be.ptr(seqs), 2, 0, 2, 'bpr', 1, be.ptr(loss))
    assert e.value.code == _native.SLK_EINVAL
    with pytest.raises(_native.SlkError):
        eng.poolnet_train(dev.tables, dev.optim, 7, be.ptr(seqs), 2, 4, 2, 'bpr', 1, be.ptr(loss))  # padding_idx
    with pytest.raises(_native.SlkError):
        eng.poolnet_train(dev.tables, dev.optim, 0, be.ptr(seqs), 2, 100000, 2, 'bpr', 1, be.ptr(loss))  # LDS



def test_seq_item_pass_with_every_head_early_is_bit_neutral(be):
    for loss, opt in (('bpr', 'adagrad'), ('adaptive_hinge', 'sparse_adam'), ('pointwise', 'adam_dense')):
        ec.check_seq_chunking_is_bit_neutral(be, loss, opt, 16, chunk=1 << 23, overlap=0, option=('item_lat_max_tiles', 2048, 0, 2048))
    ec.check_seq_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 16, bloom=3, chunk=1 << 23, overlap=0, option=('item_lat_max_tiles', 2048, 0, 2048))


@pytest.mark.parametrize('loss,opt,bloom', [('bpr', 'adagrad', 0), ('adaptive_hinge', 'sparse_adam', 0), ('hinge', 'adam_dense', 3)])
def test_seq_pass_forms_are_bit_identical(be, loss, opt, bloom):
    """Both forms of the sequence pass leave the same record (representation | own item's contribution): register-resident (1,
    the default) and LDS-staged (0)"""
    ec.check_seq_chunking_is_bit_neutral(be, loss, opt, 16, bloom=bloom, chunk=1 << 23, overlap=0, option=('seq_variant', 1, 0, 1))


@pytest.mark.parametrize('loss,opt,bloom', [('bpr', 'adagrad', 0), ('adaptive_hinge', 'sparse_adam', 0), ('pointwise', 'adam_dense', 0),
                                            ('bpr', 'adagrad', 4)])
@pytest.mark.parametrize('overlap', [0, 1])
def test_seq_chunked_and_pipelined_prep_is_bit_neutral(be, loss, opt, bloom, overlap):
    """PoolNet: several prep chunks per call, in line and with the next chunk's prep on the second stream"""
    ec.check_seq_chunking_is_bit_neutral(be, loss, opt, 16, bloom=bloom, overlap=overlap)
