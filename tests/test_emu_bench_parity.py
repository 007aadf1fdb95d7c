"""The bench-size parity checks (tests/bench_parity.py) at toy sizes through the emulator build: keeps the checker
itself -- compaction, teacher forcing, the conditioned Adagrad bound -- under test on a box without a GPU."""
import pytest
import torch

import bench_parity as bp
from emu_backend import emu_lib
from spotlight_amd import _native


@pytest.fixture(scope='module')
def emu():
    eng = _native.Engine(0, lib=emu_lib())
    yield eng, torch.device('cpu'), 0
    eng.close()


@pytest.mark.parametrize('trained', [False, True])
def test_bilinear_compacted_minibatch(emu, trained):
    eng, dev, stream = emu
    out = bp.bilinear_minibatch_parity(eng, dev, stream, 5000, 900, 16, 700, loss='bpr', trained=trained,
                                       scale=None if not trained else 0.125, seed=int(trained))
    assert out['users_touched'] <= 700 and out['items_touched'] <= 1400


def test_bilinear_bloom_adaptive_minibatch(emu):
    eng, dev, stream = emu
    bp.bilinear_minibatch_parity(eng, dev, stream, 4000, 800, 32, 300, loss='adaptive_hinge', nn=5, bloom_rows=160, n_hash=4,
                                 trained=True, scale=0.1, seed=3)


@pytest.mark.parametrize('pad_frac', [0.0, 0.3])
def test_poolnet_minibatch(emu, pad_frac):
    eng, dev, stream = emu
    bp.poolnet_minibatch_parity(eng, dev, stream, 500, 16, 12, 20, loss='bpr', trained=pad_frac > 0,
                                scale=None if pad_frac == 0 else 0.125, pad_frac=pad_frac, seed=4)


def test_multi_chunk_teacher_forcing(emu):
    eng, dev, stream = emu
    eng.set_option('chunk_interactions', 512)  # 2 minibatches of 256 per prep chunk
    try:
        out = bp.multi_chunk_parity(eng, dev, stream, 3000, 700, 16, 256, n_full=5, tail=77, check_at=(2, 5))
    finally:
        eng.set_option('chunk_interactions', 1 << 23)
    assert out['minibatches'] == 6 and len(out['checked']) == 2


def test_bilinear_saturated_pairs(emu):
    eng, dev, stream = emu
    U, I, D, B = 5000, 900, 16, 700
    tables, state, users, items = bp.saturated_problem(dev, U, I, D, B, seed=1, row_scale=0.7)
    out = bp.bilinear_minibatch_parity(eng, dev, stream, U, I, D, B, loss='bpr', tables=tables, state=state, users=users,
                                       items=items, seed=11)
    assert out['loss'] < 0.4, out


@pytest.mark.parametrize('trained', [False, True])
def test_bilinear_sparse_adam_minibatch(emu, trained):
    eng, dev, stream = emu
    bp.bilinear_minibatch_parity(eng, dev, stream, 5000, 900, 16, 700, loss='bpr', trained=trained,
                                 scale=None if not trained else 0.5, bias_scale=0.5, seed=20 + int(trained), opt='sparse_adam',
                                 step0=100 if trained else 0)


@pytest.mark.parametrize('route', ['epoch', 'launch'])
def test_small_minibatches_route_check(emu, route):
    eng, dev, stream = emu
    eng.set_option('epoch_kernel', 1 if route == 'epoch' else 0)
    try:
        out = bp.multi_chunk_parity(eng, dev, stream, 3000, 700, 16, 64, n_full=6, tail=23, check_at=(3, 6), seed=64,
                                    expect_route=route)
    finally:
        eng.set_option('epoch_kernel', 1)
    assert out['minibatches'] == 7


def test_sharded_world1_vs_fused(emu):
    import os
    import torch.distributed as dist
    eng, dev, stream = emu
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29462')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        out = bp.sharded_world1_vs_fused(eng, dev, stream, 3000, 5000, 16, 512, n_mb=2, seed=5, block_rows=1024)
    finally:
        dist.destroy_process_group()
    assert out['rows_differing']['param1'] < 5000
