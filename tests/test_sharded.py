"""Row-sharded training: world_size-2 (and 3) gloo runs of the real host
code (spotlight_amd/factorization/sharded.py) over the emulator build of the engine,
compared on rank 0 with the CPU oracle and with the single-device engine."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'dist', 'shard_worker.py')
MODEL_WORKER = os.path.join(HERE, 'dist', 'shard_model_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_world(world, args, backend='emu', timeout=600, worker=None, token='SHARD_PARITY_OK'):
    worker = worker or WORKER
    from emu_backend import emu_lib
    if backend == 'emu':
        emu_lib()  # build once, before the ranks race for it
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, worker, backend] + [str(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-4000:])
    assert token in outs[0], outs[0][-2000:]


@pytest.mark.parametrize('loss,opt,D', [('bpr', 'adagrad', 8), ('hinge', 'sparse_adam', 8),
                                        ('pointwise', 'adam_dense', 6), ('bpr', 'adagrad', 64)])
def test_sharded_world2_matches_oracle_and_single_device(loss, opt, D):
    run_world(2, [loss, opt, D])


def test_sharded_popular_items_long_runs(monkeypatch):
    """4 items in all: every owner's item pass sees runs of ~800 gradient slots (a dozen tiles at dim 64), summed through the
    per-tile partials + k_item_stitch in the exchange-slot (BLK) mode"""
    monkeypatch.setenv('SHARD_TEST_SHAPE', '300,4,1600,1')  # one minibatch: no trajectory to go chaotic
    run_world(2, ['bpr', 'adagrad', 64, 'chunk', 2])


@pytest.mark.parametrize('world,slices,loss,opt', [(2, 1, 'bpr', 'adagrad'), (2, 3, 'pointwise', 'sparse_adam'),
                                                   (3, 2, 'hinge', 'adam_dense'), (1, 2, 'bpr', 'adagrad')])
def test_sharded_whole_chunk_with_slices(world, slices, loss, opt):
    """Every minibatch of the run in ONE chunk (one count exchange, no per-minibatch host sync), each
    minibatch cut into `slices` user-slices whose exchanges are issued asynchronously."""
    run_world(world, [loss, opt, 8, 'chunk', slices])


@pytest.mark.parametrize('world,mode,slices,loss', [(2, 'chunk', 3, 'bpr'), (3, 'train', 2, 'pointwise'), (2, None, None, 'adaptive_hinge')])
def test_sharded_bias_shadow_is_bit_neutral(world, mode, slices, loss, monkeypatch, tmp_path):
    """The row-sharded path inside a bias-shadow scope (slk_bias_shadow_begin: every rank's item biases + Adagrad accumulator
    interleaved; the owner-side gather and the item pass index the copy) against the plain layout: every rank's shards and
    accumulators bit for bit, and each run against the oracle and the one-GPU engine as always."""
    import numpy as np
    args = [loss, 'adagrad', 8] + ([mode, slices] if mode else [])
    if mode == 'chunk':  # 4 items in all: long runs, summed through the per-tile partials + k_item_stitch of the BLK mode
        monkeypatch.setenv('SHARD_TEST_SHAPE', '300,4,1600,1')
        args[2] = 64
    for tag in ('plain', 'shadow'):
        monkeypatch.setenv('SHARD_TEST_SHADOW', '1' if tag == 'shadow' else '0')
        monkeypatch.setenv('SHARD_TEST_DUMP', str(tmp_path / tag))
        run_world(world, args)
    for r in range(world):
        a, b = (np.load(str(tmp_path / tag) + '.rank%d.npz' % r) for tag in ('plain', 'shadow'))
        assert len(a.files) == len(b.files) == 8
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (r, k)
        assert np.abs(a['arr_3']).max() > 0  # (the item biases did train)


@pytest.mark.parametrize('world,slices', [(2, 4), (3, 2), (4, 4), (8, 4), (8, 1)])
def test_sharded_bench_loop_equal_shares(world, slices):
    """The loop bench.py runs at N > 1: ShardedBilinearTrainer.train over equal per-rank shares of every
    global minibatch, negatives drawn on each rank's device, several minibatches per chunk; up to the 8 ranks x 4
    slices of `bench.py --gpus 8`."""
    run_world(world, ['bpr', 'adagrad', 16, 'train', slices])


@pytest.mark.parametrize('world,mode,slices,opt', [(2, None, None, 'adagrad'), (3, 'chunk', 2, 'adagrad'), (2, 'chunk', 3, 'sparse_adam'),
                                                   (1, 'chunk', 2, 'adagrad'), (2, 'train', 2, 'adagrad'), (3, None, None, 'adam_dense'),
                                                   (1, 'sample', None, 'adagrad')])
def test_sharded_adaptive_hinge(world, mode, slices, opt):
    """loss='adaptive_hinge' on the row-sharded path (VERDICT r04 missing 1; spotlight/factorization/implicit.py:266-275,
    spotlight/losses.py:127-166): the view(n, B) quirk makes a column's candidates the draws of OTHER interactions -- other
    ranks' -- so the step scores all 1 + n pairs first, sums the score matrix over the ranks and selects on every rank.
    Against the oracle and the one-GPU engine on the same minibatches and draws."""
    args = ['adaptive_hinge', opt, 8]
    if mode:
        args += [mode] + ([slices] if slices else [])
    run_world(world, args)


def test_sharded_adaptive_hinge_many_draws_dim64(monkeypatch):
    monkeypatch.setenv('SHARD_TEST_NNEG', '7')
    run_world(2, ['adaptive_hinge', 'adagrad', 64, 'chunk', 2])


def test_sharded_world3_device_sampled_negatives():
    run_world(3, ['bpr', 'adagrad', 16, 'sample'])


def test_sharded_world1_degenerates_to_local_exchange():
    run_world(1, ['pointwise', 'adagrad_dense', 8])


def test_sharded_train_loop_chunked_sampling():
    # ShardedBilinearTrainer.train: negatives drawn several minibatches at a time == numpy's stream
    run_world(1, ['bpr', 'adagrad', 16, 'sample'])


@pytest.mark.parametrize('world,loss,opt', [(2, 'bpr', 'adagrad'), (3, 'pointwise', 'adam'), (2, 'hinge', 'sparse_adam'),
                                            (2, 'adaptive_hinge', 'adagrad'), (3, 'adaptive_hinge', 'adagrad')])
def test_sharded_model_fit_predict_match_single_device_model(world, loss, opt):
    """The drop-in ShardedImplicitFactorizationModel: same seed => same RandomState consumption
    (shuffles, negatives), same tables and predictions as ImplicitFactorizationModel."""
    run_world(world, [loss, opt], worker=MODEL_WORKER, token='SHARD_MODEL_OK')


@pytest.mark.parametrize('world,loss,opt', [(2, 'bpr', 'adagrad'), (3, 'adaptive_hinge', 'adagrad'), (2, 'hinge', 'sparse_adam')])
def test_sharded_model_fit_with_shadowed_item_biases(world, loss, opt, monkeypatch):
    """The same comparison with every rank's item biases shadowed for the duration of fit() (what the model does for local
    shards of >= 2^24 item rows; forced here): Adagrad trains on the interleaved copy, SparseAdam is left alone."""
    monkeypatch.setenv('SHARD_TEST_SHADOW_FLOOR', '1')
    run_world(world, [loss, opt], worker=MODEL_WORKER, token='SHARD_MODEL_OK')


@pytest.mark.gpu
def test_gpu_sharded_model_world1_nccl(monkeypatch):
    run_world(1, ['bpr', 'adagrad'], backend='hip', worker=MODEL_WORKER, token='SHARD_MODEL_OK')
    monkeypatch.setenv('SHARD_TEST_SHADOW_FLOOR', '1')  # ... and with the item biases shadowed for the duration of fit()
    run_world(1, ['bpr', 'adagrad'], backend='hip', worker=MODEL_WORKER, token='SHARD_MODEL_OK')


@pytest.mark.gpu
@pytest.mark.parametrize('loss,opt,D', [('bpr', 'adagrad', 64), ('hinge', 'sparse_adam', 32),
                                        ('pointwise', 'adam_dense', 8)])
def test_gpu_sharded_phases_world1_nccl(loss, opt, D):
    """The real gfx950 kernels of the four shard phases + RCCL all_to_all_single (a single rank:
    the exchange degenerates to a device copy), against the oracle and the fused one-GPU path."""
    run_world(1, [loss, opt, D], backend='hip')


@pytest.mark.gpu
@pytest.mark.parametrize('opt,D', [('adagrad', 64), ('sparse_adam', 32)])
def test_gpu_sharded_adaptive_hinge_world1_nccl(opt, D):
    run_world(1, ['adaptive_hinge', opt, D, 'chunk', 2], backend='hip')


@pytest.mark.gpu
def test_gpu_sharded_adaptive_hinge_model_world1_nccl():
    run_world(1, ['adaptive_hinge', 'adagrad'], backend='hip', worker=MODEL_WORKER, token='SHARD_MODEL_OK')


@pytest.mark.gpu
def test_gpu_sharded_bias_shadow_world1_nccl(monkeypatch, tmp_path):
    import numpy as np
    for tag in ('plain', 'shadow'):
        monkeypatch.setenv('SHARD_TEST_SHADOW', '1' if tag == 'shadow' else '0')
        monkeypatch.setenv('SHARD_TEST_DUMP', str(tmp_path / tag))
        run_world(1, ['bpr', 'adagrad', 64, 'chunk', 2], backend='hip')
    a, b = (np.load(str(tmp_path / tag) + '.rank0.npz') for tag in ('plain', 'shadow'))
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_gpu_sharded_whole_chunk_with_slices_world1_nccl():
    run_world(1, ['bpr', 'adagrad', 64, 'chunk', 4], backend='hip')


@pytest.mark.gpu
def test_gpu_sharded_world1_device_sampled_negatives():
    run_world(1, ['bpr', 'adagrad', 16, 'sample'], backend='hip')


@pytest.mark.gpu
@pytest.mark.parametrize('world,args', [(2, ['bpr', 'adagrad', 64]), (2, ['bpr', 'adagrad', 64, 'chunk', 3]),
                                        (3, ['hinge', 'sparse_adam', 32, 'chunk', 2]), (2, ['bpr', 'adagrad', 16, 'train', 2]),
                                        (2, ['adaptive_hinge', 'adagrad', 64, 'chunk', 2]), (3, ['bpr', 'adagrad', 16, 'sample'])])
def test_gpu_sharded_remote_peers_on_one_gpu_over_gloo(world, args):
    """the N > 1 path had never run a HIP kernel against a REMOTE peer (gloo + emulator, or RCCL world 1).
    Here `world` processes share the box's one GPU and exchange over gloo (device tensors; RCCL refuses two ranks per device):
    the gfx950 kernels of slk_shard.hip on real remote ids, rows and gradient slots, against the oracle and the one-GPU engine."""
    run_world(world, args, backend='hipgloo')


@pytest.mark.gpu
def test_gpu_sharded_remote_peers_long_runs_and_bias_shadow_over_gloo(monkeypatch, tmp_path):
    """... with 4 items in all (runs of ~800 gradient slots per owner: partials + stitch in the exchange-slot mode), plain and on
    the item-bias shadow: every rank's shards bit for bit."""
    import numpy as np
    monkeypatch.setenv('SHARD_TEST_SHAPE', '300,4,1600,1')
    for tag in ('plain', 'shadow'):
        monkeypatch.setenv('SHARD_TEST_SHADOW', '1' if tag == 'shadow' else '0')
        monkeypatch.setenv('SHARD_TEST_DUMP', str(tmp_path / tag))
        run_world(2, ['bpr', 'adagrad', 64, 'chunk', 2], backend='hipgloo')
    for r in range(2):
        a, b = (np.load(str(tmp_path / tag) + '.rank%d.npz' % r) for tag in ('plain', 'shadow'))
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (r, k)
