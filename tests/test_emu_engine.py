"""Engine sources (spotlight_amd/csrc/*.hip, unmodified) executed by the fiber emulator and
compared with the CPU oracle / numpy / golden vectors.  Covers what a GPU-less box can
cover: indices, keys, segments, loss/optimizer formulas and the C-ABI control flow.  The
same checks run on the real gfx950 library in tests/test_gpu_engine.py."""
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


@pytest.mark.parametrize('num_items', [1, 2, 100, 1682, 4096, 10 ** 6, 10 ** 9, 2 ** 32])
def test_sampler_bit_exact(be, num_items):
    ec.check_sampler_bit_exact(be, num_items)


def test_sampler_fresh_seed_and_block_boundaries(be):
    ec.check_sampler_block_boundaries(be)


def test_sampler_parallel_jump_ahead(be):
    # > 128 state blocks: several workgroups, each jumping ahead with its GF(2) polynomial
    ec.check_sampler_bit_exact(be, 10 ** 6, counts=(300000, 7, 90000))
    ec.check_sampler_bit_exact(be, 2 ** 32, counts=(624 * 128 * 2 + 5,))


@pytest.mark.parametrize('loss,nn', [('bpr', 1), ('hinge', 1), ('pointwise', 1), ('adaptive_hinge', 4), ('regression', 1), ('logistic', 1)])
def test_bias_shadow_is_bit_neutral(be, loss, nn):
    # plain passes; items hot enough for the long-run form + stitch; the latency-bound user pass and the every-head-early item pass
    ec.check_bias_shadow_is_bit_neutral(be, loss, 16, U=400, I=300, N=3000, B=512, nn=nn)
    ec.check_bias_shadow_is_bit_neutral(be, loss, 8, U=50, I=6, N=4000, B=2048, nn=nn, seed=47)
    ec.check_bias_shadow_is_bit_neutral(be, loss, 16, U=400, I=300, N=3000, B=512, nn=nn, seed=48,
                                        options={'user_lat_max_batch': 0, 'item_lat_max_tiles': 0, 'item_long_gate': 0})


@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('hinge', 'sparse_adam'), ('pointwise', 'sgd'), ('bpr', 'sparse_adam')])
def test_user_pingpong_is_bit_neutral(be, loss, opt):
    # plain passes (users recur: a row's current copy alternates); hot users AND hot items (long runs + both stitch kernels); the
    # bandwidth-bound forms forced; together with the item-bias shadow
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 16, U=400, I=300, N=3000, B=512)
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 8, U=7, I=6, N=4000, B=2048, seed=57)
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 16, U=400, I=300, N=3000, B=512, seed=58,
                                          options={'user_lat_max_batch': 0, 'item_lat_max_tiles': 0, 'item_long_gate': 0})
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 5, U=90, I=70, N=2000, B=700, seed=59, options={'chunk_interactions': 1400})
    if opt == 'adagrad':
        ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 16, U=400, I=300, N=3000, B=512, seed=60, with_bias_shadow=True)


@pytest.mark.parametrize('loss', ['bpr', 'hinge', 'pointwise'])
def test_single_occurrence_fast_path_is_bit_neutral(be, loss):
    # inside a ping-pong scope, Adagrad, option item_single_min_items forced to 1: items that occur once in a minibatch are updated by
    # the user pass (k_user_pass<..., SGL>), the item pass skips them (SLK_ITEM_SNAPPPS).  Against plain training, bit for bit:
    # a catalogue far larger than the minibatch (nearly every occurrence single), one in which singles and runs mix, hot items
    # (long runs + stitch beside the skipped singles), several chunks, the bias shadow, odd dims; B > user_lat_max_batch so that the
    # bandwidth-bound form (the one that carries the path) is launched
    opts = {'item_single_min_items': 1, 'user_lat_max_batch': 0}
    n0 = be.engine.get_stat('single_minibatches')
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 16, U=400, I=40000, N=3000, B=512, seed=61, options=opts)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 16, U=400, I=700, N=3000, B=512, seed=62, options=opts)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 8, U=300, I=6, N=4000, B=2048, seed=63, options=opts)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 5, U=90, I=900, N=2000, B=700, seed=64, options=dict(opts, chunk_interactions=1400))
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 16, U=400, I=5000, N=3000, B=512, seed=65, options=opts, with_bias_shadow=True)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 64, U=50, I=3000, N=1500, B=500, seed=66, options=dict(opts, item_long_gate=0))
    assert be.engine.get_stat('single_minibatches') > n0  # (the path did run)
    n1 = be.engine.get_stat('single_minibatches')
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'sparse_adam', 16, U=400, I=40000, N=3000, B=512, seed=67, options=opts)
    assert be.engine.get_stat('single_minibatches') == n1  # (Adagrad only)


def test_user_pingpong_contract(be):
    ec.check_user_pingpong_contract(be)


def test_bias_shadow_refuses_what_it_does_not_cover(be):
    ec.check_bias_shadow_refusals(be)


def test_bias_shadow_lifetime_contract(be):
    ec.check_bias_shadow_lifetime_contract(be)


def test_sampler_long_streams(be):
    # the second stream length class (256 state blocks per stream: the jump table of stride 256), forced at a small size; a
    # draw of several groups (the group's last block is the next one's key) by lowering the class's switch below one stream
    with be.engine.options(mt_long_min_blocks=300):
        ec.check_sampler_bit_exact(be, 10 ** 6, counts=(624 * 256 * 2 + 11, 3, 200000))
        ec.check_sampler_bit_exact(be, 2 ** 32, counts=(624 * 700,))


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
@pytest.mark.parametrize('D', [8, 6])
def test_train_matches_oracle(be, loss, opt, D):
    ec.check_train_matches_oracle(be, loss, opt, D)


@pytest.mark.parametrize('D,B', [(32, 100), (64, 70), (128, 40), (20, 64), (3, 64), (256, 20)])
def test_train_other_layouts(be, D, B):
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', D, U=23, I=31, N=B + 7, B=B, epochs=1)


def test_train_heavy_duplicates_and_tiny_tables(be):
    # every row is hit many times per minibatch; 1 user / 2 items
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 8, U=1, I=2, N=130, B=64, degenerate=True)
    ec.check_train_matches_oracle(be, 'pointwise', 'sparse_adam', 8, U=3, I=1, N=100, B=100, degenerate=True)
    ec.check_train_matches_oracle(be, 'adaptive_hinge', 'adagrad', 8, U=2, I=3, N=65, B=64, nn=5, degenerate=True)
    ec.check_train_matches_oracle(be, 'hinge', 'adam_dense', 4, U=5, I=4, N=1, B=256)  # one interaction


@pytest.mark.parametrize('opt', ['adagrad', 'adam_dense', 'adagrad_dense'])
@pytest.mark.parametrize('late_min', [0, 1 << 40])
def test_adaptive_hinge_item_side_sorted_per_minibatch_or_per_chunk(be, opt, late_min):
    """adaptive hinge's item side: only the live occurrences re-sorted per minibatch after the selection (the default from
    2^18 interactions per minibatch), or all 1+n occurrences sorted once per chunk (the default below): both against the oracle"""
    be.engine.set_option('adaptive_late_min_batch', late_min)
    try:
        ec.check_train_matches_oracle(be, 'adaptive_hinge', opt, 8, nn=4)
        ec.check_train_matches_oracle(be, 'adaptive_hinge', opt, 32, U=23, I=31, N=207, B=100, nn=3, epochs=1)
    finally:
        be.engine.set_option('adaptive_late_min_batch', 1 << 18)


@pytest.mark.parametrize('D,I,B,opt', [(64, 3, 4096, 'adagrad'), (64, 2, 3000, 'sparse_adam'), (8, 2, 3000, 'adam_dense'),
                                        (32, 5, 2500, 'adagrad_dense')])
def test_rows_that_collect_thousands_of_occurrences(be, D, I, B, opt):
    """popular items: an item row's occurrences fill dozens of the item pass's tiles; their partial sums are added by
    k_item_stitch (several batches of G tiles for D = 64) -- against the oracle's plain sequential sums"""
    ec.check_train_matches_oracle(be, 'bpr', opt, D, U=600, I=I, N=2 * B + 50, B=B, epochs=1, degenerate=True)
    ec.check_train_matches_oracle(be, 'pointwise', opt, D, U=3, I=I, N=B + 50, B=B, epochs=1, degenerate=True)
    # the summed item gradients of ONE such minibatch against the exact (float64) ones
    ec.check_long_run_gradients_against_exact(be, 'bpr', D, U=600, I=I, B=B)
    ec.check_long_run_gradients_against_exact(be, 'pointwise', D, U=3, I=I, B=B)


@pytest.mark.parametrize('D,U,I,N,B,opt', [(64, 300, 170, 2500, 512, 'adagrad'), (64, 300, 3, 9000, 4096, 'adagrad'),
                                            (8, 50, 2, 7000, 3000, 'sparse_adam'), (32, 40, 30, 1000, 300, 'adam_dense')])
def test_item_long_gate_is_bit_neutral(be, D, U, I, N, B, opt):
    """short runs only, and tables of 2-3 items whose runs fill dozens of tiles: gated or not, the same bits"""
    ec.check_item_long_gate_is_bit_neutral(be, 'bpr', opt, D, U, I, N, B)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
def test_single_step_loss_and_gradients(be, loss):
    ec.check_single_step_gradients(be, loss, 16)


@pytest.mark.parametrize('name', ec.FIXTURES)
def test_train_replays_reference_fixture(be, name):
    ec.check_replays_reference_fixture(be, GOLDEN, name)


def test_argument_errors(be):
    eng = be.engine
    dev = be.model([np.zeros((4, 8)), np.zeros((5, 8)), np.zeros(4), np.zeros(5)])
    ids = np.zeros(4, dtype=np.int64)
    loss = np.zeros(1, dtype=np.float32)
    with pytest.raises(_native.SlkError) as e:
        eng.bilinear_train(dev.tables, dev.optim, be.ptr(ids), be.ptr(ids), 4, 0, 'bpr', 1, be.ptr(loss))
    assert e.value.code == _native.SLK_EINVAL
    bad = _native.make_tables([be.ptr(x) for x in dev.p], 4, 5, 67)  # dim 67: not %4, > 64
    with pytest.raises(_native.SlkError):
        eng.bilinear_train(bad, dev.optim, be.ptr(ids), be.ptr(ids), 4, 2, 'bpr', 1, be.ptr(loss))
    with pytest.raises(_native.SlkError):
        eng.sample_items(0, 4, be.ptr(ids))
    with pytest.raises(_native.SlkError):
        _native.Engine(7, lib=eng._lib)  # no such device


@pytest.mark.parametrize('loss,opt,nn', [('bpr', 'adagrad', 1), ('hinge', 'sparse_adam', 1), ('pointwise', 'adam_dense', 1),
                                         ('adaptive_hinge', 'adagrad', 4), ('bpr', 'sgd', 1)])
def test_item_pass_with_every_head_early_is_bit_neutral(be, loss, opt, nn):
    """launches of few tiles take k_item_pass<..., NPRE 4> (every head's row + state loaded with the record gather, option
    item_lat_max_tiles); the default form (one head early) must give the same bits"""
    ec.check_item_long_gate_is_bit_neutral(be, loss, opt, 8, U=60, I=45, N=900, B=128, option='item_lat_max_tiles', values=(2048, 0),
                                           default=2048, nn=nn)
    ec.check_item_long_gate_is_bit_neutral(be, loss, opt, 32, U=5, I=3, N=700, B=256, option='item_lat_max_tiles', values=(2048, 0),
                                           default=2048, nn=nn)  # three items: spilled and long runs


def test_pipelined_chunks_match_single_chunk(be):
    """chunk_interactions small => several prep chunks per call: the double-buffered prep pipeline
    (negatives + sorts of chunk c+1 on the prep stream while chunk c trains) must give the same
    results, negatives and RNG state as one big chunk."""
    eng = be.engine
    try:
        eng.set_option('overlap_prep', 1)
        eng.set_option('overlap_min_batch', 0)  # (by default only minibatches >= 2^16 overlap their prep)
        eng.set_option('chunk_interactions', 100)
        ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 8, N=450, B=32, epochs=2)
        ec.check_train_matches_oracle(be, 'adaptive_hinge', 'sparse_adam', 8, N=450, B=32, nn=4, epochs=1)
        ec.check_bloom_train_matches_oracle(be, 'bpr', 'adagrad', 8, user_bloom=2, item_bloom=3, N=450, B=32)
        eng.set_option('overlap_prep', 0)  # several chunks, in order on one stream
        ec.check_train_matches_oracle(be, 'pointwise', 'adagrad', 8, N=450, B=32, epochs=1)
        eng.set_option('item_grid_mult', 1)
        eng.set_option('user_grid_mult', 1)
        ec.check_train_matches_oracle(be, 'hinge', 'adam_dense', 8, N=450, B=32, epochs=1)
    finally:
        eng.set_option('chunk_interactions', 1 << 23)
        eng.set_option('overlap_prep', 0)
        eng.set_option('overlap_min_batch', 1 << 16)
        eng.set_option('item_grid_mult', 128)
        eng.set_option('user_grid_mult', 8)
    ec.check_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 8, U=40, I=30, N=500, B=32, chunk=100)
    ec.check_chunking_is_bit_neutral(be, 'hinge', 'adagrad', 8, U=40, I=30, N=500, B=32, chunk=100, overlap=2)
    ec.check_chunking_is_bit_neutral(be, 'adaptive_hinge', 'sparse_adam', 8, U=40, I=30, N=300, B=32, nn=3, chunk=64,
                                     overlap=0)
    ec.check_chunking_is_bit_neutral(be, 'pointwise', 'adam_dense', 8, U=40, I=50, N=300, B=32, chunk=64,
                                     user_bloom=2, item_bloom=3)
    # the cache-policy option only changes load/store hints
    ec.check_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 8, U=40, I=30, N=500, B=32, chunk=100, overlap=0, nt=0)
    with pytest.raises(_native.SlkError):
        eng.set_option('no_such_option', 1)


@pytest.mark.parametrize('n', [0, 1, 2, 3, 10, 623, 4095, 4096, 4097, 8192, 12345, 65535, 65536, 65537, 65538, 70001, 80000,
                               131073])
def test_device_shuffle_is_numpy_exact(be, n):
    """slk_shuffle_perm: sizes around the in-order tail (4096), the power-of-two range edges and the
    MT19937 block size; RandomState continuity checked through the next randint."""
    ec.check_shuffle_matches_numpy(be, n, seed=n + 1, burn=n % 5, rows=3 if n in (10, 4097) else 0)


@pytest.mark.parametrize('n,band', [(4097, 0), (70001, 0), (131073, 0), (5000, 64), (70001, 8), (131073, 32), (65537, 1024)])
def test_device_shuffle_full_sweeps_and_band_fallback(be, n, band):
    """band 0: the full fixpoint sweeps alone; band > 1: a band that many times too narrow -- ranges leave it, the emit pass
    notices, and the full sweeps redo them: same permutation either way."""
    ec.check_shuffle_matches_numpy(be, n, seed=n + 11, burn=n % 3, band=band)


def test_device_shuffle_random_sizes_and_states(be):
    """sizes and RandomState positions drawn at random (every power-of-two range edge is somebody's neighbour): the banded
    draws must agree with numpy and never need the fall-back"""
    rs = np.random.RandomState(2024)
    for _ in range(24):
        n = int(rs.choice([rs.randint(2, 5000), rs.randint(4000, 9000), rs.randint(9000, 60000)]))
        ec.check_shuffle_matches_numpy(be, n, seed=int(rs.randint(0, 2 ** 31 - 1)), burn=int(rs.randint(0, 1300)))


def test_minibatch_of_one_interaction(be):
    """batch_size 1 (the reference crashes there: squeeze() collapses [1, D]) and a
    last minibatch of one interaction."""
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 16, U=9, I=7, N=5, B=1, epochs=1)
    ec.check_train_matches_oracle(be, 'hinge', 'adam_dense', 8, U=9, I=7, N=65, B=64, epochs=1)


def test_row_ids_near_the_top_of_the_tables(be):
    ec.check_high_row_ids(be, U=(1 << 17) + 3, I=(1 << 16) + 1, N=600, B=256)


TO_SEQUENCE_CASES = [
    # n, users, items, timestamps, max_len, min_len, step
    (1, 5, 9, 'int32', 4, None, None),
    (57, 1, 9, 'int32', 5, None, 1),            # one user: every prefix is a window
    (200, 11, 50, 'int32', 5, None, None),
    (200, 11, 50, 'int32', 5, None, 1),
    (200, 11, 50, 'int32', 5, 3, 2),
    (200, 11, 50, 'int32', 5, 5, 1),            # only full windows
    (200, 11, 50, 'int32', 5, 0, 1),            # sequences[:, -0] is column 0: full windows again
    (200, 11, 50, 'int32', 5, -2, 1),           # negative index: column 2 non-zero
    (300, 40, 50, 'negative', 3, None, 2),
    (300, 40, 50, 'int64_wide', 7, 2, 3),
    (300, 40, 50, 'float', 4, None, 1),
    (300, 40, 50, 'float32', 20, None, 7),      # windows longer than most histories
    (300, 40, 50, 'constant', 6, None, 4),      # all timestamps equal: input order kept
    (5000, 300, 1000, 'int32', 10, None, None),  # several scan tiles
    (5000, 2, 1000, 'int32', 16, 4, 1),          # long histories, many rows per user
    (4500, 3000, 1000, 'int32', 4, 2, 1),        # most users dropped by the minimum length
]


@pytest.mark.parametrize('case', TO_SEQUENCE_CASES, ids=lambda c: '-'.join(str(x) for x in c))
def test_device_to_sequence_matches_host(be, case):
    """slk_to_sequence_plan / _fill == Interactions.to_sequence (interactions.py:170-266), cell for cell."""
    n, users, items, ts_mode, L, min_len, step = case
    ec.check_to_sequence(be, n, users, items, ts_mode, L, min_len, step, seed=n + L)
    # the loop-free restatement the large GPU cases are checked with agrees with the host loop
    inter = ec.to_sequence_case(np.random.RandomState(n + L), n, users, items, ts_mode)
    want = inter.to_sequence(L, min_len, step)
    got, got_users = ec.vectorized_to_sequence(inter, L, min_len, step)
    assert np.array_equal(got, want.sequences) and np.array_equal(got_users, want.user_ids)


def test_device_to_sequence_argument_errors(be):
    eng = be.engine
    d = be.alloc(np.zeros(4, dtype=np.int64))
    for bad in (dict(L=0), dict(step=0), dict(thr=0), dict(thr=5), dict(kind=2)):
        kw = dict(L=4, step=1, thr=1, kind=0)
        kw.update(bad)
        with pytest.raises(_native.SlkError):
            eng.to_sequence_plan(be.ptr(d), be.ptr(d), be.ptr(d), kw['kind'], 4, 3, kw['L'], kw['step'], kw['thr'],
                                 stream=be.stream)
    assert eng.to_sequence_plan(be.ptr(d), be.ptr(d), be.ptr(d), 0, 0, 3, 4, 1, 1, stream=be.stream) == 0  # empty
    eng.to_sequence_fill(be.ptr(d), be.ptr(d), stream=be.stream)
    with pytest.raises(_native.SlkError):  # no plan pending
        eng.to_sequence_fill(be.ptr(d), be.ptr(d), stream=be.stream)



@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('pointwise', 'sparse_adam'), ('adaptive_hinge', 'adagrad'),
                                       ('hinge', 'adam_dense')])
def test_train_closed_loop_several_minibatches(be, loss, opt):
    """the quota-free form of the multi-minibatch comparison (the GPU suite runs it at 50k interactions)"""
    ec.check_train_closed_loop(be, loss, opt, 16, U=300, I=100, N=1300, B=256, nn=3, epochs=2)


@pytest.mark.parametrize('D,U,B', [(64, 3, 4096), (16, 2, 1000), (32, 40, 3000), (8, 1, 700)])
def test_users_that_collect_thousands_of_occurrences(be, D, U, B):
    """hot users: a user's occurrences fill dozens of the user pass's tiles (k_user_pass<ULONG>: one row group per tile-sized
    segment, k_user_stitch adds the partials) -- the summed user gradients against the exact (float64) ones, every loss;
    U = 40: runs of ~75 that cover one or two tiles next to runs that cover none"""
    ec.check_long_user_run_gradients_against_exact(be, 'bpr', D, U=U, I=5000, B=B)
    ec.check_long_user_run_gradients_against_exact(be, 'pointwise', D, U=U, I=300, B=B)


@pytest.mark.parametrize('loss,opt', [('pointwise', 'adagrad'), ('bpr', 'sparse_adam'), ('hinge', 'adam_dense'),
                                       ('adaptive_hinge', 'adagrad')])
def test_hot_users_closed_loop(be, loss, opt):
    """minibatches of 1500 over 25 users (runs of ~60: some cover a tile of the user pass, some do not) against the oracle,
    step by step and element by element"""
    ec.check_train_closed_loop(be, loss, opt, 16, U=25, I=4000, N=3300, B=1500, nn=3, epochs=1, seed=8)


@pytest.mark.parametrize('D,U,I,N,B,opt', [(64, 300, 170, 2500, 512, 'adagrad'), (16, 3, 2000, 5000, 2048, 'adagrad'),
                                            (8, 2, 50, 7000, 3000, 'sparse_adam')])
def test_user_long_gate_is_bit_neutral(be, D, U, I, N, B, opt):
    """the plain user pass for minibatches without a long user run (k_user_long_flags) against the partial-writing form +
    k_user_stitch for every minibatch: same bits (short runs are walked identically; long ones take the long form either way)"""
    ec.check_item_long_gate_is_bit_neutral(be, 'bpr', opt, D, U, I, N, B)


@pytest.mark.parametrize('D', [4, 16, 31, 64, 100, 128, 256])
def test_scores_are_the_fma_chain(be, D):
    ec.check_scores_are_the_fma_chain(be, D)


def test_fused_ranks(be):
    ec.check_fused_ranks(be)
    ec.check_fused_ranks(be, D=64, U=200, I=1500, n_rows=300, seed=9)


def test_every_option_of_the_library_is_covered_below():
    """"result-neutral" is a property the tests must keep proving for every option -- a new row in the
    library's table without a row here fails."""
    names = ec.option_names_of_the_library()
    assert len(names) >= 26 and len(set(names)) == len(names)
    assert set(names) == set(ec.OPTION_VALUES) | set(ec.OPTIONS_NOT_RESULT_NEUTRAL) | set(ec.OPTIONS_NEUTRAL_TO_SUMMATION_ORDER)


def test_every_option_is_documented_in_the_header():
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'spotlight_hip.h')).read()
    doc = hdr[hdr.index('/* Tuning knobs'):hdr.index('int slk_ctx_set_option')]
    missing = set(ec.option_names_of_the_library()) - set(re.findall(r'"([a-z_]+)"', doc))
    assert not missing, missing


@pytest.mark.parametrize('name', sorted(ec.OPTION_VALUES))
def test_option_is_result_neutral(be, name):
    ec.check_option_is_result_neutral(be, name, ec.OPTION_VALUES[name])


@pytest.mark.parametrize('name', sorted(ec.OPTIONS_NEUTRAL_TO_SUMMATION_ORDER))
def test_option_is_neutral_to_summation_order(be, name):
    ec.check_option_is_neutral_to_summation_order(be, name, ec.OPTIONS_NEUTRAL_TO_SUMMATION_ORDER[name])


def test_prefetch_behind_an_inline_draw(be):
    ec.check_prefetch_behind_an_inline_draw(be, D=8, U=40, I=30, N=500, B=32, chunk=100)


def test_user_bias_zero_hint_is_bit_neutral(be):
    ec.check_user_bias_zero_hint_is_bit_neutral(be, D=8, U=60, I=40, N=900, B=256)
