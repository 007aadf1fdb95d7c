"""Parity tests proper: the real gfx950 library (csrc/libspotlight_hip.so) on cuda:0 through
its C ABI, against the CPU oracle, numpy and the golden vectors recorded from the live
reference.  Run with `python -m pytest tests -m gpu` on an MI355X."""
import pytest

import engine_checks as ec
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from hip_backend import HipBackend
    b = HipBackend()
    yield b
    b.close()


@pytest.mark.parametrize('num_items', [1, 2, 100, 1682, 4096, 10 ** 6, 10 ** 9, 2 ** 32])
def test_sampler_bit_exact(be, num_items):
    ec.check_sampler_bit_exact(be, num_items, counts=(1, 5, 700, 3000, 200000))


def test_sampler_fresh_seed_and_block_boundaries(be):
    ec.check_sampler_block_boundaries(be)


def test_sampler_parallel_jump_ahead(be):
    # many workgroups jumping ahead; 12M draws also crosses the 10.2M-word launch limit
    ec.check_sampler_bit_exact(be, 10 ** 6, counts=(300000, 7, 90000, 12000000))
    ec.check_sampler_bit_exact(be, 2 ** 32, counts=(624 * 128 * 2 + 5, 624 * 128 * 128 + 1))
    ec.check_sampler_bit_exact(be, 1682, counts=(5000000,))


@pytest.mark.parametrize('loss,nn', [('bpr', 1), ('pointwise', 1), ('adaptive_hinge', 5), ('poisson', 1)])
def test_bias_shadow_is_bit_neutral(be, loss, nn):
    ec.check_bias_shadow_is_bit_neutral(be, loss, 64, U=200000, I=100000, N=300000, B=65536, nn=nn)
    ec.check_bias_shadow_is_bit_neutral(be, loss, 32, U=943, I=1682, N=20000, B=4096, nn=nn, seed=47)
    ec.check_bias_shadow_is_bit_neutral(be, loss, 64, U=5000, I=40, N=200000, B=1 << 17, nn=nn, seed=48)  # hot items: long runs + stitch


@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('hinge', 'sparse_adam'), ('pointwise', 'sgd')])
def test_user_pingpong_is_bit_neutral(be, loss, opt):
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 64, U=200000, I=100000, N=600000, B=1 << 18, calls=3)  # the bandwidth-bound forms
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 32, U=943, I=1682, N=20000, B=4096, seed=57)
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 64, U=50, I=40, N=400000, B=1 << 18, seed=58)  # hot users and items: long runs + both stitch kernels
    ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 64, U=300000, I=50000, N=3000000, B=1 << 18, seed=59,
                                          options={'chunk_interactions': 1 << 20, 'overlap_prep': 1})  # several chunks, prep beside the passes
    if opt == 'adagrad':
        ec.check_user_pingpong_is_bit_neutral(be, loss, opt, 64, U=200000, I=100000, N=300000, B=65536, seed=60, with_bias_shadow=True)


@pytest.mark.parametrize('loss', ['bpr', 'hinge', 'pointwise'])
def test_single_occurrence_fast_path_is_bit_neutral(be, loss):
    # the ping-pong scope's single-occurrence fast path (option item_single_min_items forced to 1) against plain training, bit for bit:
    # a catalogue far larger than the minibatch, one where singles and runs mix, hot items beside singles, several chunks with the prep
    # beside the passes, the bias shadow
    opts = {'item_single_min_items': 1}
    n0 = be.engine.get_stat('single_minibatches')
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 64, U=200000, I=20000000, N=600000, B=1 << 18, seed=61, options=opts, calls=3)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 64, U=200000, I=300000, N=600000, B=1 << 18, seed=62, options=opts)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 32, U=5000, I=40, N=600000, B=1 << 18, seed=63, options=opts)
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 64, U=300000, I=5000000, N=3000000, B=1 << 18, seed=64,
                                          options=dict(opts, chunk_interactions=1 << 20, overlap_prep=1))
    ec.check_user_pingpong_is_bit_neutral(be, loss, 'adagrad', 64, U=200000, I=3000000, N=600000, B=1 << 18, seed=65, options=opts,
                                          with_bias_shadow=True)
    assert be.engine.get_stat('single_minibatches') > n0


def test_user_pingpong_contract(be):
    ec.check_user_pingpong_contract(be)


def test_bias_shadow_refuses_what_it_does_not_cover(be):
    ec.check_bias_shadow_refusals(be)


def test_bias_shadow_lifetime_contract(be):
    ec.check_bias_shadow_lifetime_contract(be)


def test_sampler_long_streams(be):
    # draws of more than 16384 state blocks take 256 blocks per stream (the stride-256 jump table): 12 M and 45 M words (the
    # latter also crosses that class's 40.9 M-word group limit); the class forced at a small size
    ec.check_sampler_bit_exact(be, 10 ** 6, counts=(11000000, 5, 43000000))
    with be.engine.options(mt_long_min_blocks=300):
        ec.check_sampler_bit_exact(be, 2 ** 32, counts=(624 * 256 * 2 + 11, 624 * 700))


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_train_matches_oracle(be, loss, opt):
    ec.check_train_matches_oracle(be, loss, opt, 8)


@pytest.mark.parametrize('D,B', [(32, 100), (64, 70), (128, 40), (20, 64), (3, 64), (6, 64), (256, 20)])
def test_train_other_layouts(be, D, B):
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', D, U=23, I=31, N=B + 7, B=B, epochs=1)


def test_train_heavy_duplicates_and_tiny_tables(be):
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 8, U=1, I=2, N=130, B=64, degenerate=True)
    ec.check_train_matches_oracle(be, 'pointwise', 'sparse_adam', 8, U=3, I=1, N=100, B=100, degenerate=True)
    ec.check_train_matches_oracle(be, 'adaptive_hinge', 'adagrad', 8, U=2, I=3, N=65, B=64, nn=5, degenerate=True)
    ec.check_train_matches_oracle(be, 'hinge', 'adam_dense', 4, U=5, I=4, N=1, B=256)


@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('bpr', 'sparse_adam'), ('hinge', 'adagrad'),
                                       ('pointwise', 'adagrad'), ('adaptive_hinge', 'adagrad')])
def test_train_matches_oracle_at_scale(be, loss, opt):
    """Sizes where many workgroups, grid-stride loops and multi-minibatch chunks are live:
    50k interactions, D=64, 6 minibatches of 8192 (+ a short one), duplicates everywhere.  Open loop for the RNG stream
    and the loss trajectory, closed loop (per minibatch, per element, no outlier allowance) for the tables."""
    ec.check_train_closed_loop(be, loss, opt, 64, U=3000, I=1000, N=50000, B=8192, nn=5, epochs=1)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('D', [16, 64])
def test_single_step_loss_and_gradients(be, loss, D):
    ec.check_single_step_gradients(be, loss, D)
    ec.check_single_step_gradients(be, loss, D, U=5000, I=2000, B=20000, seed=3)


@pytest.mark.parametrize('name', ec.FIXTURES)
def test_train_replays_reference_fixture(be, name):
    ec.check_replays_reference_fixture(be, GOLDEN, name)


# ---- PoolNet / ImplicitSequenceModel kernels (csrc/slk_seq.hip) ----
@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_seq_train_matches_oracle(be, loss, opt):
    ec.check_seq_train_matches_oracle(be, loss, opt, 8)


@pytest.mark.parametrize('D,L,B', [(32, 20, 10), (64, 33, 6), (128, 5, 8), (6, 300, 3), (3, 1, 16), (16, 130, 4),
                                   (64, 200, 5), (128, 200, 3), (256, 100, 2)])
def test_seq_other_layouts_and_lengths(be, D, L, B):
    ec.check_seq_train_matches_oracle(be, 'bpr', 'adagrad', D, I=47, N=B + 3, L=L, B=B, epochs=1)


@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('pointwise', 'sparse_adam'), ('adaptive_hinge', 'adagrad')])
def test_seq_train_matches_oracle_at_scale(be, loss, opt):
    """Many workgroups, several minibatches per chunk, duplicates everywhere: 600 sequences of
    length 50, D=64, minibatches of 128 (+ a short one)."""
    ec.check_seq_train_matches_oracle(be, loss, opt, 64, I=2000, N=600, L=50, B=128, nn=5, epochs=1, tol=1e-4,
                                      pad_frac=0.2)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('D', [16, 64])
def test_seq_single_step_loss_and_gradients(be, loss, D):
    ec.check_seq_single_step_gradients(be, loss, D)
    ec.check_seq_single_step_gradients(be, loss, D, I=3000, B=200, L=40, seed=3)


@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('adaptive_hinge', 'adam_dense'), ('pointwise', 'sparse_adam'),
                                      ('hinge', 'adagrad')])
def test_seq_bloom_item_layer_matches_oracle(be, loss, opt):
    """PoolNet over a BloomEmbedding item layer: small case, single-step gradients, and a case with many
    workgroups / minibatches per chunk (C4-like rows: D=64, 4 hash functions, ratio 0.2)."""
    ec.check_seq_train_matches_oracle(be, loss, opt, 8, I=60, bloom=2)
    ec.check_seq_single_step_gradients(be, loss, 64, I=3000, B=200, L=40, seed=3, bloom=4, ratio=0.2)
    ec.check_seq_train_matches_oracle(be, loss, opt, 64, I=2000, N=600, L=50, B=128, nn=5, epochs=1, tol=1e-4,
                                      pad_frac=0.2, bloom=4, ratio=0.2)


@pytest.mark.parametrize('name', ec.SEQ_FIXTURES)
def test_seq_replays_reference_fixture(be, name):
    ec.check_seq_replays_reference_fixture(be, GOLDEN, name)


# ---- BloomEmbedding layers in the BilinearNet path ----
@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('opt', ['adagrad', 'adam_dense'])
def test_item_bloom_train_matches_oracle(be, loss, opt):
    ec.check_bloom_train_matches_oracle(be, loss, opt, 8, user_bloom=0, item_bloom=4)


@pytest.mark.parametrize('ub,ib', [(4, 0), (4, 4), (2, 3), (1, 8)])
@pytest.mark.parametrize('loss,opt', [('bpr', 'adagrad'), ('adaptive_hinge', 'adam_dense')])
def test_user_and_both_bloom_train_matches_oracle(be, ub, ib, loss, opt):
    ec.check_bloom_train_matches_oracle(be, loss, opt, 8, user_bloom=ub, item_bloom=ib)


def test_bloom_c3_shape_at_scale(be):
    """C3-shaped (adaptive hinge n=5, bloom item table ratio 0.2 x 4 hashes, D=128) at a size with
    many workgroups and several minibatches per chunk."""
    ec.check_bloom_train_matches_oracle(be, 'adaptive_hinge', 'adagrad', 128, user_bloom=0, item_bloom=4, U=3000,
                                        I=5000, N=20000, B=4096, nn=5, epochs=1, ratio=0.2, tol=1e-4)
    ec.check_bloom_train_matches_oracle(be, 'bpr', 'adagrad', 64, user_bloom=4, item_bloom=4, U=3000,
                                        I=5000, N=20000, B=4096, epochs=1, ratio=0.2, tol=1e-4)


@pytest.mark.parametrize('loss', ec.ALL_LOSSES)
@pytest.mark.parametrize('ub,ib', [(0, 4), (4, 4)])
def test_bloom_single_step_loss_and_gradients(be, loss, ub, ib):
    ec.check_bloom_single_step_gradients(be, loss, 16, user_bloom=ub, item_bloom=ib)
    ec.check_bloom_single_step_gradients(be, loss, 64, user_bloom=ub, item_bloom=ib, U=4000, I=6000, B=20000, seed=3)


@pytest.mark.parametrize('name', ec.BLOOM_FIXTURES)
def test_bloom_replays_reference_fixture(be, name):
    ec.check_bloom_replays_reference_fixture(be, GOLDEN, name)


def test_pipelined_chunks_match_oracle(be):
    """Real streams: prep of chunk c+1 on the engine's second HIP stream while chunk c trains on the
    caller's.  Many chunks per call; results must match the oracle as in the single-chunk runs and,
    stronger, the single-chunk engine run BIT FOR BIT (a missing event dependency shows up here)."""
    eng = be.engine
    try:
        eng.set_option('overlap_prep', 1)
        eng.set_option('overlap_min_batch', 0)  # (by default only minibatches >= 2^16 overlap their prep)
        eng.set_option('chunk_interactions', 4096)
        for _ in range(3):
            ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 64, U=3000, I=1000, N=50000, B=1024, epochs=1, tol=1e-4)
    finally:
        eng.set_option('chunk_interactions', 1 << 23)
        eng.set_option('overlap_prep', 0)
        eng.set_option('overlap_min_batch', 1 << 16)
    for overlap in (0, 1, 2):
        ec.check_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, overlap=overlap)
        ec.check_chunking_is_bit_neutral(be, 'adaptive_hinge', 'sparse_adam', 32, overlap=overlap)
        ec.check_chunking_is_bit_neutral(be, 'hinge', 'adam_dense', 16, N=12000, chunk=2048, overlap=overlap)
        ec.check_chunking_is_bit_neutral(be, 'pointwise', 'adagrad', 64, user_bloom=2, item_bloom=4, I=5000,
                                         N=20000, B=512, overlap=overlap)
        # adaptive hinge over a bloom item table: the live occurrences are re-sorted per minibatch on the passes' stream while
        # the next chunk's sorts run on the prep stream (separate temporary storage, slk_bilinear.hip BL_LATE_SORT)
        ec.check_chunking_is_bit_neutral(be, 'adaptive_hinge', 'adagrad', 32, item_bloom=4, I=5000, N=20000, B=512,
                                         overlap=overlap)
        be.engine.set_option('adaptive_late_min_batch', 0)  # ... and over a plain table
        try:
            ec.check_chunking_is_bit_neutral(be, 'adaptive_hinge', 'adagrad', 32, N=20000, B=512, overlap=overlap)
        finally:
            be.engine.set_option('adaptive_late_min_batch', 1 << 18)
    ec.check_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, U=200000, I=50000, N=600000, B=65536, chunk=131072)
    # the cache-policy option (non-temporal hints on once-per-pass rows) is bit-neutral
    ec.check_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, U=200000, I=50000, N=600000, B=65536, chunk=131072,
                                     overlap=0, nt=0)
    ec.check_chunking_is_bit_neutral(be, 'adaptive_hinge', 'sparse_adam', 32, overlap=0, nt=15)


@pytest.mark.parametrize('n', [0, 1, 2, 5000, 65536, 65537, 1000003, (1 << 24) + 1, 30000000, (1 << 27) + 12345])
def test_device_shuffle_is_numpy_exact(be, n):
    """slk_shuffle_perm on the real device, up to 3e7 elements (every power-of-two range up to 2^25)."""
    ec.check_shuffle_matches_numpy(be, n, seed=n % 1000 + 3, burn=n % 7, rows=2 if n == 5000 else 0)


@pytest.mark.parametrize('n,band', [(1000003, 0), (1000003, 16), ((1 << 24) + 1, 4), (30000000, 0)])
def test_device_shuffle_full_sweeps_and_band_fallback(be, n, band):
    ec.check_shuffle_matches_numpy(be, n, seed=n % 1000 + 5, burn=n % 7, band=band)


# ---- Interactions.to_sequence on the device (slk_seqprep.hip) ----
@pytest.mark.parametrize('case', [
    (1, 5, 9, 'int32', 4, None, None), (200, 11, 50, 'int32', 5, 3, 2), (300, 40, 50, 'float', 4, None, 1),
    (300, 40, 50, 'int64_wide', 7, 2, 3), (300, 40, 50, 'constant', 6, None, 4), (5000, 2, 1000, 'int32', 16, 4, 1),
    (70000, 3000, 1000, 'negative', 10, None, None)], ids=lambda c: '-'.join(str(x) for x in c))
def test_device_to_sequence_matches_host(be, case):
    n, users, items, ts_mode, L, min_len, step = case
    ec.check_to_sequence(be, n, users, items, ts_mode, L, min_len, step, seed=n + L)


@pytest.mark.parametrize('case', [
    (3000000, 200000, 1000000, 'int32', 10, None, None), (3000000, 200000, 1000000, 'int64_wide', 20, 5, 3),
    (2000000, 1500000, 50000, 'float', 8, 2, 1), (10000000, 1000000, 1000000, 'int32', 50, 3, 25)],
    ids=lambda c: '-'.join(str(x) for x in c))
def test_device_to_sequence_large(be, case):
    """Millions of interactions against the loop-free numpy restatement (the host double loop takes minutes there)."""
    n, users, items, ts_mode, L, min_len, step = case
    rows = ec.check_to_sequence_large(be, n, users, items, ts_mode, L, min_len, step, seed=L)
    assert rows > 0


# ---- explicit feedback (slk_bilinear_train_explicit) ----
@pytest.mark.parametrize('loss', ec.EXPLICIT_LOSSES)
def test_explicit_train_and_gradients(be, loss):
    ec.check_explicit_train_matches_oracle(be, loss, 'adagrad', 8)
    ec.check_explicit_train_matches_oracle(be, loss, 'adam_dense', 64, U=3000, I=1000, N=50000, B=4096, epochs=1, tol=1e-4)
    ec.check_explicit_single_step_gradients(be, loss, 64, U=4000, I=6000, B=20000, seed=3)


@pytest.mark.parametrize('loss', ec.EXPLICIT_LOSSES)
def test_explicit_fused_and_staged_routes_agree(be, loss):
    ec.check_explicit_routes_agree(be, loss, 'adagrad', 64, U=30000, I=8000, N=200000, B=32768)
    ec.check_explicit_routes_agree(be, loss, 'sparse_adam', 32, U=3000, I=800, N=20000, B=4096)


@pytest.mark.parametrize('name', ec.EXPLICIT_FIXTURES)
def test_explicit_replays_reference_fixture(be, name):
    ec.check_explicit_replays_reference_fixture(be, GOLDEN, name)


def test_row_ids_beyond_24_bits(be):
    """Tables with 2^25 + 3 user rows and 2^21 + 1 item rows, ids at both ends of the range."""
    ec.check_high_row_ids(be)


def test_minibatch_of_one_interaction(be):
    """batch_size 1 (the reference crashes there: squeeze() collapses [1, D]) and a
    last minibatch of one interaction."""
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', 16, U=9, I=7, N=5, B=1, epochs=1)
    ec.check_train_matches_oracle(be, 'hinge', 'adam_dense', 8, U=9, I=7, N=65, B=64, epochs=1)


@pytest.mark.parametrize('opt', ['adagrad', 'adam_dense'])
@pytest.mark.parametrize('late_min', [0, 1 << 40])
def test_adaptive_hinge_item_side_sorted_per_minibatch_or_per_chunk(be, opt, late_min):
    """adaptive hinge's item side re-sorted per minibatch after the selection (default from 2^18 interactions per minibatch)
    or all 1+n occurrences sorted once per chunk (default below): both against the oracle"""
    be.engine.set_option('adaptive_late_min_batch', late_min)
    try:
        ec.check_train_matches_oracle(be, 'adaptive_hinge', opt, 64, U=3000, I=1000, N=30000, B=4096, nn=5, epochs=1, tol=1e-4)
    finally:
        be.engine.set_option('adaptive_late_min_batch', 1 << 18)


@pytest.mark.parametrize('D,I,B', [(64, 3, 1 << 16), (64, 50, 1 << 18), (32, 7, 40000), (128, 2, 30000)])
def test_rows_that_collect_thousands_of_occurrences(be, D, I, B):
    """popular items: one item row's occurrences fill hundreds to thousands of the item pass's tiles; k_item_stitch adds their
    partial sums.  Summed item gradients of one minibatch against the exact (float64) ones, then a training run."""
    ec.check_long_run_gradients_against_exact(be, 'bpr', D, U=5000, I=I, B=B, tol=1e-5)
    ec.check_long_run_gradients_against_exact(be, 'pointwise', D, U=3, I=I, B=B, tol=1e-5)
    ec.check_train_matches_oracle(be, 'bpr', 'adagrad', D, U=5000, I=I, N=2 * B + 50, B=B, epochs=1, degenerate=True)


@pytest.mark.parametrize('D,U,I,N,B', [(64, 3000, 1000, 50000, 8192), (64, 3000, 3, 200000, 65536), (32, 500, 40, 100000, 30000)])
def test_item_long_gate_is_bit_neutral(be, D, U, I, N, B):
    """the plain item pass for minibatches without long runs (per-chunk flags) against partials + stitch everywhere"""
    ec.check_item_long_gate_is_bit_neutral(be, 'bpr', 'adagrad', D, U, I, N, B)


@pytest.mark.parametrize('loss,opt,nn', [('bpr', 'adagrad', 1), ('hinge', 'sparse_adam', 1), ('pointwise', 'adam_dense', 1),
                                         ('adaptive_hinge', 'adagrad', 5)])
def test_item_pass_with_every_head_early_is_bit_neutral(be, loss, opt, nn):
    """k_item_pass<..., NPRE 4> (launches of up to item_lat_max_tiles tiles) against the default form, minibatches of 4096-65 536"""
    ec.check_item_long_gate_is_bit_neutral(be, loss, opt, 64, U=200000, I=50000, N=200000, B=65536, option='item_lat_max_tiles',
                                           values=(1 << 30, 0), default=2048, nn=nn)
    ec.check_item_long_gate_is_bit_neutral(be, loss, opt, 32, U=943, I=1682, N=20000, B=4096, option='item_lat_max_tiles',
                                           values=(1 << 30, 0), default=2048, nn=nn)


def test_seq_item_pass_with_every_head_early_is_bit_neutral(be):
    ec.check_seq_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, I=20000, N=2000, L=10, B=256, chunk=1 << 23, overlap=0,
                                         option=('item_lat_max_tiles', 1 << 30, 0, 2048))
    ec.check_seq_chunking_is_bit_neutral(be, 'adaptive_hinge', 'sparse_adam', 32, chunk=1 << 23, overlap=0,
                                         option=('item_lat_max_tiles', 1 << 30, 0, 2048))


def test_seq_pass_forms_are_bit_identical(be):
    """register-resident sequence pass (1, the default) and the LDS-staged pass (0): the same record, the same tables"""
    ec.check_seq_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, I=20000, N=2000, L=10, B=256, chunk=1 << 23, overlap=0,
                                         option=('seq_variant', 1, 0, 1))
    ec.check_seq_chunking_is_bit_neutral(be, 'adaptive_hinge', 'sparse_adam', 32, chunk=1 << 23, overlap=0,
                                         option=('seq_variant', 1, 0, 1))
    ec.check_seq_chunking_is_bit_neutral(be, 'bpr', 'adagrad', 64, I=50000, N=512, L=200, B=128, chunk=1 << 23, overlap=0,
                                         option=('seq_variant', 1, 0, 1))


# ---- persistent epoch kernel (csrc/slk_epoch.hip): one cooperative launch per chunk of minibatches ----
@pytest.mark.parametrize('loss', ['pointwise', 'bpr', 'hinge'])
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_bit_identical_to_launch_path(be, loss, opt):
    """C1 shape (MovieLens-100K: 943 x 1682, dim 32, minibatch 1024, the reference's test kwargs,
    tests/factorization/test_implicit.py:40-57): 64 workgroups across all 8 XCDs hand rows and records to each other
    through sc1 accesses only -- a stale read anywhere would break the bit-identity with the launch path."""
    ec.check_epoch_kernel_is_bit_identical(be, loss, opt, 32, U=943, I=1682, N=20000, B=1024, epochs=2)


@pytest.mark.parametrize('D,U,I,B,N', [(64, 100000, 50000, 4096, 40000), (64, 3000, 1000, 256, 9000), (128, 5000, 700, 2048, 10000),
                                       (20, 300, 200, 512, 3000), (32, 5, 3, 1024, 5000)])
def test_epoch_kernel_sizes_and_layouts(be, D, U, I, B, N):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', D, U=U, I=I, N=N, B=B, epochs=2)
    ec.check_epoch_kernel_is_bit_identical(be, 'hinge', 'sparse_adam', D, U=U, I=I, N=N, B=B, epochs=1)
    if (U + I) * D < 2_000_000:
        ec.check_epoch_kernel_is_bit_identical(be, 'pointwise', 'adam_dense', D, U=U, I=I, N=N, B=B, epochs=1)


def test_epoch_kernel_cooperative_launch_form(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', 32, U=943, I=1682, N=20000, B=1024, epochs=2, cooperative=1)


@pytest.mark.parametrize('loss', ['regression', 'poisson', 'logistic'])
@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_explicit_feedback_bit_identical_to_launch_path(be, loss, opt):
    """ExplicitFactorizationModel's loop at the reference's MovieLens-100K shape and default batch size (explicit.py:71)"""
    ec.check_epoch_kernel_is_bit_identical(be, loss, opt, 32, U=943, I=1682, N=20000, B=256, epochs=2)


@pytest.mark.parametrize('opt', ec.ALL_OPTS)
def test_epoch_kernel_adaptive_hinge_bit_identical_to_launch_path(be, opt):
    """adaptive hinge with the reference's default of 5 draws (implicit.py:72) at its MovieLens-100K test shape: score phase,
    the selection inside the user phase (scores written by other workgroups, read through sc1 loads), three barriers per minibatch"""
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', opt, 32, U=943, I=1682, N=20000, B=256, epochs=2)


def test_epoch_kernel_adaptive_hinge_sizes_and_layouts(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 32, U=943, I=1682, N=20000, B=1024, epochs=2)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 64, U=100000, I=50000, N=40000, B=4096, epochs=1, nn=3)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'sparse_adam', 64, U=3000, I=1000, N=9000, B=256, epochs=1, nn=20, barrier=1)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 20, U=5, I=3, N=5000, B=512, epochs=1, nn=2)
    ec.check_epoch_kernel_is_bit_identical(be, 'adaptive_hinge', 'adagrad', 32, U=943, I=1682, N=51200, B=256, epochs=1, chunk=12800)


def test_epoch_kernel_two_level_barrier(be):
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', 32, U=943, I=1682, N=20000, B=1024, epochs=2, barrier=1)
    ec.check_epoch_kernel_is_bit_identical(be, 'pointwise', 'sparse_adam', 64, U=3000, I=1000, N=9000, B=256, epochs=1, barrier=1)


def test_epoch_kernel_many_minibatches_and_chunks(be):
    """400 minibatches in 4 launches (chunks of 100): thousands of grid barriers back to back"""
    ec.check_epoch_kernel_is_bit_identical(be, 'bpr', 'adagrad', 32, U=943, I=1682, N=102400, B=256, epochs=1, chunk=25600)




@pytest.mark.parametrize('loss,opt,bloom', [('bpr', 'adagrad', 0), ('adaptive_hinge', 'sparse_adam', 0), ('pointwise', 'adam_dense', 0),
                                            ('bpr', 'adagrad', 4)])
@pytest.mark.parametrize('overlap', [0, 1])
def test_seq_chunked_and_pipelined_prep_is_bit_neutral(be, loss, opt, bloom, overlap):
    """PoolNet: several prep chunks per call, in line and with the next chunk's prep on the second stream"""
    ec.check_seq_chunking_is_bit_neutral(be, loss, opt, 16, bloom=bloom, overlap=overlap)


@pytest.mark.parametrize('D,U,B', [(64, 3, 1 << 16), (64, 50, 1 << 18), (32, 7, 40000), (128, 2, 30000), (64, 2000, 1 << 18)])
def test_users_that_collect_thousands_of_occurrences(be, D, U, B):
    """hot users (VERDICT r02 missing 3): a user's occurrences fill hundreds to thousands of the user pass's 32-position tiles;
    every tile-sized segment is walked by its own row group, k_user_stitch adds the partials.  Summed user-embedding and
    user-bias gradients of one minibatch against the exact (float64) ones; U = 2000 at 2^18: runs of ~130 next to shorter ones."""
    ec.check_long_user_run_gradients_against_exact(be, 'bpr', D, U=U, I=100000, B=B, tol=1e-5)
    ec.check_long_user_run_gradients_against_exact(be, 'pointwise', D, U=U, I=3000, B=B, tol=1e-5)


@pytest.mark.parametrize('loss,opt', [('pointwise', 'adagrad'), ('bpr', 'sparse_adam'), ('hinge', 'sgd'), ('adaptive_hinge', 'adagrad')])
def test_hot_users_closed_loop(be, loss, opt):
    """minibatches of 20 000 over 300 users (runs of ~65: some cover a tile of the user pass, some do not) against the oracle,
    step by step and element by element"""
    ec.check_train_closed_loop(be, loss, opt, 32, U=300, I=20000, N=50000, B=20000, nn=3, epochs=1, seed=8)


@pytest.mark.parametrize('D,U,I,N,B,opt', [(64, 3, 50000, 200000, 65536, 'adagrad'), (32, 40, 500, 100000, 30000, 'sparse_adam')])
def test_user_long_gate_is_bit_neutral(be, D, U, I, N, B, opt):
    ec.check_item_long_gate_is_bit_neutral(be, 'bpr', opt, D, U, I, N, B)


@pytest.mark.parametrize('D', [4, 16, 31, 64, 100, 128, 256])
def test_scores_are_the_fma_chain(be, D):
    ec.check_scores_are_the_fma_chain(be, D)


def test_fused_ranks(be):
    ec.check_fused_ranks(be)
    ec.check_fused_ranks(be, D=64, U=200, I=1500, n_rows=300, seed=9)


def test_prefetch_behind_an_inline_draw(be):
    ec.check_prefetch_behind_an_inline_draw(be)
    ec.check_prefetch_behind_an_inline_draw(be, D=64, U=200000, I=50000, N=600000, B=65536, chunk=131072)


def test_user_bias_zero_hint_is_bit_neutral(be):
    ec.check_user_bias_zero_hint_is_bit_neutral(be)
    ec.check_user_bias_zero_hint_is_bit_neutral(be, D=64, U=200000, I=50000, N=300000, B=65536)
