/*
 * spotlight_hip.h -- C ABI of libspotlight_hip.so, the MI355X (gfx950) engine behind
 * spotlight_amd's drop-in ImplicitFactorizationModel.fit()/predict().
 *
 * The reference (maciejkula/spotlight, pure Python) has no FFI seam; this boundary is
 * created INSIDE its fit()/predict().  Each entry point names the
 * reference code it replaces.  Conventions:
 *   - plain C, no torch / C++ types; every function returns 0 or a negative errno-style
 *     code (SLK_EINVAL bad argument/shape, SLK_ERANGE id >= table rows, SLK_ENOMEM,
 *     SLK_EIO HIP failure); text via slk_last_error().
 *   - pointers named d_* are DEVICE pointers owned by the caller (torch tensors'
 *     data_ptr()); h_* are host pointers.  The library owns only the opaque ctx and
 *     its scratch.  No caller pointer is retained across calls (the two training scopes,
 *     slk_bias_shadow_begin and slk_user_pingpong_begin, are the documented exceptions).
 *   - all work is enqueued on the caller's hipStream_t (`stream`, passed as void*);
 *     calls return without synchronising unless stated.
 *   - a ctx is bound to one device and is not thread-safe.
 *   - all tables are row-major fp32, rows contiguous (stride == dim); ids are int64 as on
 *     the reference path (factorization/implicit.py:202-203); tables have < 2^31 rows.
 */
#ifndef SPOTLIGHT_HIP_H
#define SPOTLIGHT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLK_ABI_VERSION 12

#define SLK_OK 0
#define SLK_EIO (-5)
#define SLK_ENOMEM (-12)
#define SLK_EINVAL (-22)
#define SLK_ERANGE (-34)

typedef struct slk_ctx slk_ctx;

/* spotlight/losses.py: pointwise_loss :18-50, bpr_loss :53-90, hinge_loss :93-124,
 * adaptive_hinge_loss :127-166 */
enum slk_loss {
    SLK_LOSS_POINTWISE = 0,
    SLK_LOSS_BPR = 1,
    SLK_LOSS_HINGE = 2,
    SLK_LOSS_ADAPTIVE_HINGE = 3,
    /* explicit feedback (spotlight/losses.py:169-244), slk_bilinear_train_explicit only */
    SLK_LOSS_REGRESSION = 4,
    SLK_LOSS_POISSON = 5,
    SLK_LOSS_LOGISTIC = 6
};

/* Optimizers the reference can instantiate (factorization/implicit.py:143-150):
 *  ADAGRAD       torch.optim.Adagrad, weight_decay == 0: row-sparse update of looked-up rows
 *                (== dense Adagrad, zero grad => zero update; == sparse=True path)
 *  SPARSE_ADAM   torch.optim.SparseAdam (sparse=True): looked-up rows only, stale moments
 *  ADAM_DENSE    torch.optim.Adam (+weight_decay=l2), the reference default: every step
 *                sweeps every row of all four tables
 *  ADAGRAD_DENSE torch.optim.Adagrad with weight_decay != 0: full sweep
 *  SGD           torch.optim.SGD(lr) with momentum == 0 and weight_decay == 0 (torch/optim/sgd.py: param.add_(grad, alpha=-lr)):
 *                stateless, touched rows only -- dense and sparse gradients give the same update; d_state1 / d_state2 unused */
enum slk_opt {
    SLK_OPT_ADAGRAD = 0,
    SLK_OPT_SPARSE_ADAM = 1,
    SLK_OPT_ADAM_DENSE = 2,
    SLK_OPT_ADAGRAD_DENSE = 3,
    SLK_OPT_SGD = 4
};

/* spotlight/layers.py:74-244 BloomEmbedding(num_embeddings, embedding_dim, compression_ratio,
 * num_hash_functions, padding_idx): the embedding of id x is the SUM of the rows
 *   row_h(x) = murmurhash3_32(int32 x, seed = seeds[h]) mod rows     (signed hash, floor-mod)
 * h = 0..n_hash-1, of a compressed table with rows = int(compression_ratio * num_embeddings);
 * x == padding_idx maps to row 0 for every h (:180-186); the inner table's padding row
 * `skip_row` (:167-169) is zero and never receives a gradient (-1: none).  Hashes are computed
 * in-kernel; the reference's [num_embeddings, n_hash] int64 hash cache (:188-198) is not used. */
typedef struct slk_bloom {
    int64_t rows;
    int32_t n_hash; /* 1..8 (layers.py:13-20 SEEDS, in order) */
    int32_t reserved;
    int64_t padding_idx;
    int64_t skip_row;
    uint32_t seeds[8];
} slk_bloom;

/* Table order everywhere = BilinearNet parameter creation order
 * (factorization/representations.py:46-59):
 *   0 user_embeddings.weight [num_users, dim]   1 item_embeddings.weight [num_items, dim]
 *   2 user_biases.weight     [num_users, 1]     3 item_biases.weight     [num_items, 1] */
/* slk_tables::flags.
 * SLK_TABLES_USER_BIAS_ZERO: the caller GUARANTEES that d_param[2] (user_biases.weight) is identically zero.  Under bpr and hinge
 * the user bias enters both scores of a pair and leaves the loss's gradient exactly (+g - g == 0: SURVEY.md 8(a) #11), so a model
 * the reference initialised (ZeroEmbedding, spotlight/layers.py:41-56) and trained with those losses keeps it at zero forever.
 * With the flag, slk_bilinear_train's pair-loss user pass (bpr / hinge, row-sparse Adagrad or plain SGD, plain tables) does not
 * fetch user biases at all -- a random 4-byte read costs a whole cache line per interaction -- and uses 0.0f: same values.  Every
 * other route ignores the flag.  This package's fit() sets it after checking the tensor on the device. */
#define SLK_TABLES_USER_BIAS_ZERO 1

typedef struct slk_tables {
    float *d_param[4];
    int64_t num_users;
    int64_t num_items;
    int32_t dim;
    int32_t flags; /* SLK_TABLES_* hints (0: none; the field was `reserved`, always 0, up to ABI 10) */
    /* NULL: plain table ([num_users|num_items, dim]).  Otherwise d_param[0] / d_param[1] is the
     * BloomEmbedding's compressed table [bloom->rows, dim] (BilinearNet's user_embedding_layer /
     * item_embedding_layer arguments, factorization/representations.py:46-56); the bias tables
     * stay [num_users] / [num_items]. */
    const slk_bloom *user_bloom;
    const slk_bloom *item_bloom;
} slk_tables;

/* Optimizer hyper-parameters are doubles because torch keeps them as Python floats and
 * forms 1-beta, bias corrections and step sizes in double before rounding to fp32. */
typedef struct slk_optim {
    int32_t kind; /* enum slk_opt */
    int32_t reserved;
    int64_t step; /* optimizer steps already taken (torch state['step']); advanced by the
                     number of minibatches processed */
    double lr, eps, beta1, beta2, weight_decay, lr_decay;
    float *d_state1[4]; /* Adagrad: state['sum'];  Adam/SparseAdam: state['exp_avg']   */
    float *d_state2[4]; /* Adam/SparseAdam: state['exp_avg_sq']; unused for Adagrad      */
} slk_optim;

int slk_abi_version(void);

/* One ctx per model per device (replaces nothing in the reference: it is the home of the
 * scratch torch would allocate per op, and of the on-device MT19937 state). */
int slk_ctx_create(slk_ctx **out, int device_id);
void slk_ctx_destroy(slk_ctx *ctx);
const char *slk_last_error(const slk_ctx *ctx); /* ctx may be NULL: last create error */

/* Tuning knobs.  None changes results: negatives, RandomState and every table and state tensor come out bit for bit the same under
 * any setting (tests/test_emu_engine.py: test_option_is_result_neutral walks the library's option table) -- with ONE exception,
 * "adaptive_late_min_batch", whose two forms sum an item row's contributions in different orders (results agree to fp32
 * rounding: both are checked against the oracle), and the two debug switches ("sort_debug", "epoch_debug").  Round 4 removed the ones whose A/Bs lost everywhere in round 3 ("first_chunk",
 * "chunk_ramp", "prep_cus", "prep_priority", "epoch_seq" with its PoolNet persistent kernel): the numbers are in
 * profiles/r03_a_*, r03_c_*, r03_w_*, r03_x_*.
 *   "chunk_interactions"  interactions per prep chunk (default 2^23)
 *   "overlap_prep"        1: the negatives + sorts of chunk c+1 run on a second HIP stream while chunk c trains (what this
 *                         package's fit() sets for its epochs: +1.5..4 % in the steady state of a run of training calls);
 *                         2: only the negatives; 0 (default of a bare ctx): everything on the caller's stream -- since ABI 11
 *                         chunk c+1's negatives + sorts are ENQUEUED before chunk c's passes (two buffer sets, one stream: the
 *                         host's one wait per chunk falls while the GPU prepares the next chunk), and the negatives of a call
 *                         of several chunks are ONE draw in front of the first chunk (fewer than 2^30 draws; it is one
 *                         contiguous draw of the stream however the call is chunked)
 *   "overlap_min_batch"   the prep overlaps the passes only for minibatches of at least this size (default 2^16)
 *   "prefetch_wait"       measurement switch: 1 = slk_bilinear_prefetch's chunk waits for everything `stream` holds, as it did up
 *                         to ABI 8 (default 0: it runs beside the passes `stream` still holds)
 *   "item_grid_mult"      item pass: workgroups per CU (default 128)
 *   "user_grid_mult"      other row passes: workgroups per CU (default 8, grid-stride beyond); the user passes never launch more
 *                         than their kernel holds resident per CU (72 VGPRs = 7 for the SparseAdam form: a surplus workgroup per
 *                         CU ran as a second, partial round and cost the pass 13 %, profiles/r05_w_*)
 *   "epoch_kernel" (0/1), "epoch_max_batch", "epoch_dense_elems", "epoch_max_grid", "epoch_barrier", "epoch_cooperative"
 *                         the persistent epoch kernel of slk_bilinear_train / _explicit (csrc/slk_epoch.hip)
 *   "epoch_adaptive"      1 (default): adaptive hinge takes the persistent kernel too (score phase + in-phase selection),
 *   "epoch_adaptive_max_batch"  for minibatches up to this size (default 1024)
 *   "explicit_fused"      explicit feedback: score + loss inside the user pass (default 1)
 *   "adaptive_late_min_batch"  adaptive hinge on a plain item table: from this minibatch size the live occurrences are
 *                         re-sorted per minibatch after the selection (default 2^18; below, all 1+n are sorted per chunk).
 *                         The one option that is not bit-neutral: a row's live contributions are summed in the order of the
 *                         compacted list instead of the (1+n)-slot order -- 1-ulp differences, deterministic either way.
 *   "item_long_gate"      1 (default): minibatches in which no item row's occurrences fill a whole 64-position tile of the
 *                         item pass take the plain pass; 0: always the partial-writing pass + stitch kernel (same results).
 *                         The same switch gates the user pass's long-run form (hot users: runs that fill a 32-position tile).
 *   "item_lat_max_tiles"  item pass launches of up to this many 64-occurrence tiles (default 2048; PoolNet: a quarter) load every
 *                         head's row + state with the record gather (k_item_pass<..., NPRE 4>: -10..-19 % on minibatches of
 *                         4096-65 536, profiles/r03_y_*); 0: never
 *   "user_lat_max_batch"  minibatches up to this size (default 2^17) take the latency-bound form of the pair-mode user pass
 *   "sort_big_min"        radix sort (csrc/slk_sort.hip): sorts of at least this many pairs (default 2^20) use tiles of 512 threads x
 *                         16 keys, smaller ones 256 x 16 (test hook: both shapes at any size);  "sort_debug": measurement only,
 *                         1-3 skip parts of a pass (profiles/r04_b_*; the output is then not sorted)
 *   "shuffle_band"        slk_shuffle_perm: 1 banded acceptance decisions (default), 0 full fixpoint sweeps,
 *                         > 1 a band that many times too narrow (test hook for the fall-back)
 *   "mt_long_min_blocks"  the MT19937 generator (csrc/slk_rng.hip): draws of at least this many 624-word state blocks (default
 *                         16385 = more than 10.2 M words) use 256 blocks per stream and the stride-256 jump table, smaller
 *                         ones 64 (test hook: the long class at a small size)
 *   "epoch_max_grid"      workgroups of the persistent kernel, at most one per CU -- values above 1024 lift that limit (a
 *                         measurement switch: several one-wave workgroups per CU, profiles/r05_b_persistent_kernel_wide_grid_*)
 *   "nt", "seq_variant"   cache-policy bits of the passes (non-temporal accesses: 1 user rows + state, 2 item rows + state, 8 key /
 *                         payload streams, 16 the user pass's record stores, 32 the item pass's record loads; default 3);
 *                         PoolNet sequence-pass variant
 *   "record_nt_min_bytes" slk_bilinear_train: a minibatch whose pre-step user-row records take at least this many bytes (default
 *                         192 MB: they would fill the 256 MB Infinity Cache) stores them non-temporally, as "nt" bit 16 does for
 *                         every size; 0: never (profiles/r06_mall_ab.jsonl)
 *   "user_bias_zero_hint" 1 (default): slk_tables::flags' SLK_TABLES_USER_BIAS_ZERO is honoured; 0: the user biases are fetched
 *                         regardless (A/B and test switch: same results)
 *   "item_single_min_items"  slk_bilinear_train inside a user-row ping-pong scope, row-sparse Adagrad, item tables of at least this
 *                         many rows (default 2^24; 0: never): an item that occurs ONCE in its minibatch -- on a catalogue far larger
 *                         than a minibatch nearly every one -- is updated by the user pass, which already holds its row, dL/dscore
 *                         and the pre-step user row; the item pass skips it (no re-read of the row, no gather of the user row).
 *                         Same operations in the same order: bit-identical tables.  Minibatches that take the latency-bound or
 *                         the long-run form of the user pass leave every item to the item pass.
 *   "user_grid_own_occ"   0 (default): the user pass launches at most as many workgroups per CU as the form with the most registers
 *                         (plain, latency-bound, long runs, ping-pong) holds resident; 1: as the form it launches does (measured:
 *                         no gain at C2, -12 % on the C5 shard, profiles/r06_w_*; same tables, the fp32 loss sums associate by
 *                         workgroup) */
int slk_ctx_set_option(slk_ctx *ctx, const char *name, int64_t value);
/* The current value of an option (ABI 9): lets a caller change an option for one piece of work and restore it afterwards --
 * a ctx is shared by every model of a process on its device (spotlight_amd/_native.py: `with engine.options(...)`). */
int slk_ctx_get_option(slk_ctx *ctx, const char *name, int64_t *value);
/* Diagnostics of the last calls (no effect on results): "shuffle_sweeps" / "shuffle_fallbacks" (slk_shuffle_perm: full
 * fixpoint sweeps run, ranges that left the band and were redone), "epoch_refused" (the persistent launch was refused once),
 * "user_long_launches" / "item_long_launches" (launches of the long-run forms of the two passes since the ctx was created),
 * "overlapped_chunks" (chunks whose negatives + sorts ran on the ctx's second stream beside the passes of the chunk before),
 * "prefetched_chunks" (first chunks prepared ahead by slk_bilinear_prefetch that a training call took over),
 * "shadowed_calls" (slk_bilinear_train calls that ran on the item-bias shadow of slk_bias_shadow_begin),
 * "pingpong_calls" (slk_bilinear_train calls that ran on the user-row ping-pong of slk_user_pingpong_begin),
 * "single_minibatches" (minibatches whose once-only items were updated by the user pass: option "item_single_min_items"),
 * "prefetch_pending" (what the last slk_bilinear_prefetch left for the next training call: 0 nothing -- it was a no-op --,
 * 1 the first chunk, 2 the first chunk and the negatives of the whole call). */
int slk_ctx_get_stat(slk_ctx *ctx, const char *name, int64_t *value);

/* numpy RandomState.set_state()/get_state() hand-over of the MT19937 stream the reference
 * draws shuffles and negatives from (torch_utils.py:46-47, sampling.py:34).  h_key is
 * uint32[624].  get_state synchronises the stream last used by this ctx. */
int slk_rng_set_state(slk_ctx *ctx, const uint32_t *h_key, int32_t pos);
int slk_rng_get_state(slk_ctx *ctx, uint32_t *h_key, int32_t *pos);
/* The same hand-over, waiting only for the LAST DRAW of negatives this ctx enqueued (a training call's, slk_sample_items')
 * instead of the whole stream: a training call draws a chunk's negatives ahead of its passes, so the position the next
 * epoch's `shuffle` continues from (spotlight/factorization/implicit.py:213-216 -> torch_utils.py:46-47) is known while the
 * epoch's last passes are still running -- fit() draws that shuffle on a second ctx / stream beside them.  Falls back to
 * slk_rng_get_state when nothing was drawn since slk_rng_set_state / slk_shuffle_perm.  Error flags: both calls REPORT AND
 * CLEAR the ctx's sticky device-side flags (sampler ran out of words, abandoned persistent launch, abandoned sort look-back)
 * -- each is reported exactly once, by whichever of the two calls reads it first; a flag raised by a kernel that is still
 * running when slk_rng_get_state_sampled returns is reported by the next slk_rng_get_state (this package's fit() calls it,
 * via check(), behind every epoch's loss read-back). */
int slk_rng_get_state_sampled(slk_ctx *ctx, uint32_t *h_key, int32_t *pos);

/* spotlight/sampling.py:8-36 sample_items(num_items, shape, random_state): `count` uniform
 * ids in [0, num_items) by numpy's masked rejection over the ctx's MT19937 stream,
 * bit-exact, written row-major as int64 to d_out.  1 <= num_items <= 2^32. */
int slk_sample_items(slk_ctx *ctx, int64_t num_items, int64_t count, int64_t *d_out, void *stream);

/* The minibatch loop of one epoch of ImplicitFactorizationModel.fit() after the shuffle
 * (factorization/implicit.py:223-243): contiguous minibatches of `batch_size` over the n
 * (user, item) pairs (short last batch); per minibatch: negatives (drawn from the ctx RNG
 * exactly as _get_negative_prediction / _get_multiple_negative_predictions do, :254-275,
 * or taken from d_neg_in), BilinearNet forward for positives and negatives
 * (representations.py:61-91), the loss (losses.py), backward with duplicate rows summed,
 * and the optimizer update (torch.optim.*), in place on tables/optimizer state.
 *   d_neg_in   NULL, or n*nn int64 negatives to use instead of sampling (nn = n_neg for
 *              adaptive hinge, else 1), minibatch-major in the reference's draw order
 *   d_neg_out  NULL, or receives the n*nn negatives used
 *   d_mb_loss  float[ceil(n / batch_size)]: loss.item() of every minibatch (:240)
 * optim->step is advanced on return.  ids are NOT range-checked here (the reference checks
 * on the host, :161-182; so does spotlight_amd). */
int slk_bilinear_train(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                       const int64_t *d_users, const int64_t *d_items, int64_t n,
                       int64_t batch_size, int32_t loss, int32_t n_neg, const int64_t *d_neg_in,
                       int64_t *d_neg_out, float *d_mb_loss, void *stream);

/* Allocates every scratch buffer a later slk_bilinear_train call of this shape (same tables,
 * optimizer kind, n, batch_size, loss, n_neg) will need, so that the training call itself
 * performs no allocation (torch's caching allocator plays this role for the reference).  Optional:
 * slk_bilinear_train grows its scratch on demand. */
/* ExplicitFactorizationModel.fit's minibatch loop (spotlight/factorization/explicit.py:213-236) for one
 * epoch of already shuffled (user, item, rating) triples: forward (BilinearNet, exp() of the score under
 * the poisson loss), regression / poisson / logistic loss (losses.py:169-244), backward with duplicate
 * rows summed, optimizer step.  No negatives are drawn; d_mb_loss[m] = loss.item() of minibatch m.
 * predict() is slk_bilinear_predict followed by exp (poisson) / sigmoid (logistic) (explicit.py:277-282). */
int slk_bilinear_train_explicit(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                                const int64_t *d_users, const int64_t *d_items, const float *d_ratings,
                                int64_t n, int64_t batch_size, int32_t loss, float *d_mb_loss, void *stream);

/* The FIRST chunk of the next slk_bilinear_train call with these very arguments, prepared NOW (ABI 7): its negatives and
 * sorts -- and the negatives of the call's OTHER chunks too, when the call draws fewer than 2^30 of them: one contiguous draw,
 * after which slk_rng_get_state_sampled returns the stream position behind the whole call -- run on the ctx's second stream
 * beside whatever `stream` still holds -- in this package's fit() the last passes of the
 * epoch before, as soon as the next epoch's shuffled ids exist (spotlight/factorization/implicit.py:212-221 is the loop being
 * pipelined; the draws are the ones `sample_items` would make first, sampling.py:34).  h_key (uint32[624]) / pos: optional --
 * the MT19937 state the call's draws start from, written without waiting for the ctx's stream (the caller has waited for the
 * last draw: slk_rng_get_state_sampled).  The next slk_bilinear_train must be that call (same ids, n, batch_size, loss,
 * n_neg, no d_neg_in): anything else is refused -- the prepared draws are consumed -- until slk_rng_set_state (a call that
 * draws nothing -- d_neg_in given, explicit feedback -- drops the prepared chunk instead).  d_users / d_items must be COMPLETE
 * when this call is made (written by work the caller has waited for): the prepared chunk does not wait for `stream`, so that it
 * runs beside the passes `stream` still holds, not behind them (with "overlap_prep" 2 it does wait: sorts on `stream` read the
 * draws of the running call).  A no-op for
 * calls that would not pipeline their prep (one chunk, minibatches below "overlap_min_batch", the persistent route, option
 * "overlap_prep" 0). */
int slk_bilinear_prefetch(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, const int64_t *d_users,
                          const int64_t *d_items, int64_t n, int64_t batch_size, int32_t loss, int32_t n_neg,
                          const uint32_t *h_key, int32_t pos, void *stream);
int slk_bilinear_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t n,
                         int64_t batch_size, int32_t loss, int32_t n_neg, void *stream);

/* Item-bias shadow of a training scope (ABI 10).  The reference keeps the item biases as an nn.Embedding of dim 1
 * (spotlight/factorization/representations.py:80-91) and Adagrad's `sum` for them as a second tensor: two 4-byte scalars in two
 * arrays.  On a table of 10^8 rows every occurrence then costs the owner pass two cache-line reads and two sector writes for
 * 8 bytes.  Between _begin and _end the engine trains on an interleaved copy {bias, sum} per item (8 bytes per row in the ctx;
 * one line per occurrence) -- tables->d_param[3] and optim->d_state1[3] are STALE meanwhile and rewritten by _end; every
 * slk_bilinear_train / _train_explicit call with these very pointers takes the launch path and uses the copy, and so do the
 * row-sharded calls on a rank's local tables (slk_shard_gather reads the biases it sends from the copy, slk_shard_item_pass
 * updates it; the other slk_shard_* calls do not touch item biases).  Row-sparse
 * Adagrad over a plain item table only (SLK_EINVAL otherwise).  Same arithmetic: tables bit-identical to training without it.
 * Every other call that names the shadowed bias array (predict, scores, ranks, the PoolNet calls) is refused with
 * SLK_EINVAL until _end.  _begin, the training calls and _end are ordered by the caller: the same stream, or events between
 * them.  One scope per ctx; the copy's storage stays with the ctx for the next scope (slk_ctx_destroy frees it; an open scope is
 * NOT written back by it).  _end without an open scope is a no-op.
 * What this package's fit() does for item tables of >= 2^24 rows.
 *
 * LIFETIME CONTRACT (ABI 11) -- the ONE place where the library keeps caller pointers across calls (SURVEY.md 8(b) "no pointer
 * is retained across calls except inside ctx scratch"): from _begin to _end / _abort the ctx holds tables->d_param[3] and
 * optim->d_state1[3], and _end WRITES through them.  The caller keeps both arrays alive and in place for the whole scope.
 *   - _end closes the scope on every return, error returns included (the arrays then keep their _begin values).
 *   - slk_bias_shadow_abort closes it WITHOUT writing: for a caller whose arrays are gone (the training of the scope is lost).
 *   - a training call (slk_bilinear_train / _train_explicit / _prefetch) that names the shadowed bias array with ANOTHER
 *     d_state1[3] (the optimizer state tensor was swapped inside the scope), or that names OTHER tables altogether, is refused
 *     with SLK_EINVAL: the scope belongs to one model's training.
 *   - slk_ctx_destroy with an open scope does not write back (it cannot know the arrays still exist) and says so on stderr. */
int slk_bias_shadow_begin(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, void *stream);
int slk_bias_shadow_end(slk_ctx *ctx, void *stream);
int slk_bias_shadow_abort(slk_ctx *ctx);

/* User-row ping-pong of a training scope (ABI 12).  Within a minibatch every score and every gradient is formed from PRE-STEP
 * rows (autograd keeps the gathered user rows alive as saved tensors until backward: spotlight/factorization/representations.py:
 * 61-91 under implicit.py:229-243), while the two ownership passes update rows in place -- so the user pass used to save every
 * position's pre-step user row in a record for the item pass (4 * dim bytes written per interaction).  Between _begin and _end
 * the user embedding table exists TWICE (the second copy, num_users * dim * 4 bytes, and one byte per user -- which copy holds
 * the user's current row -- live in the ctx): the user pass writes a user's updated row to the copy that does not hold the
 * current one and flips the byte, the item pass gathers the pre-step row from where it still stands, and no record is written.
 * Same arithmetic in the same order: tables bit-identical to training without the scope.
 *   - tables->d_param[0] is a MIX of current and superseded rows inside the scope; _end copies the rows whose current copy is
 *     the ctx's back into it.  Every call that names it other than slk_bilinear_train / _prefetch / _reserve (predict, scores,
 *     ranks, the row-sharded calls) is refused with SLK_EINVAL until then.
 *   - slk_bilinear_train inside the scope: the pair losses (pointwise, bpr, hinge) over plain tables with a row-sparse optimizer
 *     (Adagrad, SparseAdam, SGD), any minibatch size, on the launch path (not the persistent kernel).  Adaptive hinge,
 *     explicit feedback, dense optimizers and other tables are refused with SLK_EINVAL -- the scope belongs to one model's
 *     training.  optimizer state and the three other tables are used in place as always.
 *   - lifetime: as for the item-bias shadow -- the ctx holds tables->d_param[0] from _begin to _end / _abort and _end WRITES
 *     through it; _end closes the scope on every return; slk_user_pingpong_abort closes it without writing (the training of
 *     the scope is lost: the caller's table is left a mix of rows); slk_ctx_destroy with an open scope says so on stderr.
 *   - _begin, the training calls and _end are ordered by the caller (same stream, or events).  One scope per ctx; may be open
 *     together with an item-bias shadow.  SLK_ENOMEM when the second copy does not fit: train without it.  The second copy's
 *     storage stays with the ctx for the next scope (as all ctx scratch does; slk_ctx_destroy frees it).
 * What this package's fit() does for minibatches of >= 2^17 interactions ("statistic" pingpong_calls counts the calls). */
int slk_user_pingpong_begin(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, void *stream);
int slk_user_pingpong_end(slk_ctx *ctx, void *stream);
int slk_user_pingpong_abort(slk_ctx *ctx);

/* ImplicitFactorizationModel.predict (factorization/implicit.py:277-311 with
 * _components.py:8-25): d_out[k] = score(user_k, item_k); n_users == 1 broadcasts the user;
 * d_items == NULL means items 0..n-1. */
int slk_bilinear_predict(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_users,
                         int64_t n_users, const int64_t *d_items, int64_t n, float *d_out,
                         void *stream);

/* ---------------------------------------------------------------------------------------
 * PoolNet / ImplicitSequenceModel (spotlight/sequence/implicit.py, representation='pooling').
 * `tables` uses d_param[1] = item_embeddings.weight [num_items, dim] and d_param[3] =
 * item_biases.weight [num_items]; d_param[0], d_param[2] and num_users are ignored, as are
 * optim->d_state*[0] and [2].  padding_idx: the item row that never receives a gradient
 * (nn.Embedding padding_idx; PoolNet uses 0, sequence/representations.py:13,66-74), -1 = none.
 *
 * slk_poolnet_train: the minibatch loop of one epoch of ImplicitSequenceModel.fit() after the
 * shuffle (sequence/implicit.py:213-255) over d_sequences[n_seq][seq_len] (int64, left-padded
 * with 0): contiguous minibatches of `batch_size` sequences; per minibatch the negatives
 * randint(0, num_items, (B, L)) -- or ((n*B), L) for adaptive hinge, viewed (n, B, L), :266-286 --
 * from the ctx RNG or d_neg_in (same layout, minibatch after minibatch), PoolNet
 * user_representation + forward for the sequence itself and the negatives
 * (sequence/representations.py:76-144), the masked loss (losses.py, mask = sequence != 0),
 * backward and the optimizer update.  d_mb_loss[ceil(n_seq / batch_size)] = loss.item(). */
int slk_poolnet_train(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, int64_t padding_idx,
                      const int64_t *d_sequences, int64_t n_seq, int64_t seq_len, int64_t batch_size,
                      int32_t loss, int32_t n_neg, const int64_t *d_neg_in, int64_t *d_neg_out,
                      float *d_mb_loss, void *stream);

/* Pre-allocates the scratch of a later slk_poolnet_train of the same shape (no hipMalloc in the epoch loop). */
int slk_poolnet_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t n_seq,
                        int64_t seq_len, int64_t batch_size, int32_t loss, int32_t n_neg, void *stream);

/* ImplicitSequenceModel.predict (sequence/implicit.py:288-340): d_out[k] = score of item
 * d_items[k] (NULL: item k) as the next item after d_sequence[seq_len]. */
int slk_poolnet_predict(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_sequence,
                        int64_t seq_len, const int64_t *d_items, int64_t n, float *d_out, void *stream);

/* spotlight/torch_utils.py:35-52 (shuffle): d_perm_out[0..n) = the array numpy's legacy
 * RandomState.shuffle(arange(n)) produces from the ctx RNG state, bit for bit (Fisher-Yates with
 * masked-rejection draws over the MT19937 stream), computed on the GPU; the RNG state afterwards is
 * numpy's.  Synchronises the stream once per power-of-two range of the draw (a few dozen times).
 * slk_gather_rows_i64: d_dst[r][:] = d_src[d_perm[r]][:] (the x[shuffle_indices] that follows). */
int slk_shuffle_perm(slk_ctx *ctx, int64_t n, int64_t *d_perm_out, void *stream);
int slk_gather_rows_i64(slk_ctx *ctx, const int64_t *d_src, const int64_t *d_perm, int64_t n, int64_t row_len,
                        int64_t *d_dst, void *stream);
/* The same x[shuffle_indices] for fit()'s TWO id arrays at once (torch_utils.py:46-52 applies one permutation to every
 * array; factorization/implicit.py:212-215 passes user_ids and item_ids): slk_pack_id_pairs narrows the ids to 32 bits and
 * interleaves them, d_pairs[2 r] = d_users[r], d_pairs[2 r + 1] = d_items[r] (once per fit(); ids < 2^32), and
 * slk_gather_id_pairs writes d_users_out[r] = user of pair d_perm[r], d_items_out[r] = its item -- one random 8-byte read
 * per interaction instead of two, the permutation read once. (ABI 8) */
int slk_pack_id_pairs(slk_ctx *ctx, const int64_t *d_users, const int64_t *d_items, int64_t n, uint32_t *d_pairs,
                      void *stream);
int slk_gather_id_pairs(slk_ctx *ctx, const uint32_t *d_pairs, const int64_t *d_perm, int64_t n, int64_t *d_users_out,
                        int64_t *d_items_out, void *stream);

/* Interactions.to_sequence (spotlight/interactions.py:170-266): the interactions ordered by
 * np.lexsort((timestamps, user_ids)) and cut, per user, into left-zero-padded windows of
 * max_sequence_length items ending at positions count, count - step_size, ... of the user's history
 * (newest window first, users ascending); windows shorter than min_length are dropped (pass 1 for
 * min_sequence_length=None).  d_timestamps: int64 (ts_kind 0) or float64 (ts_kind 1; NaN last).
 * slk_to_sequence_plan sorts and counts (synchronises; *num_sequences_out is a HOST int64), the
 * caller allocates, slk_to_sequence_fill writes d_sequences[num_sequences][max_sequence_length]
 * (int32, as the reference) and d_sequence_users[num_sequences].  num_users bounds the user ids
 * (<= 0: unknown, 32 bits sorted). */
int slk_to_sequence_plan(slk_ctx *ctx, const int64_t *d_users, const int64_t *d_items, const void *d_timestamps,
                         int32_t ts_kind, int64_t n, int64_t num_users, int32_t max_sequence_length,
                         int32_t step_size, int32_t min_length, int64_t *num_sequences_out, void *stream);
int slk_to_sequence_fill(slk_ctx *ctx, int32_t *d_sequences, int32_t *d_sequence_users, void *stream);

/* ---------------------------------------------------------------------------------------
 * Embedding front-end for encoders whose body stays on stock PyTorch (LSTMNet / CNNNet /
 * MixtureLSTMNet, spotlight/sequence/representations.py:147-596): the lookups
 * `self.item_embeddings(ids)` / `self.item_biases(ids)` and their autograd backward.
 * slk_embedding_forward: d_out[k][:] = d_weight[d_ids[k]][:] (layers.py:23-56), or, with a bloom
 *   descriptor, the sum of the n_hash hashed rows (layers.py:236-242).
 * slk_embedding_backward_plan / _fill: gradient of that lookup w.r.t. the table from
 *   d_grad_out[n][dim]: per distinct row the sum of its gradient rows in ascending lookup order
 *   (torch's CPU embedding backward order; no atomics), rows equal to padding_idx (bloom: skip_row)
 *   excluded like nn.Embedding(padding_idx=...).  Dense output: pass d_grad_dense[rows][dim] (fully
 *   overwritten).  Coalesced COO output (sparse=True layers): pass num_rows_out to the plan (a HOST
 *   int64; synchronises), allocate, then pass d_rows_out[num_rows] (ascending) and
 *   d_values_out[num_rows][dim] with d_grad_dense = NULL. */
int slk_embedding_forward(slk_ctx *ctx, const float *d_weight, int64_t rows, int32_t dim, const slk_bloom *bloom,
                          const int64_t *d_ids, int64_t n, float *d_out, void *stream);
int slk_embedding_backward_plan(slk_ctx *ctx, int64_t rows, int32_t dim, const slk_bloom *bloom, int64_t padding_idx,
                                const int64_t *d_ids, int64_t n, int64_t *num_rows_out, void *stream);
int slk_embedding_backward_fill(slk_ctx *ctx, const float *d_grad_out, float *d_grad_dense, int64_t *d_rows_out,
                                float *d_values_out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Evaluation side of the path (spotlight/evaluation.py:9-109: mrr_score / sequence_mrr_score call
 * predict() once per user or sequence and rank on the host with scipy.stats.rankdata).
 * slk_bilinear_scores: d_out[r * num_items + i] = score of item i for user d_users[r] -- the rows
 * predict(user) returns (factorization/implicit.py:277-311), a tile of users per pass over the item
 * table.  slk_poolnet_scores: the same for sequences d_sequences[n_seq][seq_len]
 * (sequence/implicit.py:288-340).  slk_rank_targets: first d_scores[row][item] = -FLT_MAX for the
 * (row, item) pairs to exclude (the reference's predictions[train items] = FLOAT_MAX on the negated
 * scores), then d_rank_out[t] = rankdata(-d_scores[row_t])[item_t] with 'average' ties. */
int slk_bilinear_scores(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_users, int64_t n_users,
                        float *d_out, void *stream);
int slk_poolnet_scores(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_sequences, int64_t n_seq,
                       int64_t seq_len, float *d_out, void *stream);
int slk_rank_targets(slk_ctx *ctx, float *d_scores, int64_t n_rows, int64_t num_items,
                     const int64_t *d_exc_rows, const int64_t *d_exc_items, int64_t n_exc,
                     const int64_t *d_tgt_rows, const int64_t *d_tgt_items, int64_t n_tgt,
                     double *d_rank_out, void *stream);
/* Fused ranking (ABI 7): the ranks WITHOUT a score matrix -- what mrr_score / sequence_mrr_score need of
 * `predictions = -model.predict(u); predictions[train items] = FLOAT_MAX; rankdata(predictions)[test items]`
 * (spotlight/evaluation.py:39-52, 88-106).  A GROUP is one user (slk_bilinear_rank: d_group_users[n_groups]) or one
 * sequence (slk_poolnet_rank: d_group_sequences[n_groups][seq_len]); a ROW is one held-out item of a group: its group
 * d_row_group[r] and the item d_row_target[r].  d_exc_off[n_groups + 1] / d_exc_items: per group the items pushed to the end
 * of its ranking (CSR; every item at most once per group; NULL: none).  d_rank_out[r] = the 'average' rank of the row's
 * target (1 = best).  The scores are the ones slk_bilinear_predict / slk_poolnet_predict return, bit for bit: the item
 * table streams ONCE per 64 rows through the matrix cores (exact-fp32 MFMA) and every score is compared with its row's
 * target score as it leaves the accumulators. */
int slk_bilinear_rank(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_group_users, int64_t n_groups,
                      const int64_t *d_row_group, const int64_t *d_row_target, int64_t n_rows,
                      const int64_t *d_exc_off, const int64_t *d_exc_items, double *d_rank_out, void *stream);
int slk_poolnet_rank(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_group_sequences, int64_t n_groups,
                     int64_t seq_len, const int64_t *d_row_group, const int64_t *d_row_target, int64_t n_rows,
                     const int64_t *d_exc_off, const int64_t *d_exc_items, double *d_rank_out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Row-sharded BilinearNet training (the reference has no multi-GPU code, the
 * boundary is fixed by BASELINE.json's north star).  One process per GPU; user and item
 * tables (+ biases, optimizer state) are row-sharded cyclically: owner(row) = row % world,
 * local row = row / world; `tables` passed below are the LOCAL shards.  A rank processes the
 * interactions of every global minibatch whose user it owns.  The host (torch.distributed
 * all_to_all_single = RCCL over xGMI) runs the exchanges between these calls.
 *
 * Per CHUNK of M minibatches, each cut into S slices by user (unit t = minibatch * S + slice):
 *   slk_shard_chunk_begin  -> d_send_ids[2n] (owner-local item rows, int32, grouped
 *                             [owner][unit]), d_send_counts[world * M * S] (device, int64,
 *                             [owner][unit])
 *        a2a: counts (M*S per peer); the ONE host synchronisation of the chunk; a2a: ids
 *   slk_shard_chunk_commit    both count matrices (host) + the received ids ([source][unit])
 * Per unit t (no host synchronisation: every split size is known from the matrices):
 *   slk_shard_gather       owner: the rows of unit t's requests into d_rows_out ([source] segments)
 *        a2a: rows back to the requesters
 *   slk_shard_user_pass    forward/loss/backward/user update (factorization/implicit.py:229-243
 *                          restricted to this rank's users of unit t); the gradient of a lookup goes
 *                          to the slot of d_grad_out its row came in at in d_rows_in
 *        a2a: gradients to the owners
 * Per minibatch:
 *   slk_shard_item_pass    owner: per unique item row, sum of the gradients received for the S
 *                          units of the minibatch (d_grad_in: the units' regions one after the other),
 *                          then ONE optimizer update (duplicates summed before the update, as autograd
 *                          does); advances optim->step.
 *
 * Exchange buffers (rows and gradients alike) are arrays of SLOTS in blocks of SLK_SHARD_BLOCK = 64:
 * a block is 64 rows of `dim` floats followed by the 64 scalars (bias / bias gradient) of those rows, so
 * a row of a dim-64 table is one aligned 256-B line on both sides of the wire.  Inside a unit's buffer
 * every peer's segment starts on a block boundary: a peer that exchanges c lookups of the unit owns
 * ceil(c / 64) * 64 slots = slk_shard_buffer_floats(dim, c) floats -- the split sizes of the all-to-all,
 * in [peer] order; the slots past c are never read.  These entry points: pointwise/bpr/hinge (one negative per
 * interaction); adaptive hinge: the *_adaptive entry points below. */
#define SLK_SHARD_BLOCK 64
typedef struct slk_shard {
    int32_t world, rank;
    int64_t num_items_global; /* negatives are drawn in [0, num_items_global) (sampling.py:34) */
    int64_t global_batch;     /* unused since ABI 3 (passed per minibatch to slk_shard_user_pass) */
} slk_shard;

/* floats of a buffer segment that holds `slots` lookups (rounded up to whole blocks) */
int64_t slk_shard_buffer_floats(int32_t dim, int64_t slots);
/* Optional: allocates the scratch of chunks of up to n local interactions whose owner side receives up to n_recv
 * lookups now, so that a loop whose first chunk is its largest does not allocate (and synchronise the device) inside it. */
int slk_shard_reserve(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh, int64_t n, int64_t n_recv);
/* d_users_local: LOCAL user rows (user / world) of this rank's n interactions of the chunk;
 * d_items: GLOBAL item ids; h_mb_off[M + 1] (host): minibatch boundaries inside [0, n];
 * negatives: sampled from the ctx RNG over the global item range (one contiguous draw for the
 * chunk), or d_neg_in[n]. */
int slk_shard_chunk_begin(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                          const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                          const int64_t *h_mb_off, int32_t n_minibatches, int32_t n_slices,
                          const int64_t *d_neg_in, int64_t *d_neg_out, int32_t *d_send_ids,
                          int64_t *d_send_counts, void *stream);
/* h_send_counts / h_recv_counts: [world][M * S] (host): lookups this rank sends to / receives
 * from each peer per unit; d_recv_ids: the ids received, grouped [source][unit]. */
int slk_shard_chunk_commit(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                           const int64_t *h_send_counts, const int64_t *h_recv_counts,
                           const int32_t *d_recv_ids, void *stream);
int slk_shard_gather(slk_ctx *ctx, const slk_tables *local, int32_t unit, float *d_rows_out, void *stream);
/* d_loss_out[0] (+)= (sum of this rank's per-interaction losses of the unit) / global_batch;
 * a minibatch's loss.item() is the sum over its slices and over the ranks. */
int slk_shard_user_pass(slk_ctx *ctx, const slk_tables *local, const slk_optim *optim,
                        const slk_shard *sh, int32_t unit, int64_t global_batch, int32_t loss,
                        const float *d_rows_in, float *d_grad_out, float *d_loss_out, int32_t accumulate,
                        void *stream);
int slk_shard_item_pass(slk_ctx *ctx, const slk_tables *local, slk_optim *optim, int32_t minibatch,
                        const float *d_grad_in, void *stream);

/* Adaptive hinge on the row-sharded path (ABI 9; spotlight/factorization/implicit.py:266-275 + spotlight/losses.py:127-166).
 * The reference draws B * n negatives in ONE randint call, scores flat entry k with user k / n, and views the scores as
 * [n, B]: column c's candidates are the flat entries {r B + c} -- draws that belong to OTHER interactions, which on this path
 * live on other ranks.  The step therefore gets a score phase and a selection in front of the user pass:
 *   slk_shard_chunk_begin_adaptive   as slk_shard_chunk_begin with 1 + n_neg lookups per interaction: d_neg_in[n * n_neg]
 *                                    (the draws [k * n_neg, (k + 1) * n_neg) of interaction k -- its slice of its minibatch's
 *                                    flat draw -- or NULL: drawn from the ctx RNG), d_mb_pos[n]: the interaction's position
 *                                    inside its GLOBAL minibatch, d_send_ids[n * (1 + n_neg)]
 *        [count / id exchange, slk_shard_chunk_commit, per unit slk_shard_gather + the row all-to-all, as before]
 *   per unit:      slk_shard_score_pass       d_scores[pos * (1 + n_neg) + s] = score of pair s of the interaction at global
 *                                             position pos (s = 0: the positive), for this rank's interactions of the unit;
 *                                             d_scores: THIS minibatch's matrix [global_batch][1 + n_neg], zeroed by the caller
 *        [all-reduce(sum) of d_scores over the ranks: every entry has exactly one non-zero contributor]
 *   per minibatch: slk_shard_adaptive_select  every rank, on the whole matrix: d_gk[pos * (1 + n_neg) + s] = dL/dscore (the
 *                                             positive of a live column -1/B, the column's first maximum +1/B, else 0);
 *                                             d_loss_out[0] = the minibatch's loss if report_loss, else 0 (the ranks' shares
 *                                             add up to the loss, as for the other losses)
 *   per unit:      slk_shard_user_pass_adaptive  user gradient and update from d_gk; EVERY lookup's gradient slot is written
 *                                             (zeros for the pairs the selection left out: the owner's "touched" semantics
 *                                             are then the one-GPU path's)
 *        [gradient all-to-all, slk_shard_item_pass, as before] */
int slk_shard_chunk_begin_adaptive(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                   const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                                   const int64_t *h_mb_off, int32_t n_minibatches, int32_t n_slices, int32_t n_neg,
                                   const int64_t *d_neg_in, int64_t *d_neg_out, const int64_t *d_mb_pos,
                                   int32_t *d_send_ids, int64_t *d_send_counts, void *stream);
int slk_shard_score_pass(slk_ctx *ctx, const slk_tables *local, int32_t unit, const float *d_rows_in, float *d_scores,
                         void *stream);
int slk_shard_adaptive_select(slk_ctx *ctx, int64_t global_batch, int32_t n_neg, const float *d_scores, float *d_gk,
                              float *d_loss_out, int32_t report_loss, void *stream);
int slk_shard_user_pass_adaptive(slk_ctx *ctx, const slk_tables *local, const slk_optim *optim, int32_t unit,
                                 const float *d_gk, const float *d_rows_in, float *d_grad_out, void *stream);

/* Measurement support (the reference has none; examples/bloom_embeddings/performance.py
 * times fit() with time.time()): when enabled, every launch of the engine's kernels is
 * bracketed by hipEvents on the launch stream.  slk_profile_read synchronises and returns,
 * per kernel class, the number of launches and the summed duration since the last reset. */
enum slk_kernel_class {
    SLK_K_SAMPLE = 0,   /* MT19937 generate + rejection compaction */
    SLK_K_PREP = 1,     /* key build + radix sorts + pack          */
    SLK_K_USER_PASS = 2,
    SLK_K_ITEM_PASS = 3,
    SLK_K_DENSE_SWEEP = 4,
    SLK_K_SCORE = 5,    /* adaptive-hinge score/select; predict    */
    SLK_K_EXCHANGE = 6, /* row-sharded path: owner-side row gather  */
    SLK_K_SEQ_PASS = 7, /* PoolNet sequence pass                    */
    SLK_K_EPOCH = 8,    /* persistent epoch kernel: all minibatches of a chunk in one launch */
    SLK_K_COUNT = 9
};
int slk_profile_enable(slk_ctx *ctx, int32_t on);
int slk_profile_read(slk_ctx *ctx, int32_t kernel_class, int64_t *launches, double *total_ms);
int slk_profile_reset(slk_ctx *ctx);

/* Bandwidth probes of the device (measurement support; SURVEY.md 8(d): "report the measured triad number
 * next to the nominal peak").  Both synchronise the stream and return average launch durations from
 * hipEvents recorded on it.
 *  slk_probe_stream        kind 0: a = b (float4 copy, 8 B moved per float); kind 1: a = b + s*c (triad,
 *                          12 B per float) over caller-owned device buffers of n_floats (multiple of 4);
 *                          kinds 2 / 3: the same with four independent non-temporal 16-B accesses per lane per
 *                          iteration; kind 4: read only (4 B per float); kind 5: write only (4 B per float);
 *                          kinds 6..11: chunked copies -- a workgroup moves contiguous chunks of 4 / 8 / 16 KB-per-wave
 *                          accesses, every load of a chunk in flight before its first store, plain or non-temporal.
 *  slk_probe_step_ceiling  the BilinearNet training step's ALGORITHMIC row accesses and nothing else (no
 *                          sorts, no user->item records, no id / key streams, no biases), on the caller's live
 *                          tables (values are written back unchanged), in the passes' lane layout:
 *                          user side = per interaction U[u] + state1 read and written, V[pos], V[neg] read
 *                          (factorization/representations.py:80-91 + the optimizer's row update); item side =
 *                          per distinct item of 2*batch uniform draws V[i] + state1 read and written.  The sum
 *                          of the two durations is what an exact fused step could reach if grouping and the
 *                          hand-over of pre-step user rows were free. */
int slk_probe_stream(slk_ctx *ctx, int32_t kind, float *d_a, const float *d_b, const float *d_c, int64_t n_floats,
                     int32_t iters, double *avg_ms, void *stream);
int slk_probe_step_ceiling(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t batch,
                           int32_t iters, double *user_ms, double *item_ms, int64_t *items_touched, void *stream);
/*  slk_probe_random_rows   the random-row regime of a table far larger than the caches (the 1B-item configuration):
 *                          n_access row groups each read (rmw = 0) or read and write back (rmw = 1) one embedding row
 *                          and its optimizer-state row in d_buf (2 * rows * dim floats); layout 0 = two separate
 *                          tables [rows][dim] + [rows][dim] (torch's parameter / state tensors), layout 1 = interleaved
 *                          [rows][2 * dim] records; order 0 = ascending rows (what an ownership pass over sorted ids
 *                          sees), order 1 = uniformly random rows (what the user pass's item lookups see). */
int slk_probe_random_rows(slk_ctx *ctx, float *d_buf, int64_t rows, int32_t dim, int32_t layout, int32_t order,
                          int32_t rmw, int64_t n_access, int32_t iters, double *avg_ms, void *stream);

/*  slk_probe_sort          the engine's stable LSD radix sort (csrc/slk_sort.hip) on caller-owned device arrays, for tests
 *                          and measurement: n (key, payload) pairs sorted by the key bits [0, bits); kind 0 = uint32 keys
 *                          + uint32 payloads, 1 = uint32 + uint64, 2 = uint64 + uint32 (+ 8: the input arrays may be overwritten --
 *                          they serve as the sort's second buffer pair, as in the training prep; iters must be 0); seg_len > 0: every run of seg_len
 *                          pairs is sorted on its own (the training prep's one-segment-per-minibatch form; the key bits at and
 *                          above `bits` are carried, not sorted).  iters > 0: the call is repeated and *avg_ms receives the
 *                          average duration from hipEvents on the stream (it synchronises); the inputs are left intact.
 *                          Replaces the sparse-gradient coalesce's sort inside `loss.backward(); optimizer.step()`
 *                          (spotlight/factorization/implicit.py:242-243). */
int slk_probe_sort(slk_ctx *ctx, int32_t kind, const void *d_keys_in, void *d_keys_out, const void *d_vals_in,
                   void *d_vals_out, int64_t n, int64_t seg_len, int32_t bits, int32_t iters, double *avg_ms, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPOTLIGHT_HIP_H */
