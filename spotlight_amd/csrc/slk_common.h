// slk_common.h -- internal declarations shared by the gfx950 engine's translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/spotlight_hip.h"

#define SLK_EAGAIN_EPOCH 1  // internal: the persistent route declined a chunk, take the launch path
#define SLK_EXPORT extern "C" __attribute__((visibility("default")))

// Occupancy target of a kernel in waves per SIMD (caps its VGPR budget at 512 / n).  hipcc only;
// the test harness's host build of these sources ignores it.
#if defined(__HIPCC__)
#define SLK_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define SLK_WAVES_PER_EU_RANGE(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#else
#define SLK_WAVES_PER_EU(n)
#define SLK_WAVES_PER_EU_RANGE(lo, hi)
#endif

// Wait until every outstanding vector-memory operation of this wave (stores included) has been acknowledged: what a
// workgroup does before it signals another one that its write-through (sc1) stores may be read (slk_epoch.hip).  Inline
// asm because the compiler's own wait-count bookkeeping may drop a builtin wait it considers redundant
// (MI355X_MICROARCH.md, "Compiler hazard").  hipcc only; the test harness's host build has no memory pipeline.
#if defined(__HIPCC__)
#define SLK_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SLK_DRAIN_VMEM() ((void)0)
#endif

// Synchronisation of a workgroup that is ONE wavefront (kernels launched with 64 threads: the MT19937 generator,
// slk_rng.hip): the wave's LDS accesses execute in program order, so nothing has to be waited for -- the wavefront-scope
// fences only keep the compiler from moving a read of another lane's word above the write that produces it.  No
// s_barrier and, unlike __syncthreads()'s workgroup-scope release, no wait for the global stores still in flight.
// The test harness runs every HIP thread as a fiber: there the marker is a switch point like __syncthreads().
#if defined(__HIPCC__)
#define SLK_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#else
#define SLK_WAVE_SYNC() __syncthreads()
#endif

// Marks the next plain kernel launch as one whose workgroups wait for each other inside the kernel (grid barrier): all of
// them must be resident at once.  On the GPU that is a property of the launch geometry (at most one wavefront-sized
// workgroup per CU, slk_epoch.hip) and the marker is empty; the test harness, which otherwise executes one block at a
// time, runs the marked grid's blocks concurrently.
#if defined(__HIPCC__)
#define SLK_RESIDENT_GRID_LAUNCH() ((void)0)
#else
#define SLK_RESIDENT_GRID_LAUNCH() ::emu::next_launch_resident()
#endif

// ---------------------------------------------------------------------------------------
// ctx
// ---------------------------------------------------------------------------------------
struct slk_buf {
    void *p = nullptr;
    size_t cap = 0;
};

// On-device RNG block (numpy RandomState layout: key[624] + pos) plus sampler bookkeeping.
struct slk_rng_dev {
    uint32_t key[624];
    int32_t pos;
    int32_t insufficient;     // sticky: a sampling call ran out of generated words
    unsigned long long t_last;  // stream index of the word that produced the last output
    unsigned long long accepted;
    int32_t epoch_abort;      // sticky: the persistent epoch kernel abandoned a launch (grid barrier time-out)
    int32_t sort_abort;       // sticky: a radix-sort look-back gave up waiting for the tile before it (slk_sort.hip)
};

#define SLK_EXTRA_BUFS 56

// buffers filled by the value-independent prep of one chunk of minibatches (slk_bilinear.hip)
struct slk_prep_bufs {
    slk_buf neg32, ukey[2], uval[2], uit, ikey[2], ipay[2];
    slk_buf bik[2], bip[2], buk[2], bup[2];  // BloomEmbedding hashed-row occurrence lists
    slk_buf mflag, msorted;         // single-occurrence fast path: "the item occurs more than once in its minibatch" per occurrence
    slk_buf lflags;                 // per minibatch of the chunk: does a run of the item-sorted list wholly cover a tile?
    int *h_lflags = nullptr;        //   (k_item_long_flags; read back once per chunk into PINNED host memory -- a pageable
    size_t h_lflags_cap = 0;        //   destination makes the copy wait for the stream -- see slk_launch_item_pass)
    size_t h_lflags_n = 0;
    hipEvent_t ev_lflags = nullptr; //   recorded behind the read-back: the host waits for it, not for the stream
};

struct slk_prof_span {
    int cls;
    hipEvent_t a, b;
};

struct slk_ctx {
    int device = 0;
    int num_cus = 256;
    size_t lds_per_block = (size_t)160 * 1024;  // LDS one workgroup may be granted (hipDeviceProp_t::sharedMemPerBlock)
    size_t lds_per_cu = (size_t)160 * 1024;     // LDS of a CU (maxSharedMemoryPerMultiProcessor; gfx950: 160 KB): occupancy pads
    char err[512] = {0};
    hipStream_t last_stream = nullptr;
    slk_rng_dev *d_rng = nullptr;
    uint32_t *d_jump[2] = {nullptr, nullptr};  // device copies of the jump polynomial tables (SLK_MT_JUMP_LEVELS)
    bool mt_attr_set = false;    // k_mt_jump's dynamic-LDS limit raised on this device
    // item-bias shadow of a training scope (slk_bias_shadow_begin / _end, slk_bilinear.hip): {bias, Adagrad sum} interleaved
    slk_buf bias_shadow;
    float *shadow_src_p = nullptr, *shadow_src_s = nullptr;  // the arrays it stands for (tables->d_param[3], optim->d_state1[3])
    int64_t shadow_rows = 0;
    bool shadow_active = false;
    // user-row ping-pong of a training scope (slk_user_pingpong_begin / _end, slk_bilinear.hip): the second copy of the user
    // embedding table and one byte per user (which copy holds the current row)
    slk_buf pp_alt, pp_flags;
    float *pp_src_u = nullptr;   // the array it doubles (tables->d_param[0])
    int64_t pp_rows = 0;
    int pp_dim = 0;
    bool pp_active = false;
    slk_buf mt_tmp;              // generator scratch: the 33-block prefix + one start block per stream (slk_rng.hip)

    // scratch (grown on demand, freed in slk_ctx_destroy)
    slk_buf raw, cnt, neg32, ukey[2], uval[2], uit, ikey[2], ipay[2], gk, sk, snap, losspart,
        sort_tmp, dgrad[4], ipart, ipart_meta, upart_meta;  // (the *_meta buffers keep launch stamps ACROSS calls: never shared)
    size_t dgrad_elems[4] = {0, 0, 0, 0};
    // tuning (slk_ctx_set_option)
    int64_t opt_chunk_interactions = (int64_t)1 << 23;  // interactions per prep chunk
    int opt_overlap_prep = 0;      // 1: prep of chunk c+1 on a second stream while chunk c trains (2: only its negatives).  The host
                                   // models switch it on for their epochs (spotlight_amd/factorization/implicit.py::_engine_for): in the
                                   // steady state of a run of training calls it gains 1.5-4 % (profiles/r03_*); a lone call of a few
                                   // chunks gains nothing, every pass runs ~10 % longer beside the sorts, and kernel timings under a
                                   // tracer stop agreeing with the untraced ones -- so a bare ctx keeps everything on one stream
    int64_t opt_mt_long_min_blocks = 16385;  // (= SLK_MT_JUMP_WG * SLK_MT_JUMP_BLOCKS + 1, asserted below)  // draws of at least this many state blocks: 256 blocks per stream
    int opt_prefetch_wait = 0;     // measurement switch (A/B of round 5): 1 = slk_bilinear_prefetch waits for the caller's stream as it did up to round 4
    int64_t opt_overlap_min_batch = (int64_t)1 << 16;  // the prep overlaps the passes only for minibatches of at least this size
                                   // (measured: +3 % at 8192, where the passes are short latency-bound kernels; -1..-3 % at 65 536)
    int64_t opt_sort_big_min = (int64_t)1 << 20;  // radix sort: sorts of at least this many pairs use tiles of 512 threads x 16 keys, smaller ones 256 x 16
    int opt_sort_debug = 0;        // measurement only: 1 the sort skips its look-back walks, 2 ranks from LDS atomics (results are wrong)
    int opt_item_grid_mult = 128;  // item pass: at most this many workgroups per CU (128 = one tile per workgroup at the C2 size, the
                                   // hardware balances the tail: item pass -2 % at C2 and the C5 shard, neutral at C3 / C4 / 65 536:
                                   // profiles/r04_g_*, r04_h_*)
    int opt_user_grid_mult = 8;    // user pass / other row passes
    int opt_seq_variant = 1;       // PoolNet: 1 = register-resident sequence pass when it fits, 0 = LDS-staged
    // adaptive hinge, plain item table: smallest minibatch whose item side is re-sorted per minibatch after the selection
    // (measured: one sort of all 1+n occurrences per chunk is faster up to 2^17 interactions per minibatch, slower from 2^18
    // -- profiles/r02_x_adaptive_small_batches.jsonl); a bloom item table (H rows per occurrence) always re-sorts
    int64_t opt_adaptive_late_min_batch = (int64_t)1 << 18;
    int64_t opt_user_lat_max_batch = (int64_t)1 << 17;  // minibatches up to this size take the latency-bound form of the pair-mode
                                   // user pass (k_user_pass<..., LAT>): two round trips per position instead of four.  Round 5
                                   // re-measured the crossover (profiles/r05_zb_*): user pass 21.2 -> 19.1 us at 32 768, 32.2 ->
                                   // 30.3 at 65 536, 53.7 -> 52.3 at 131 072, but 95 -> 97 at 262 144 (round 3 had it at 2^14)
    int64_t opt_item_lat_max_tiles = 2048;  // item pass: launches of up to this many 64-occurrence tiles take the form with every head's
                                   // row loaded early (k_item_pass<..., NPRE 4>); 0: never.  Same-box A/B (profiles/r03_y_*): item pass
                                   // 16.1 -> 13.5 us at minibatch 4096, 20.5 -> 16.5 at 16 384, 42.2 -> 38.0 at 65 536 (2048 tiles);
                                   // PoolNet 256 x 10 15.1 -> 13.4, but 256 x 200 (1600 tiles) 31.9 -> 34.2: a quarter of the limit there
    int opt_item_long_gate = 1;    // 1: minibatches without a long run (k_item_long_flags) take the plain item pass; 0: every item
                                   // pass is the partial-writing one + k_item_stitch (same results; a test / measurement switch)
    int opt_explicit_fused = 1;    // explicit feedback: 1 = score + loss inside the user pass, 0 = score pass + loss kernel first
    // minibatches <= opt_epoch_max_batch run inside ONE persistent launch per chunk (slk_epoch.hip).  Defaults from the
    // same-box A/Bs in profiles/r02_c_small_batch.jsonl: the persistent route wins at 256 (13 vs 21 us per minibatch) and
    // 1024 (17 vs 24), loses at 4096 (57 vs 30)
    int opt_epoch_kernel = 1;
    int64_t opt_epoch_max_batch = 1024;
    bool epoch_refused = false;    // a cooperative launch was refused on this device: stay on the launch path
    int opt_epoch_adaptive = 1;    // adaptive hinge on the persistent route (score phase + the selection inside the user phase)
    int64_t opt_epoch_adaptive_max_batch = 1024;  // ... for minibatches up to this size (three barriers and 1 + n occurrences per
                                   // interaction: same-box A/B in profiles/r03_u_*, r03_v_*)
    int epoch_bars_per_mb = 2;     // grid barriers per minibatch of the last persistent launch (the time-out report names the minibatch)
    int opt_epoch_barrier = -1;    // grid barrier of the persistent launch: 0 one arrival counter, 1 two levels (8 sub-counters), -1 by grid size
    int opt_epoch_cooperative = 0; // 1: hipLaunchCooperativeKernel (launch-time residency check; it also keeps kernels of OTHER streams
                                   // from running beside it, measured: profiles/r02_h_c1_fit_timeline_adagrad.json), 0: plain launch
    int opt_epoch_debug = 0;       // measurement only: 1 skip the phases' work, 2 do not wait at barriers, 4 no store drain
    int opt_epoch_max_grid = 256;  // workgroups (one wavefront each) of the persistent launch, <= one per CU
    int64_t opt_epoch_dense_elems = (int64_t)1 << 18;  // dense optimizers: largest model (parameters) the persistent route takes:
                                   // every row group sweeps rows / row-groups rows per phase, so only models of the reference's own
                                   // scale (MovieLens-100K at dim 32: 87 K parameters)
    void *h_coef[2] = {nullptr, nullptr};  // pinned host staging of the per-minibatch optimizer coefficients (slk_step_coef),
    size_t h_coef_cap[2] = {0, 0};         //   double-buffered: ev_coef[b] is recorded behind the copy issued from buffer b
    hipEvent_t ev_coef[2] = {nullptr, nullptr};
    int coef_flip = 0;
    int opt_nt = 3;                // cache policy: bit 0 user rows + state, bit 1 item rows + state non-temporal
                                   // (streamed once per pass); bit 3 key/payload streams (no gain measured); bit 4 the user
                                   // pass's record stores, bit 5 the item pass's record loads (round 6: the Infinity-Cache A/B,
                                   // profiles/r06_mall_ab.*)
    int64_t opt_item_single_min_items = (int64_t)1 << 24;  // item rows from which a ping-pong scope's Adagrad calls take the single-occurrence fast path (0: never)
    int opt_user_grid_own_occ = 0;    // 1: the user pass's grid is capped at the occupancy of the FORM it launches; 0 (default): at the smallest of
                                      // the four forms' -- measured, profiles/r06_w_*: 7 instead of 6 workgroups per CU buys the C2 pass nothing
                                      // (0.285-0.287 against 0.283 ms) and costs the C5 shard's 12 % (0.393 against 0.350: more rows in flight
                                      // over a 64-GB working set = more translation misses)
    int opt_user_bias_zero_hint = 1;  // 1: honour SLK_TABLES_USER_BIAS_ZERO (0: fetch the user biases regardless -- A/B and test switch)
    int64_t opt_record_nt_min_bytes = (int64_t)192 << 20;  // records of a minibatch from this size on are stored non-temporally
                                   // (slk_bilinear.hip::do_passes; 0: never)
    slk_prep_bufs pb[2];             // double-buffered: prep(c+1) overlaps passes(c)
    hipStream_t prep_stream = nullptr;
    bool prep_warmed = false;           // the one-off tiny prep on the prep stream has run (slk_bilinear_reserve)
    hipEvent_t ev_sampled = nullptr;    // behind the last draw of negatives (slk_sample_u32): slk_rng_get_state_sampled
    bool sampled_valid = false;         //   false after slk_rng_set_state / a shuffle (whose end only the stream sync knows)
    hipStream_t copy_stream = nullptr;  // small state copies (slk_rng_{set,get}_state): never the null stream
    hipEvent_t ev_start = nullptr, ev_prep[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    slk_buf extra[SLK_EXTRA_BUFS];  // path-specific scratch (slk_shard.hip, slk_seq.hip)
    // row-sharded path (slk_shard.hip): geometry of the chunk staged by slk_shard_chunk_begin/commit
    int64_t shard_n = -1;           // interactions of the committed chunk (-1: none)
    int64_t sh_n = 0;
    int sh_M = 0, sh_S = 0, sh_world = 0;
    int sh_NP = 2;                  // lookups per interaction of the staged chunk: 2 (pointwise / bpr / hinge), 1 + n_neg (adaptive hinge)
    bool sh_adaptive = false;       // the staged chunk was begun by slk_shard_chunk_begin_adaptive (its item lists live in SH_UIT / SH_GPOS)
    unsigned sh_ubits = 0;
    std::vector<int64_t> sh_ustart, sh_rstart;  // per unit: window in the user-sorted / received arrays
    std::vector<int64_t> sh_sslots, sh_rslots;  // per unit: slots of its requester-side buffers / first slot of its owner-side region
    std::vector<uint64_t> sh_host, sh_host2;     // host staging for small H2D tables (begin / commit)

    // Interactions.to_sequence plan (slk_seqprep.hip): rows pending a slk_to_sequence_fill (-1: none)
    int64_t ts_rows = -1, ts_nseg = 0;
    int ts_L = 0, ts_step = 0;

    // embedding front-end (slk_embed.hip): occurrences staged by slk_embedding_backward_plan (-1: none)
    int64_t em_occ = -1, em_rows = 0, em_segments = -1;
    int em_dim = 0;

    // slk_bilinear_prefetch: the first chunk of the NEXT training call, prepared (negatives + sorts) on the prep stream beside
    // the passes of the call before it
    struct {
        bool valid = false;
        int set = 0;
        const void *users = nullptr, *items = nullptr;
        int64_t n = 0, bsz = 0, nc0 = 0;
        int loss = 0, nn = 0;
        bool all = false;           // the negatives of the WHOLE call were drawn ahead (into pf_neg), not only the first chunk's
    } pf;
    slk_buf pf_neg;                 // uint32[n * negatives per interaction] of such a call
    slk_buf call_neg;               // the negatives of a whole (multi-chunk, not prefetched) training call: ONE draw (slk_bilinear.hip)
    int64_t stat_overlapped = 0;    // chunks whose negatives + sorts ran on the prep stream beside the chunk before's passes
    int64_t stat_prefetched = 0;    // chunks prepared ahead that a training call took over (slk_ctx_get_stat)
    int64_t stat_shadowed = 0;      // training calls that ran on the item-bias shadow (slk_bias_shadow_begin)
    int64_t stat_pingpong = 0;      // training calls that ran on the user-row ping-pong (slk_user_pingpong_begin)
    int64_t stat_single = 0;        // minibatches whose once-only items were updated by the user pass (k_user_pass<..., SGL>)
    int last_pipe_set = -1;         // buffer set of the last chunk of the last pipelined training call (-1: none yet)
    uint32_t ipart_gen = 0;         // item pass: stamp of the last launch's partials (slk_launch_item_pass)
    uint32_t upart_gen = 0;         // user pass, long runs: likewise (slk_bilinear.hip)
    int64_t stat_user_long = 0, stat_item_long = 0;  // launches of the partial-writing forms (slk_ctx_get_stat)
    int fy_sweeps = 0;              // slk_shuffle_perm: fixpoint sweeps of the last call (diagnostic)
    int fy_fallbacks = 0;           //   ranges of the last call that left the band and were redone with the full sweeps
    int opt_shuffle_band = 1;       // slk_shuffle_perm: 1 banded draws (default), 0 full sweeps, > 1 band / value (test hook: forces fall-backs)

    // profiling
    bool prof_on = false;
    std::vector<slk_prof_span> spans;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<const void *, int>> occ_cache;  // kernel -> resident workgroups per CU (slk_occupancy_of)
    int64_t prof_launches[SLK_K_COUNT] = {0};
    double prof_ms[SLK_K_COUNT] = {0};
};

int slk_fail(slk_ctx *ctx, int code, const char *fmt, ...);
int slk_ensure(slk_ctx *ctx, slk_buf &b, size_t bytes);
int slk_ensure_lflags_host(slk_ctx *ctx, slk_prep_bufs &pb, size_t n);  // pinned int[n], every entry 1 (= may hold a long run)
void slk_prof_begin(slk_ctx *ctx, int cls, hipStream_t s);
void slk_prof_end(slk_ctx *ctx, hipStream_t s);
int slk_prof_drain(slk_ctx *ctx);
int slk_prep_stream_init(slk_ctx *ctx);
// the ctx's own non-blocking stream for small host <-> device copies that must not touch the null stream
hipStream_t slk_copy_stream(slk_ctx *ctx);

#define SLK_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return slk_fail((ctx), SLK_EIO, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                             \
    } while (0)

#define SLK_LAUNCH_CHECK(ctx, what)                                                          \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess)                                                                \
            return slk_fail((ctx), SLK_EIO, "launch of %s failed: %s", (what), hipGetErrorString(e_)); \
    } while (0)

// MT19937 jump-ahead geometry: stream w of a draw (one wavefront of k_mt_stream, started by workgroup w - 1 of k_mt_jump)
// produces state blocks [w*SLK_MT_JUMP_BLOCKS, (w+1)*SLK_MT_JUMP_BLOCKS); one group of launches covers up to
// SLK_MT_JUMP_WG * SLK_MT_JUMP_BLOCKS blocks of 624 words (10.2 M words).
// Two stream lengths (round 5): draws of up to SLK_MT_JUMP_WG * 64 blocks use 64 blocks per stream; larger ones (a whole
// epoch's negatives, the epoch shuffle) 256 blocks per stream -- a jump costs a CU ~80 us whatever its distance, a block
// ~0.5 us, so one group of 256 long streams beats four groups of short ones.  Option "mt_long_min_blocks" moves the
// switch (test hook).
#define SLK_MT_JUMP_BLOCKS 64
#define SLK_MT_JUMP_LEVELS 2
static inline int slk_mt_jump_blocks(int level) { return level ? 4 * SLK_MT_JUMP_BLOCKS : SLK_MT_JUMP_BLOCKS; }
#define SLK_MT_JUMP_WG 256
static_assert(SLK_MT_JUMP_WG * SLK_MT_JUMP_BLOCKS + 1 == 16385, "slk_ctx::opt_mt_long_min_blocks");
#define SLK_MT_JUMP_TERMS 10752                 // exponent-list capacity per polynomial (multiple of 8)
#define SLK_MT_JUMP_PAD (33 * 624 - 1)          // exponent whose window X[1 + e + j] is the zero block
const uint32_t *slk_mt_jump_table(slk_ctx *ctx, int level);  // slk_mtjump.hip (host)

// sampler (slk_rng.hip): `count` negatives into ctx->neg32 (uint32) [+ int64 copy to d_out64]
int slk_sample_u32(slk_ctx *ctx, int64_t num_items, int64_t count, uint32_t *d_out32, int64_t *d_out64,
                   hipStream_t s);

// the engine's stable LSD radix sort (slk_sort.hip).  Pairs sorted by the key bits [0, end_bit); `clobber`: the input arrays
// may be overwritten (they serve as the second buffer pair; otherwise one is taken from the scratch)
int slk_sort_pairs_u32_u32(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                           uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber = false);
int slk_sort_pairs_u32_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber = false);
int slk_sort_pairs_u64_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber = false);
int slk_sort_pairs_u32_u64(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint64_t *vin,
                           uint64_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber = false);
int slk_sort_fy_steps(slk_ctx *ctx, slk_buf &scratch, const uint32_t *J, uint32_t n, uint32_t *const key[2], uint32_t *const val[2],
                      hipStream_t s);
int slk_sort_pairs_any(slk_ctx *ctx, int kind, const void *kin, void *kout, const void *vin, void *vout, size_t n, size_t seg_len,
                       unsigned bits, hipStream_t s, bool clobber);
// the training prep's sorts: keys built by the first pass from the id arrays, one segment per minibatch of `bsz` interactions;
// result in (key[1], val[1]), (key[0], val[0]) is the second buffer pair
int slk_sort_user_fat(slk_ctx *ctx, const int64_t *users, const int64_t *items, const uint32_t *neg32, size_t nc, size_t bsz,
                      unsigned ubits, unsigned mbbits, uint32_t *const key[2], uint64_t *const val[2], hipStream_t s);
int slk_sort_user_idx(slk_ctx *ctx, const int64_t *users, size_t nc, size_t bsz, unsigned ubits, unsigned mbbits,
                      uint32_t *const key[2], uint32_t *const val[2], hipStream_t s);
int slk_sort_item_occ(slk_ctx *ctx, const uint32_t *uit, size_t nocc, size_t bsz, int NP, unsigned ibits, unsigned mbbits,
                      uint32_t *const key[2], uint32_t *const val[2], hipStream_t s);

// slk_eval.hip: one representation against every item through the GEMM sweep (predict with d_items == NULL)
int slk_eval_predict_all(slk_ctx *ctx, const slk_tables *tables, const float *rep, const float *rbias, const int64_t *gmap,
                         float *d_out, hipStream_t s);
int slk_eval_user_rep(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const int64_t *d_user, const float **rep,
                      const float **rbias, const int64_t **gmap, hipStream_t s);

// shared host helpers (slk_bilinear.hip)
// shadow_ok: the caller indexes the item-bias shadow of an open slk_bias_shadow_begin scope itself (the training call); every
// other call that names the shadowed bias array is refused -- the caller's array is stale inside the scope
// pingpong_ok: likewise for the user rows of an open slk_user_pingpong_begin scope (slk_bilinear_train alone reads both copies)
int slk_check_tables(slk_ctx *ctx, const slk_tables *t, unsigned table_mask, int *vec, int *g, bool shadow_ok = false,
                     bool pingpong_ok = false);
int slk_launch_i64_to_u32(slk_ctx *ctx, const int64_t *in, uint32_t *out, size_t n, hipStream_t s);

int slk_sort_reserve(slk_ctx *ctx, size_t n);
// positions where a sorted key array changes value -> d_heads[0..nseg), d_heads[nseg] = n; optionally the
// segment index of every position -> d_segid[n] (slk_seqprep.hip).  Synchronises the stream (nseg is
// returned to the host).
int slk_compact_heads(slk_ctx *ctx, const uint32_t *d_sorted, uint32_t n, uint32_t *d_heads, uint32_t *d_segid,
                      uint32_t *nseg_out, hipStream_t s);
// slk_epoch.hip: the persistent route of slk_bilinear_train
bool slk_epoch_eligible(const slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t bsz, int loss, bool bloom);
int slk_epoch_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, uint32_t n_mb, int64_t bsz, int loss, int NP);
int slk_epoch_run_chunk(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, const slk_prep_bufs &pb, uint32_t nc,
                        int64_t bsz, unsigned ubits, unsigned ibits, int loss, int NP, int RS, float *snap, float *gsn,
                        float *d_mb_loss, const float *d_ratings, hipStream_t s);
// slk_rng.hip: regenerate `nblocks` MT19937 state blocks from the ctx's key into ctx->raw
int slk_mt_generate_blocks(slk_ctx *ctx, unsigned long long nblocks, hipStream_t s);
int slk_sample_reserve(slk_ctx *ctx, int64_t num_items, int64_t count);
int slk_mt_jump_reserve(slk_ctx *ctx, int level);  // jump table on the device + the generator's scratch (once per ctx)
int slk_mt_level_for(const slk_ctx *ctx, unsigned long long nblocks);  // stream length class of a draw of nblocks state blocks
// slk_rng.hip: numpy state -> d_rng without waiting for the ctx's stream (the caller knows no draw is in flight)
int slk_rng_write_state(slk_ctx *ctx, const uint32_t *h_key, int32_t pos);

static inline unsigned slk_bits_for(uint64_t max_value) {
    unsigned b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------

// Sum over the G lanes that share one embedding row (G = 1..64, power of two); every lane
// receives the total.  A row group never straddles a wavefront.
template <int G>
__device__ __forceinline__ float slk_group_sum(float x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, G);
    return x;
}

// Bitwise OR over the G lanes of a row group (every lane receives the result): log2(G) exchanges.
template <int G>
__device__ __forceinline__ unsigned long long slk_group_or(unsigned long long x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x |= __shfl_xor(x, m, G);
    return x;
}

// VEC consecutive fp32 of an embedding row held by one lane: VEC == 4 moves 16 B per lane
// (a D=64 row = 16 lanes x 16 B = one 256-B line), VEC == 1 is the odd-dim fallback.
template <int VEC>
struct slk_vec {
    float v[VEC];
};

template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_vzero() {
    slk_vec<VEC> r;
#pragma unroll
    for (int i = 0; i < VEC; ++i) r.v[i] = 0.0f;
    return r;
}

template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_vload(const float *p);
template <>
__device__ __forceinline__ slk_vec<4> slk_vload<4>(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    slk_vec<4> r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
template <>
__device__ __forceinline__ slk_vec<1> slk_vload<1>(const float *p) {
    slk_vec<1> r;
    r.v[0] = *p;
    return r;
}

template <int VEC>
__device__ __forceinline__ void slk_vstore(float *p, const slk_vec<VEC> &x);
template <>
__device__ __forceinline__ void slk_vstore<4>(float *p, const slk_vec<4> &x) {
    *reinterpret_cast<float4 *>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
template <>
__device__ __forceinline__ void slk_vstore<1>(float *p, const slk_vec<1> &x) {
    *p = x.v[0];
}

// The cache-policy bits a kernel acts on: the ctx option ("nt", a kernel argument), or a build-time constant
// (-DSLK_NT_FIXED=3) that removes the policy branches around every row access.
#ifdef SLK_NT_FIXED
#define SLK_NT_OF(a) (SLK_NT_FIXED)
#else
#define SLK_NT_OF(a) ((a).nt)
#endif

// Non-temporal (streaming) variants for rows that a pass touches exactly once: they should not
// displace the re-read data (item rows, records) from L2 / Infinity Cache.  hipcc only; the test
// harness's host build uses the plain accesses.
template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_vload_nt(const float *p) {
#if defined(__HIPCC__)
    slk_vec<VEC> r;
    if (VEC == 4) {
        typedef float slk_f4 __attribute__((ext_vector_type(4)));
        const slk_f4 t = __builtin_nontemporal_load(reinterpret_cast<const slk_f4 *>(p));
        r.v[0] = t.x; r.v[1 % VEC] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) r.v[i] = __builtin_nontemporal_load(p + i);
    }
    return r;
#else
    return slk_vload<VEC>(p);
#endif
}
template <int VEC>
__device__ __forceinline__ void slk_vstore_nt(float *p, const slk_vec<VEC> &x) {
#if defined(__HIPCC__)
    if (VEC == 4) {
        typedef float slk_f4 __attribute__((ext_vector_type(4)));
        slk_f4 t;
        t.x = x.v[0]; t.y = x.v[1 % VEC]; t.z = x.v[2 % VEC]; t.w = x.v[3 % VEC];
        __builtin_nontemporal_store(t, reinterpret_cast<slk_f4 *>(p));
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) __builtin_nontemporal_store(x.v[i], p + i);
    }
#else
    slk_vstore<VEC>(p, x);
#endif
}
__device__ __forceinline__ uint32_t slk_ld_u32(const uint32_t *p, bool nt) {
#if defined(__HIPCC__)
    return nt ? __builtin_nontemporal_load(p) : *p;
#else
    return *p;
#endif
}
template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_vload_if_nt(const float *p, bool nt) {
    return nt ? slk_vload_nt<VEC>(p) : slk_vload<VEC>(p);
}
template <int VEC>
__device__ __forceinline__ void slk_vstore_if_nt(float *p, const slk_vec<VEC> &x, bool nt) {
    if (nt) slk_vstore_nt<VEC>(p, x);
    else slk_vstore<VEC>(p, x);
}

// Device-coherent accesses for data that OTHER workgroups of the SAME launch write or read (the persistent epoch
// kernel, slk_epoch.hip): agent-scope relaxed atomics = `global_load/store ... sc1` on gfx950.  An sc1 store is
// written through to the fabric; an sc1 load bypasses the CU's L1 (which another CU's stores never refresh).  With
// both sides sc1 the hand-over needs no release / acquire fence (MI355X_MICROARCH.md, "Valid forms"): only the
// producer's `s_waitcnt vmcnt(0)` before it signals.  8 bytes per access (the widest atomic): a 16-B lane slice
// is two of them -- these kernels are latency-bound, not bandwidth-bound.
__device__ __forceinline__ float slk_ld_coh(const float *p) {
    const uint32_t b = __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float f;
    memcpy(&f, &b, 4);
    return f;
}
__device__ __forceinline__ void slk_st_coh(float *p, float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_vload_coh(const float *p) {
    slk_vec<VEC> r;
    if (VEC == 4) {
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t w[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
#pragma unroll
        for (int i = 0; i < VEC; ++i) memcpy(&r.v[i], &w[i % 4], 4);
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) r.v[i] = slk_ld_coh(p + i);
    }
    return r;
}
template <int VEC>
__device__ __forceinline__ void slk_vstore_coh(float *p, const slk_vec<VEC> &x) {
    if (VEC == 4) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) memcpy(&w[i], &x.v[i % VEC], 4);
        unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
        __hip_atomic_store(q, (unsigned long long)w[0] | ((unsigned long long)w[1] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, (unsigned long long)w[2] | ((unsigned long long)w[3] << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) slk_st_coh(p + i, x.v[i]);
    }
}

template <int VEC>
__device__ __forceinline__ float slk_vdot(const slk_vec<VEC> &a, const slk_vec<VEC> &b) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += a.v[i] * b.v[i];
    return s;
}

// acc += g * x
template <int VEC>
__device__ __forceinline__ void slk_vaxpy(slk_vec<VEC> &acc, float g, const slk_vec<VEC> &x) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] += g * x.v[i];
}

// Block-wide sum of one double per thread (256 threads); result valid in thread 0.
__device__ __forceinline__ double slk_block_sum_256(double x, double *red /*[256] LDS*/) {
    const int t = threadIdx.x;
    red[t] = x;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    return red[0];
}
