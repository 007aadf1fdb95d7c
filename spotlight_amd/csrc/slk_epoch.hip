// slk_epoch.hip -- the PERSISTENT EPOCH KERNEL: every minibatch of a prepared chunk inside ONE launch.
//
// Replaces the Python minibatch loop of ImplicitFactorizationModel.fit() (spotlight/factorization/implicit.py:223-243)
// at the reference's own operating points -- batch_size 256 (its default, implicit.py:80) to a few thousand (1024 in
// its tests, tests/factorization/test_implicit.py:40-57) -- where a minibatch moves a few hundred KB and the
// per-minibatch launches of slk_bilinear.hip (user pass, item pass, dense sweep) cost more than the work.
//
// One cooperative launch (all workgroups resident) runs, for every minibatch of the chunk in order:
//
//   USER PHASE   one row group per unique user (head of its run in the (minibatch, user)-sorted list): U[u], V[pos],
//                V[neg], biases; dot products by __shfl_xor; loss + dL/dscore; records (pre-step user row, dL/dscore
//                per pair) for the item phase; the optimizer update of U[u] and its bias in place.  Explicit feedback
//                (ExplicitFactorizationModel.fit, spotlight/factorization/explicit.py:213-236, batch_size 256 by default):
//                ONE pair per interaction, the loss formed against the observed rating
//   -- grid barrier --
//   ITEM PHASE   one row group per unique item (head of its run in the (minibatch, item)-sorted occurrence list): sums
//                g * u_old over the run in occurrence order, updates V[i] and its bias
//   -- grid barrier --
//
// Adaptive hinge (implicit.py:266-275, losses.py:127-166; n negatives per interaction, default 5): a SCORE PHASE in front --
// one row group per interaction: its 1 + n scores, k_score_pass's arithmetic -- and a grid barrier; the user phase then
// evaluates, per interaction, the 1 + n columns of the reference's view(n, B) candidate matrix that decide its pairs (one
// column per lane: the positive's own column and the column each of its draws sits in; k_adaptive_select's selection, first
// maximum wins), so dL/dscore needs no phase and no barrier of its own: three barriers per minibatch instead of the launch
// path's four launches.
//
// i.e. the two ownership passes of the launch path (same sorted lists, same summation order, same arithmetic: the
// trained tables are BIT-IDENTICAL to the per-minibatch launches, which the tests assert), with the launch boundaries
// replaced by an in-kernel barrier.  The dense optimizers (the reference's default Adam + l2, Adagrad + weight decay:
// EVERY row is updated every step, torch/optim/adam.py:414-546) need no third phase and no gradient buffer: the rows a
// minibatch touches are stamped one phase ahead and every row group sweeps a share of the unstamped ones (see below).
//
// Inter-workgroup visibility (MI355X: 8 XCDs with private, mutually non-coherent L2s; per-CU L1s that other CUs' stores
// never refresh).  Everything one workgroup writes and another reads within the launch -- table rows, optimizer state,
// records, loss partials -- is accessed ONLY through slk_*_coh (agent-scope relaxed atomics = sc1 write-through stores /
// L1-bypassing loads): with both sides sc1 the hand-over needs no cache flush, only the producer's `s_waitcnt vmcnt(0)`
// before it arrives at the barrier.  The barrier is one monotonic device-scope counter (arrive = relaxed fetch_add,
// wait = relaxed sc1 polling with s_sleep by ONE lane per workgroup); it is placement-independent and every spin is
// bounded: on a time-out the high bit of the counter is raised, every workgroup leaves, and the host reports SLK_EIO.
// The sorted id lists are written by the prep kernels BEFORE the launch and never change: plain (cached) loads.
#include "slk_kernels.h"

// One wavefront per workgroup, at most one workgroup per CU: a CU retires its vector-memory instructions one line at a
// time (scripts/micro/coherent_latency.hip: ~5 cycles per loaded line, ~14 per stored line, the same for plain and sc1
// accesses), so the four waves of a 256-thread workgroup queue behind each other exactly where this kernel spends its time;
// spread over 4x as many CUs the same work measured 18.8 -> see profiles/ us per minibatch at C1 shape, minibatch 1024.
#define SLK_EPOCH_TB 64

enum { SLK_EUPD_ADAGRAD = 0, SLK_EUPD_SPARSE_ADAM = 1, SLK_EUPD_ADAM_DENSE = 2, SLK_EUPD_ADAGRAD_DENSE = 3, SLK_EUPD_SGD = 4 };
// a zero summed gradient leaves the row exactly as it is: row-sparse Adagrad and SGD
#define SLK_EUPD_ZERO_IS_NOOP(UPD) ((UPD) == SLK_EUPD_ADAGRAD || (UPD) == SLK_EUPD_SGD)

#define SLK_EPOCH_ABORT 0xffffffffu
#define SLK_EPOCH_MAX_SPINS (1u << 24)  // x (s_sleep + one fabric round trip): seconds

// per-minibatch optimizer coefficients, formed on the host in double exactly as torch does (bias corrections and
// lr decay depend on the step count)
struct slk_step_coef {
    float c0;  // Adagrad (sparse / dense): clr.  SparseAdam: lr * sqrt(bc2) / bc1.  Adam dense: lr / bc1
    float c1;  // Adam dense: sqrt(bc2)
};

struct slk_epoch_args {
    float *P[4], *S1[4], *S2[4];
    int D;
    uint32_t n_users, n_items;
    uint32_t nc, bsz, n_mb;        // interactions of the chunk, minibatch size, minibatches
    const uint32_t *ukey, *uit;    // (minibatch << ubits) | user, sorted; [2 * position] = (pos item, neg item)
    const uint32_t *uk;            // explicit feedback: uit[position] = item, uk[position] = the interaction's index in
    const float *ratings;          //   ratings[] (chunk-local)
    int NP;                        // adaptive hinge: uit[NP * position + s] = the positive (s = 0) and the n draws of the interaction
    float *sk;                     //   uk[position] (chunk-local); sk[NP * interaction + s] = their scores (score phase)
    uint32_t umask;
    const uint32_t *ikey, *ipay;   // (minibatch << ibits) | item, sorted; occurrence -> 2 * position + pair (explicit: position)
    uint32_t imask;
    float *snap;                   // records: [position - b0][RS] pre-step user rows
    int RS;
    float *gsn;                    // [2 * (position - b0) + pair] dL/dscore (explicit: [position - b0])
    double *partial;               // [n_mb][gridDim.x] per-workgroup loss sums
    float *mb_loss;                // [n_mb] loss.item() of each minibatch
    const slk_step_coef *coef;     // [n_mb]
    uint32_t *touch_u, *touch_i;   // dense optimizers: touch[row] = (last minibatch that looks the row up) + 1; zeroed before the launch
    unsigned *bar;                 // barrier counter, zeroed before the launch
    int *status;                   // barrier time-out: the number (>= 1) of the barrier that was abandoned, 2 per minibatch (adaptive: 3)
    int loss_kind;
    float eps, omb1, omb2, beta2, wd;
    int bar_kind;                  // 0: one arrival counter, 1: 8 sub-counters + a top counter
    int debug;                     // measurement only (option "epoch_debug"): 1 skip the phases' work, 2 do not wait at barriers, 4 no drain,
                                   // 8 / 16 / 32 skip the user (sequence) / item / score phase alone
};

// ---- the grid barrier -------------------------------------------------------------------------------------------------
// `epoch` = number of barriers passed so far + 1 (monotonic within the launch; all words are zeroed before it).
// Arrivals are counted by device-scope fetch_adds; the LAST arriver publishes the epoch in a separate word that all
// others poll (relaxed sc1 loads + s_sleep, ONE lane per workgroup), so the pollers do not queue behind the arrivals at
// the counter's memory channel.  bar_kind 1 adds a level: workgroups first meet on one of 8 sub-counters (blockIdx % 8 --
// a grouping, not a placement assumption) and only the last of each group touches the top counter.
// Returns false when the launch is being abandoned (a spin ran out: the flag word is raised to ~0 for everyone).
// `partial_out` (optional): a value thread 0 publishes on its way in (the workgroup's loss sum; read barriers later).
__device__ __forceinline__ bool slk_epoch_barrier(const slk_epoch_args &e, unsigned epoch, int *s_flag, const double *s_wave_sums,
                                                  double *partial_out) {
    if (!(e.debug & 4)) SLK_DRAIN_VMEM();  // every wave: its write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (partial_out) {
            const double tot = ((s_wave_sums[0] + s_wave_sums[1]) + s_wave_sums[2]) + s_wave_sums[3];
            unsigned long long bits;
            memcpy(&bits, &tot, 8);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(partial_out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned *flag = e.bar + 32;  // its own 128-B line
        bool last;
        if (e.bar_kind == 1) {
            const unsigned sub = blockIdx.x & 7u;
            const unsigned members = (gridDim.x - sub + 7u) / 8u;
            const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
            last = __hip_atomic_fetch_add(e.bar + 64 + 32 * sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members * epoch;
            if (last) last = __hip_atomic_fetch_add(e.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == groups * epoch;
        } else {
            last = __hip_atomic_fetch_add(e.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x * epoch;
        }
        unsigned seen = epoch;
        if (last) {
            __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!(e.debug & 2)) {
            unsigned spins = 0;
            seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (seen < epoch) {
                __builtin_amdgcn_s_sleep(1);
                seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > SLK_EPOCH_MAX_SPINS) {  // a workgroup never arrived: give up, tell everyone
                    __hip_atomic_store(flag, SLK_EPOCH_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(e.status, (int)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // which barrier: 2 per minibatch
                    seen = SLK_EPOCH_ABORT;
                }
            }
        }
        *s_flag = seen == SLK_EPOCH_ABORT ? 0 : 1;
    }
    __syncthreads();
    return *s_flag != 0;  // the caller alternates between two flag words, so no third barrier is needed
}

// ---- row updates: the arithmetic of slk_apply_vec / k_dense_sweep_all, element for element, on coherent accesses -------
template <int VEC, int UPD>
__device__ __forceinline__ void slk_epoch_update(const slk_epoch_args &e, const slk_step_coef &c, int t, size_t off,
                                                 slk_vec<VEC> p, slk_vec<VEC> s1, slk_vec<VEC> s2, const slk_vec<VEC> &g) {
    if (UPD == SLK_EUPD_ADAGRAD) {  // torch/optim/adagrad.py:360-385
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s1.v[i] += g.v[i] * g.v[i];
            p.v[i] += -c.c0 * (g.v[i] / (sqrtf(s1.v[i]) + e.eps));
        }
        slk_vstore_coh<VEC>(e.S1[t] + off, s1);
    } else if (UPD == SLK_EUPD_SPARSE_ADAM) {  // torch/optim/_functional.py:61-84
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float mu = (g.v[i] - s1.v[i]) * e.omb1;
            const float vu = (g.v[i] * g.v[i] - s2.v[i]) * e.omb2;
            s1.v[i] = mu + s1.v[i];
            s2.v[i] = vu + s2.v[i];
            p.v[i] += -c.c0 * (s1.v[i] / (sqrtf(s2.v[i]) + e.eps));
        }
        slk_vstore_coh<VEC>(e.S1[t] + off, s1);
        slk_vstore_coh<VEC>(e.S2[t] + off, s2);
    } else if (UPD == SLK_EUPD_SGD) {  // torch/optim/sgd.py, momentum 0, weight_decay 0: param.add_(grad, alpha=-lr)
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.v[i] += -c.c0 * g.v[i];
    } else if (UPD == SLK_EUPD_ADAM_DENSE) {  // torch/optim/adam.py:414-546 (k_dense_sweep_all)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float gv = g.v[i] + e.wd * p.v[i];
            const float m = s1.v[i] + e.omb1 * (gv - s1.v[i]);
            const float v = s2.v[i] * e.beta2 + e.omb2 * (gv * gv);
            s1.v[i] = m;
            s2.v[i] = v;
            p.v[i] += -c.c0 * (m / (sqrtf(v) / c.c1 + e.eps));
        }
        slk_vstore_coh<VEC>(e.S1[t] + off, s1);
        slk_vstore_coh<VEC>(e.S2[t] + off, s2);
    } else {  // torch/optim/adagrad.py:350-385, dense branch with weight decay
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float gv = g.v[i] + e.wd * p.v[i];
            const float s = s1.v[i] + gv * gv;
            s1.v[i] = s;
            p.v[i] += -c.c0 * (gv / (sqrtf(s) + e.eps));
        }
        slk_vstore_coh<VEC>(e.S1[t] + off, s1);
    }
    slk_vstore_coh<VEC>(e.P[t] + off, p);
}

template <int UPD>
__device__ __forceinline__ constexpr bool slk_epoch_has_s2() {
    return UPD == SLK_EUPD_SPARSE_ADAM || UPD == SLK_EUPD_ADAM_DENSE;
}
template <int UPD>
__device__ __forceinline__ constexpr bool slk_epoch_has_s1() {
    return UPD != SLK_EUPD_SGD;
}
template <int UPD>
__device__ __forceinline__ constexpr bool slk_epoch_dense() {
    return UPD == SLK_EUPD_ADAM_DENSE || UPD == SLK_EUPD_ADAGRAD_DENSE;
}

// Dense optimizers (the reference's default Adam + l2, Adagrad + weight decay): EVERY row is updated every step, the rows a
// minibatch does not touch with a zero gradient (torch/optim/adam.py:414-546; k_dense_sweep_all on the launch path).  Which rows
// a minibatch touches depends on ids only, so the phase BEFORE a table's turn stamps them (touch[row] = minibatch + 1: the
// users of minibatch m + 1 during item phase m, the items of minibatch m during user phase m) and during the table's own phase
// every row group sweeps a strided share of ALL rows, skipping the stamped ones -- their owners update them with the summed
// gradient.  Balanced (rows / row groups iterations per phase), no gradient buffer, no third phase; the arithmetic is
// k_dense_sweep_all's element for element.
template <int VEC, int UPD>
__device__ __forceinline__ void slk_epoch_sweep_untouched(const slk_epoch_args &e, const slk_step_coef &c, int te, int tb,
                                                          const uint32_t *touch, uint32_t want, uint32_t n_rows, uint32_t gslot,
                                                          uint32_t gstride, int D, int d0, bool on, int lane) {
    for (uint32_t r = gslot; r < n_rows; r += gstride) {
        // stamp, row, state and (lane 0) the bias triple in one round trip
        const uint32_t stamp = __hip_atomic_load(touch + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slk_vec<VEC> p = slk_vzero<VEC>(), s1 = p, s2 = p;
        slk_vec<1> bp = slk_vzero<1>(), bs1 = bp, bs2 = bp;
        if (on) {
            const size_t off = (size_t)r * D + d0;
            p = slk_vload_coh<VEC>(e.P[te] + off);
            s1 = slk_vload_coh<VEC>(e.S1[te] + off);
            if (slk_epoch_has_s2<UPD>()) s2 = slk_vload_coh<VEC>(e.S2[te] + off);
        }
        if (lane == 0) {
            bp = slk_vload_coh<1>(e.P[tb] + r);
            bs1 = slk_vload_coh<1>(e.S1[tb] + r);
            if (slk_epoch_has_s2<UPD>()) bs2 = slk_vload_coh<1>(e.S2[tb] + r);
        }
        if (stamp == want) continue;  // touched: its owner applies the summed gradient
        if (on) slk_epoch_update<VEC, UPD>(e, c, te, (size_t)r * D + d0, p, s1, s2, slk_vzero<VEC>());
        if (lane == 0) slk_epoch_update<1, UPD>(e, c, tb, r, bp, bs1, bs2, slk_vzero<1>());
    }
}

// ---- the item phase of one minibatch ------------------------------------------------------------------------------------
// IMODE: how a payload names its record -- pair losses (occurrence = 2 * position + pair), explicit feedback (= position),
// adaptive hinge (NP * position + pair), PoolNet (NP * timestep + pair; every occurrence is live, pair 0 also carries the
// timestep's history gradient, padding_idx rows are nobody's).  b0: first position (timestep) of the minibatch.
enum { SLK_EI_PAIR = 0, SLK_EI_EXPL = 1, SLK_EI_ADP = 2 };

#ifndef SLK_EPOCH_ITEM_BATCH
#define SLK_EPOCH_ITEM_BATCH 4  // positions of a row group (gstride apart) whose ids, then whose heads' rows + first records, are fetched
                                // together: a minibatch with more occurrences than the grid has row groups (adaptive hinge, PoolNet,
                                // pair losses at 1024) is otherwise one dependent chain of round trips per position
#endif

template <int VEC, int G, int UPD, int IMODE>
__device__ __forceinline__ void slk_epoch_item_phase(const slk_epoch_args &e, const slk_step_coef &c, uint32_t b0, uint32_t ib0,
                                                     uint32_t ib1, uint32_t NP, uint32_t gslot, uint32_t gstride, int D, int d0,
                                                     bool on, int lane, uint32_t nx_key, uint32_t nx_prev, uint32_t nx_a) {
    constexpr bool HAS_S2 = slk_epoch_has_s2<UPD>();
    constexpr bool HAS_S1 = slk_epoch_has_s1<UPD>();
    constexpr int NB = SLK_EPOCH_ITEM_BATCH;
    const slk_vec<VEC> zero = slk_vzero<VEC>();
    const slk_vec<1> zero1 = slk_vzero<1>();
    (void)NP;
    auto pos_of = [&](uint32_t pay) -> uint32_t {
        return IMODE == SLK_EI_EXPL ? pay : (IMODE == SLK_EI_PAIR ? pay >> 1 : pay / NP);
    };
    for (uint32_t r0 = ib0 + gslot; r0 < ib1 && !(e.debug & 1); r0 += (uint32_t)NB * gstride) {
        // (1) the ids of the batch's positions; which of them are heads of their item's run
        uint32_t key[NB], pay[NB];
        bool head[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t r = r0 + (uint32_t)i * gstride;
            key[i] = pay[i] = 0u;
            head[i] = false;
            if (r < ib1) {
                const bool first = r == ib0;
                const bool pre = i == 0 && r0 == ib0 + gslot;  // this position's list entries were prefetched
                key[i] = pre ? nx_key : e.ikey[r];
                const uint32_t prev = pre ? nx_prev : (first ? 0u : e.ikey[r - 1]);
                pay[i] = pre ? nx_a : e.ipay[r];
                // padding_idx rows receive no gradient (the launch path never makes them heads)
                head[i] = first || prev != key[i];
            }
        }
        // (2) every head's row, optimizer state, bias and first record -- and the ids of the occurrence behind it -- in ONE round trip
        slk_vec<VEC> v[NB], sv1[NB], sv2[NB], uo[NB];
        slk_vec<1> bis1[NB], bis2[NB];
        float bi[NB], g[NB];
        uint32_t nkey[NB], npay[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            v[i] = sv1[i] = sv2[i] = uo[i] = zero;
            bis1[i] = bis2[i] = zero1;
            bi[i] = g[i] = 0.0f;
            nkey[i] = ~key[i];
            npay[i] = 0u;
            if (head[i]) {
                const uint32_t r = r0 + (uint32_t)i * gstride;
                const uint32_t item = key[i] & e.imask;
                const size_t voff = (size_t)item * D + d0;
                if (on) v[i] = slk_vload_coh<VEC>(e.P[1] + voff);
                if (on && HAS_S1) sv1[i] = slk_vload_coh<VEC>(e.S1[1] + voff);
                if (on && HAS_S2) sv2[i] = slk_vload_coh<VEC>(e.S2[1] + voff);
                bi[i] = slk_ld_coh(e.P[3] + item);
                if (lane == 0) {
                    if (HAS_S1) bis1[i] = slk_vload_coh<1>(e.S1[3] + item);
                    if (HAS_S2) bis2[i] = slk_vload_coh<1>(e.S2[3] + item);
                }
                g[i] = slk_ld_coh(e.gsn + (pay[i] - ib0));
                const uint32_t pos = pos_of(pay[i]);
                if (on) uo[i] = slk_vload_coh<VEC>(e.snap + (size_t)(pos - b0) * e.RS + d0);
                if (r + 1u < ib1) {
                    nkey[i] = e.ikey[r + 1u];
                    npay[i] = e.ipay[r + 1u];
                }
            }
        }
        // (3) the runs, one after the other
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!head[i]) continue;
            const uint32_t r = r0 + (uint32_t)i * gstride;
            const uint32_t kkey = key[i], item = kkey & e.imask;
            const size_t voff = (size_t)item * D + d0;
            // The run is summed as the launch path sums it (slk_kernels.h, k_item_pass + k_item_stitch): in occurrence order --
            // unless it is LONG, i.e. wholly covers one of the launch path's (full) tiles of TT positions of the minibatch's
            // occurrence list: then tile by tile (in occurrence order inside a tile), the tiles' sums added in order.
            // Both sums are kept; which one applies is known when the run ends.
            constexpr uint32_t TT = 4u * (256u / (uint32_t)G);
            slk_vec<VEC> sq = zero, gv = zero, tv = zero;
            float sqb = 0.0f, gb = 0.0f, tb = 0.0f;
            bool any = false, first_tile = true, is_long = false;
            const uint32_t p0 = r - ib0;
            uint32_t k = r;
            float gc = g[i];
            slk_vec<VEC> uoc = uo[i];
            uint32_t nk = nkey[i], np_ = npay[i];  // key and payload of occurrence k + 1 (~key: there is none)
            for (;;) {
                // the next occurrence's record is requested before this one is added; its ids arrived with this one's record
                const bool cont = nk == kkey;
                float gn = 0.0f;
                slk_vec<VEC> uon = zero;
                uint32_t nk2 = ~kkey, np2 = 0u;
                if (cont) {
                    const uint32_t posn = pos_of(np_);
                    gn = slk_ld_coh(e.gsn + (np_ - ib0));
                    if (on) uon = slk_vload_coh<VEC>(e.snap + (size_t)(posn - b0) * e.RS + d0);
                    if (k + 2u < ib1) {
                        nk2 = e.ikey[k + 2u];
                        np2 = e.ipay[k + 2u];
                    }
                }
                if (gc != 0.0f) {  // occurrences without a gradient (inactive hinge) do not touch the sum
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float cc = gc * uoc.v[q];
                        sq.v[q] += cc;
                        tv.v[q] += cc;
                    }
                    sqb += gc;
                    tb += gc;
                    any = true;
                }
                ++k;
                const bool run_ends = !cont;
                const uint32_t rel = k - ib0;
                const bool boundary = rel % TT == 0u;
                if (run_ends || boundary) {  // the tile [tile_start, rel) is done: its sum joins the run's
                    const uint32_t tile_start = (rel - 1u) / TT * TT;
                    is_long = is_long || (tile_start >= p0 && boundary);  // a FULL tile covered from its first to its last position
                    if (first_tile) {
                        gv = tv;
                        gb = tb;
                        first_tile = false;
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) gv.v[q] += tv.v[q];
                        gb += tb;
                    }
                    tv = zero;
                    tb = 0.0f;
                }
                if (run_ends) break;
                gc = gn;
                uoc = uon;
                nk = nk2;
                np_ = np2;
            }
            if (!is_long) {
                gv = sq;
                gb = sqb;
            }

            // Adagrad: a run without any gradient is an exact no-op; SparseAdam decays the moments of every looked-up
            // row; the dense optimizers update every row anyway
            if (!(SLK_EUPD_ZERO_IS_NOOP(UPD) && !any)) {
                if (on) slk_epoch_update<VEC, UPD>(e, c, 1, voff, v[i], sv1[i], sv2[i], gv);
                if (lane == 0 && !(SLK_EUPD_ZERO_IS_NOOP(UPD) && gb == 0.0f)) {
                    slk_vec<1> bp, bg;
                    bp.v[0] = bi[i];
                    bg.v[0] = gb;
                    slk_epoch_update<1, UPD>(e, c, 3, item, bp, bis1[i], bis2[i], bg);
                }
            }
        }
    }
}

enum { SLK_EMODE_PAIR = 0, SLK_EMODE_EXPLICIT = 1, SLK_EMODE_ADAPTIVE = 2 };

template <int VEC, int G, int UPD, int MODE>
__global__ __launch_bounds__(SLK_EPOCH_TB) void k_bilinear_epoch(slk_epoch_args e) {
    constexpr bool EXPL = MODE == SLK_EMODE_EXPLICIT, ADP = MODE == SLK_EMODE_ADAPTIVE;
    const uint32_t NP = ADP ? (uint32_t)e.NP : (EXPL ? 1u : 2u);  // score pairs (= item occurrences) per interaction
    HIP_DYNAMIC_SHARED(double, s_wave_sums)      // [4] per-wave loss sums (unused slots stay 0) + the barrier's two flag words
    int *s_flags = reinterpret_cast<int *>(s_wave_sums + 4);
    if (threadIdx.x < 4) s_wave_sums[threadIdx.x] = 0.0;
    constexpr int TB = SLK_EPOCH_TB, NW = TB / 64;
    constexpr int GPB = TB / G;
    constexpr bool DENSE = slk_epoch_dense<UPD>();
    constexpr bool HAS_S2 = slk_epoch_has_s2<UPD>();
    constexpr bool HAS_S1 = slk_epoch_has_s1<UPD>();
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int D = e.D, d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t gslot = blockIdx.x * GPB + grp, gstride = gridDim.x * GPB;
    const slk_vec<VEC> zero = slk_vzero<VEC>();
    const slk_vec<1> zero1 = slk_vzero<1>();
    unsigned barriers = 0;

    // The sorted id lists are immutable: what a row group needs of them for its FIRST position of the next phase is
    // fetched before the barrier it is about to wait at, off the critical path.
    uint32_t nx_key = 0, nx_prev = 0, nx_a = 0, nx_b = 0;
    if (!ADP && gslot < (e.nc < e.bsz ? e.nc : e.bsz)) {
        nx_key = e.ukey[gslot];
        nx_prev = gslot ? e.ukey[gslot - 1] : 0u;
        nx_a = e.uit[NP * (size_t)gslot];
        nx_b = EXPL ? e.uk[gslot] : e.uit[2 * (size_t)gslot + 1];
    }
    // adaptive hinge, score phase: a UNIT is one position's scores SB at a time (1 + 5 draws: two units per position, so that
    // a minibatch of 256 spreads over 512 row groups); the ids of a row group's first unit travel ahead like nx_*
    constexpr int SB = 3;
    const uint32_t nch = ADP ? (NP + SB - 1u) / SB : 1u;
    uint32_t sc_key = 0, sc_k = 0, sc_it[SB] = {0u, 0u, 0u};
    auto score_prefetch = [&](uint32_t b0, uint32_t b1) {
        if (gslot < (b1 - b0) * nch) {
            const uint32_t p = b0 + gslot / nch, s0 = (gslot % nch) * SB;
            sc_key = e.ukey[p];
            sc_k = e.uk[p];
#pragma unroll
            for (int j = 0; j < SB; ++j) sc_it[j] = s0 + j < NP ? e.uit[(size_t)p * NP + s0 + j] : 0u;
        }
    };
    if (ADP) score_prefetch(0u, e.nc < e.bsz ? e.nc : e.bsz);

    if (DENSE) {  // prologue: the users of minibatch 0
        const uint32_t nb1 = e.nc < e.bsz ? e.nc : e.bsz;
        if (lane == 0)
            for (uint32_t p = gslot; p < nb1; p += gstride)
                __hip_atomic_store(e.touch_u + (e.ukey[p] & e.umask), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!slk_epoch_barrier(e, barriers + 1, s_flags + (barriers & 1u), s_wave_sums, nullptr)) return;
        ++barriers;
    }

    for (uint32_t mb = 0; mb < e.n_mb; ++mb) {
        const uint32_t b0 = mb * e.bsz, b1 = (e.nc - b0 < e.bsz) ? e.nc : b0 + e.bsz;
        const float inv_b = 1.0f / (float)(b1 - b0);
        const slk_step_coef c = e.coef[mb];

        // ------------------------------------------------ SCORE PHASE (adaptive hinge): k_score_pass's arithmetic
        if (ADP) {
            const uint32_t n_units = (b1 - b0) * nch;
            for (uint32_t unit = gslot; unit < n_units && !(e.debug & (1 | 32)); unit += gstride) {
                const bool pre = unit == gslot;
                const uint32_t p = b0 + unit / nch, s0 = (unit % nch) * SB;
                const uint32_t user = (pre ? sc_key : e.ukey[p]) & e.umask;
                const size_t kb = (size_t)(pre ? sc_k : e.uk[p]) * NP, qb = (size_t)p * NP;
                uint32_t it[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) it[j] = pre ? sc_it[j] : (s0 + j < NP ? e.uit[qb + s0 + j] : 0u);
                const slk_vec<VEC> u = on ? slk_vload_coh<VEC>(e.P[0] + (size_t)user * D + d0) : zero;
                const float bu = slk_ld_coh(e.P[2] + user);
                slk_vec<VEC> v[SB];
                float bi[SB];
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    v[j] = zero;
                    bi[j] = 0.0f;
                    if (s0 + j < NP) {
                        if (on) v[j] = slk_vload_coh<VEC>(e.P[1] + (size_t)it[j] * D + d0);
                        bi[j] = slk_ld_coh(e.P[3] + it[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const float sc = slk_group_sum<G>(slk_vdot<VEC>(u, v[j])) + bu + bi[j];
                    if (lane == 0 && s0 + j < NP) slk_st_coh(e.sk + kb + s0 + j, sc);
                }
            }
            // this row group's first position of the user phase: key, interaction, and its pairs' items one per lane
            if (b0 + gslot < b1) {
                nx_key = e.ukey[b0 + gslot];
                nx_prev = gslot ? e.ukey[b0 + gslot - 1] : 0u;
                nx_a = e.uk[b0 + gslot];
                nx_b = (uint32_t)lane < NP ? e.uit[(size_t)(b0 + gslot) * NP + lane] : 0u;
            }
            if (!slk_epoch_barrier(e, barriers + 1, s_flags + (barriers & 1u), s_wave_sums, nullptr)) return;
            ++barriers;
        }

        // ------------------------------------------------ USER PHASE
        float loss_acc = 0.0f;
        double loss_acc_d = 0.0;  // adaptive hinge: the columns' hinge terms (k_adaptive_select sums them in double)
        for (uint32_t p = b0 + gslot; p < b1 && !(e.debug & (1 | 8)); p += gstride) {
            const bool first = p == b0;
            const bool pre = p == b0 + gslot;  // this position's list entries were prefetched
            const uint32_t key = pre ? nx_key : e.ukey[p];
            const uint32_t prev = pre ? nx_prev : (first ? 0u : e.ukey[p - 1]);
            // pair: (positive item, negative item).  explicit: (item, index of the interaction's rating)
            uint32_t ip = 0u, in = 0u;
            if (!ADP) {
                ip = pre ? nx_a : e.uit[NP * (size_t)p];
                in = pre ? nx_b : (EXPL ? e.uk[p] : e.uit[2 * (size_t)p + 1]);
            }
            if (!first && prev == key) continue;  // not the head of its user's run
            const uint32_t user = key & e.umask;
            const size_t uoff = (size_t)user * D + d0;
            // every coherent load whose address is known goes out before the first use: ONE fabric round trip
            slk_vec<VEC> u = on ? slk_vload_coh<VEC>(e.P[0] + uoff) : zero;
            const slk_vec<VEC> su1 = (on && HAS_S1) ? slk_vload_coh<VEC>(e.S1[0] + uoff) : zero;
            const slk_vec<VEC> su2 = (on && HAS_S2) ? slk_vload_coh<VEC>(e.S2[0] + uoff) : zero;
            const float bu = slk_ld_coh(e.P[2] + user);
            slk_vec<1> bus1 = zero1, bus2 = zero1;
            if (lane == 0) {
                if (HAS_S1) bus1 = slk_vload_coh<1>(e.S1[2] + user);
                if (HAS_S2) bus2 = slk_vload_coh<1>(e.S2[2] + user);
            }
            slk_vec<VEC> vi = (on && !ADP) ? slk_vload_coh<VEC>(e.P[1] + (size_t)ip * D + d0) : zero;
            slk_vec<VEC> vj = (on && MODE == SLK_EMODE_PAIR) ? slk_vload_coh<VEC>(e.P[1] + (size_t)in * D + d0) : zero;
            float bi = ADP ? 0.0f : slk_ld_coh(e.P[3] + ip), bj = MODE == SLK_EMODE_PAIR ? slk_ld_coh(e.P[3] + in) : 0.0f;
            float rating = EXPL ? e.ratings[in] : 0.0f;  // an input of the call: plain load
            // The user's run is summed as the launch path sums it (slk_bilinear.hip, k_user_pass<ULONG> + k_user_stitch): in
            // occurrence order -- unless it is LONG, i.e. wholly covers an aligned tile of SLK_USER_TILE positions of the
            // minibatch: then tile by tile, the tiles' sums added in order.  Both sums are kept; which one applies is known
            // when the run ends.
            constexpr uint32_t TU = SLK_USER_TILE;
            slk_vec<VEC> gu = zero, tu = zero, gl = zero;
            float gbu = 0.0f, tbu = 0.0f, gbl = 0.0f;
            bool u_first_tile = true, u_long = false;
            const uint32_t up0 = p - b0;
            uint32_t q = p;
            for (;;) {
                if (on) slk_vstore_coh<VEC>(e.snap + (size_t)(q - b0) * e.RS + d0, u);
                if (ADP) {
                    // dL/dscore of this interaction's 1 + n pairs.  Pair 0 (the positive) belongs to column kc of the
                    // minibatch's [n, B] candidate matrix; draw j is entry f = kc * n + j of the flat n * B draws, which the
                    // reference's view(n, B) puts in row f / B of column f % B (implicit.py:266-275).  Lane s evaluates the
                    // column of pair s as k_adaptive_select does -- the positive's score, the n candidates, first maximum wins,
                    // x = best - positive + 1, g = 1/B where x >= 0 -- and keeps the pair's share: -g for the positive, g for
                    // the selected draw, 0 for the others.  Column kc's hinge term is counted here, once.
                    const bool qpre = pre && q == p;
                    const uint32_t kc = (qpre ? nx_a : e.uk[q]) - b0, bm = b1 - b0, nn = NP - 1u;
                    for (uint32_t s0 = 0; s0 < NP; s0 += (uint32_t)G) {
                        const uint32_t s = s0 + (uint32_t)lane;
                        // the pair's item id (lane s), issued with the scores: the rows of the live pairs follow at once
                        const uint32_t it_l = (qpre && s0 == 0u) ? nx_b : (s < NP ? e.uit[(size_t)q * NP + s] : 0u);
                        float gs = 0.0f;
                        if (s < NP) {
                            const uint32_t f = s == 0u ? 0u : kc * nn + (s - 1u);
                            const uint32_t col = s == 0u ? kc : f % bm, row = s == 0u ? 0u : f / bm;
                            const float sp = slk_ld_coh(e.sk + (size_t)(b0 + col) * NP);
                            float best = 0.0f;
                            uint32_t best_r = 0u;
                            for (uint32_t r = 0; r < nn; ++r) {
                                const uint32_t fr = r * bm + col;
                                const float sc = slk_ld_coh(e.sk + (size_t)(b0 + fr / nn) * NP + 1u + fr % nn);
                                if (r == 0u || sc > best) {
                                    best = sc;
                                    best_r = r;
                                }
                            }
                            const float x = best - sp + 1.0f;
                            const float g = x >= 0.0f ? inv_b : 0.0f;
                            if (s == 0u) {
                                gs = -g;
                                loss_acc_d += (double)(x > 0.0f ? x : 0.0f);
                            } else {
                                gs = best_r == row ? g : 0.0f;
                            }
                            slk_st_coh(e.gsn + (size_t)(q - b0) * NP + s, gs);
                        }
                        // the live pairs (the positive and the selected draws), in pair order as the launch path's user pass adds
                        // them, two rows in flight
                        unsigned long long live = slk_group_or<G>(gs != 0.0f ? 1ull << lane : 0ull);
                        while (live) {
                            const int j0 = __builtin_ctzll(live);
                            live &= live - 1ull;
                            const bool two = live != 0ull;
                            const int j1 = two ? __builtin_ctzll(live) : j0;
                            if (two) live &= live - 1ull;
                            const uint32_t i0 = __shfl(it_l, j0, G), i1 = __shfl(it_l, j1, G);
                            const float g0 = __shfl(gs, j0, G), g1 = __shfl(gs, j1, G);
                            const slk_vec<VEC> v0 = on ? slk_vload_coh<VEC>(e.P[1] + (size_t)i0 * D + d0) : zero;
                            const slk_vec<VEC> v1 = (on && two) ? slk_vload_coh<VEC>(e.P[1] + (size_t)i1 * D + d0) : zero;
                            slk_vaxpy<VEC>(gu, g0, v0);
                            gbu += g0;
                            slk_vaxpy<VEC>(tu, g0, v0);
                            tbu += g0;
                            if (two) {
                                slk_vaxpy<VEC>(gu, g1, v1);
                                gbu += g1;
                                slk_vaxpy<VEC>(tu, g1, v1);
                                tbu += g1;
                            }
                        }
                    }
                } else if (EXPL) {
                    const float sc = slk_group_sum<G>(slk_vdot<VEC>(u, vi)) + bu + bi;
                    float l, g;
                    slk_explicit_loss(e.loss_kind, sc, rating, inv_b, b1 - b0, l, g);
                    if (lane == 0) {
                        slk_st_coh(e.gsn + (size_t)(q - b0), g);
                        loss_acc += l;
                    }
                    if (g != 0.0f) {
                        slk_vaxpy<VEC>(gu, g, vi);
                        gbu += g;
                        slk_vaxpy<VEC>(tu, g, vi);
                        tbu += g;
                    }
                } else {
                    const float sp = slk_group_sum<G>(slk_vdot<VEC>(u, vi)) + bu + bi;
                    const float sn = slk_group_sum<G>(slk_vdot<VEC>(u, vj)) + bu + bj;
                    float l, gp, gn;
                    slk_pair_loss(e.loss_kind, sp, sn, inv_b, l, gp, gn);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float cc = gp * vi.v[i] + gn * vj.v[i];
                        gu.v[i] += cc;
                        tu.v[i] += cc;
                    }
                    gbu += gp + gn;
                    tbu += gp + gn;
                    if (lane == 0) {
                        slk_st_coh(e.gsn + 2 * (size_t)(q - b0), gp);
                        slk_st_coh(e.gsn + 2 * (size_t)(q - b0) + 1, gn);
                        loss_acc += l;
                    }
                }
                ++q;
                {
                    const bool run_ends = !(q < b1 && e.ukey[q] == key);
                    const uint32_t rel = q - b0;
                    const bool boundary = rel % TU == 0u;
                    if (run_ends || boundary) {  // the tile's part of the run is complete: it joins the run's tile-wise sum
                        const uint32_t tile_start = (rel - 1u) / TU * TU;
                        u_long = u_long || (tile_start >= up0 && boundary);
                        if (u_first_tile) {
                            gl = tu;
                            gbl = tbu;
                            u_first_tile = false;
                        } else {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) gl.v[i] += tu.v[i];
                            gbl += tbu;
                        }
                        tu = zero;
                        tbu = 0.0f;
                    }
                }
                if (!(q < b1 && e.ukey[q] == key)) break;
                if (ADP) continue;
                ip = e.uit[NP * (size_t)q];  // further occurrences of the same user in this minibatch
                vi = on ? slk_vload_coh<VEC>(e.P[1] + (size_t)ip * D + d0) : zero;
                bi = slk_ld_coh(e.P[3] + ip);
                if (EXPL) {
                    rating = e.ratings[e.uk[q]];
                } else {
                    in = e.uit[2 * (size_t)q + 1];
                    vj = on ? slk_vload_coh<VEC>(e.P[1] + (size_t)in * D + d0) : zero;
                    bj = slk_ld_coh(e.P[3] + in);
                }
            }

            if (u_long) {
                gu = gl;
                gbu = gbl;
            }
            if (on) slk_epoch_update<VEC, UPD>(e, c, 0, uoff, u, su1, su2, gu);
            if (lane == 0 && !(SLK_EUPD_ZERO_IS_NOOP(UPD) && gbu == 0.0f)) {  // zero gradient: an exact no-op for Adagrad
                slk_vec<1> bp, bg;
                bp.v[0] = bu;
                bg.v[0] = gbu;
                slk_epoch_update<1, UPD>(e, c, 2, user, bp, bus1, bus2, bg);
            }
        }
        // this row group's first position of the item phase
        const uint32_t ib0 = NP * b0, ib1 = NP * b1;
        if (DENSE) {
            // user rows this minibatch does not touch (stamped during the previous item phase / the prologue), and the stamps
            // of the items it does touch, for the item phase behind the barrier
            slk_epoch_sweep_untouched<VEC, UPD>(e, c, 0, 2, e.touch_u, mb + 1u, e.n_users, gslot, gstride, D, d0, on, lane);
            if (lane == 0)
                for (uint32_t r = ib0 + gslot; r < ib1; r += gstride)
                    __hip_atomic_store(e.touch_i + (e.ikey[r] & e.imask), mb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ib0 + gslot < ib1) {
            nx_key = e.ikey[ib0 + gslot];
            nx_prev = gslot ? e.ikey[ib0 + gslot - 1] : 0u;
            nx_a = e.ipay[ib0 + gslot];
        }
        {   // the workgroup's loss sum: waves by shuffle, the four wave sums by thread 0 on its way into the barrier
            double x = ADP ? loss_acc_d : (double)loss_acc;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
            if ((threadIdx.x & 63) == 0) s_wave_sums[threadIdx.x >> 6] = x;
        }
        if (!slk_epoch_barrier(e, barriers + 1, s_flags + (barriers & 1u), s_wave_sums,
                               e.partial + (size_t)mb * gridDim.x + blockIdx.x))
            return;
        ++barriers;

        // ------------------------------------------------ ITEM PHASE
        if (!(e.debug & (1 | 16)))
            slk_epoch_item_phase<VEC, G, UPD, EXPL ? SLK_EI_EXPL : (ADP ? SLK_EI_ADP : SLK_EI_PAIR)>(
                e, c, b0, ib0, ib1, NP, gslot, gstride, D, d0, on, lane, nx_key, nx_prev, nx_a);
        if (DENSE) {
            slk_epoch_sweep_untouched<VEC, UPD>(e, c, 1, 3, e.touch_i, mb + 1u, e.n_items, gslot, gstride, D, d0, on, lane);
            if (lane == 0 && mb + 1 < e.n_mb) {
                const uint32_t nb1 = (e.nc - b1 < e.bsz) ? e.nc : b1 + e.bsz;
                for (uint32_t p = b1 + gslot; p < nb1; p += gstride)
                    __hip_atomic_store(e.touch_u + (e.ukey[p] & e.umask), mb + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // this row group's first position of the next minibatch's user phase
        if (ADP && mb + 1 < e.n_mb) score_prefetch(b1, (e.nc - b1 < e.bsz) ? e.nc : b1 + e.bsz);
        if (!ADP && mb + 1 < e.n_mb && b1 + gslot < e.nc && gslot < e.bsz) {
            nx_key = e.ukey[b1 + gslot];
            nx_prev = gslot ? e.ukey[b1 + gslot - 1] : 0u;
            nx_a = e.uit[NP * (size_t)(b1 + gslot)];
            nx_b = EXPL ? e.uk[b1 + gslot] : e.uit[2 * (size_t)(b1 + gslot) + 1];
        }
        if (!slk_epoch_barrier(e, barriers + 1, s_flags + (barriers & 1u), s_wave_sums, nullptr)) return;
        ++barriers;
    }

    // loss.item() of every minibatch (implicit.py:240): the workgroups' partial sums, published on the way into each
    // user-phase barrier, are all visible now; one wave per minibatch adds them up
    for (uint32_t mb = blockIdx.x * NW + (threadIdx.x >> 6); mb < e.n_mb; mb += gridDim.x * NW) {
        const uint32_t b0 = mb * e.bsz, b1 = (e.nc - b0 < e.bsz) ? e.nc : b0 + e.bsz;
        double x = 0.0;
        for (unsigned i = threadIdx.x & 63u; i < gridDim.x; i += 64) {
            const unsigned long long bits = __hip_atomic_load(
                reinterpret_cast<const unsigned long long *>(e.partial + (size_t)mb * gridDim.x + i), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            double d;
            memcpy(&d, &bits, 8);
            x += d;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
        if ((threadIdx.x & 63) == 0) e.mb_loss[mb] = (float)(x * (double)(1.0f / (float)(b1 - b0)));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
typedef void (*slk_epoch_fn)(slk_epoch_args);

template <int VEC, int G, int MODE>
static slk_epoch_fn epoch_fn_of(int upd) {
    switch (upd) {
        case SLK_EUPD_ADAGRAD: return k_bilinear_epoch<VEC, G, SLK_EUPD_ADAGRAD, MODE>;
        case SLK_EUPD_SPARSE_ADAM: return k_bilinear_epoch<VEC, G, SLK_EUPD_SPARSE_ADAM, MODE>;
        case SLK_EUPD_ADAM_DENSE: return k_bilinear_epoch<VEC, G, SLK_EUPD_ADAM_DENSE, MODE>;
        case SLK_EUPD_SGD: return k_bilinear_epoch<VEC, G, SLK_EUPD_SGD, MODE>;
        default: return k_bilinear_epoch<VEC, G, SLK_EUPD_ADAGRAD_DENSE, MODE>;
    }
}
template <int VEC, int G>
static slk_epoch_fn epoch_fn(int upd, int mode) {
    if (mode == SLK_EMODE_EXPLICIT) return epoch_fn_of<VEC, G, SLK_EMODE_EXPLICIT>(upd);
    if (mode == SLK_EMODE_ADAPTIVE) return epoch_fn_of<VEC, G, SLK_EMODE_ADAPTIVE>(upd);
    return epoch_fn_of<VEC, G, SLK_EMODE_PAIR>(upd);
}
static int epoch_mode_of(int loss) {
    return loss >= SLK_LOSS_REGRESSION ? SLK_EMODE_EXPLICIT : (loss == SLK_LOSS_ADAPTIVE_HINGE ? SLK_EMODE_ADAPTIVE : SLK_EMODE_PAIR);
}

// Whether a slk_bilinear_train call takes the persistent route (option "epoch_kernel": 0 never, 1 when eligible).
bool slk_epoch_eligible(const slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t bsz, int loss,
                        bool bloom) {
    if (!ctx->opt_epoch_kernel || ctx->epoch_refused || bloom) return false;
    // one-negative pair losses, the explicit-feedback losses, adaptive hinge over ALL 1 + n occurrences (the chunk-sorted item
    // list; the launch path's per-minibatch live lists start at adaptive_late_min_batch, far above epoch_max_batch)
    if (loss == SLK_LOSS_ADAPTIVE_HINGE &&
        (!ctx->opt_epoch_adaptive || bsz > ctx->opt_epoch_adaptive_max_batch || bsz >= ctx->opt_adaptive_late_min_batch))
        return false;
    if (bsz > ctx->opt_epoch_max_batch) return false;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    // the dense optimizers rewrite every row every minibatch: in one launch of at most one workgroup per CU that
    // is only sensible for tables that are small next to the chip's caches
    if (dense && (tables->num_users + tables->num_items) * (int64_t)(tables->dim + 1) > ctx->opt_epoch_dense_elems) return false;
    return true;
}

enum { EP_COEF = 40, EP_BAR, EP_PARTIAL, EP_TOUCH };  // ctx->extra slots

static int epoch_upd_of(const slk_optim *optim) {
    switch (optim->kind) {
        case SLK_OPT_ADAGRAD: return SLK_EUPD_ADAGRAD;
        case SLK_OPT_SPARSE_ADAM: return SLK_EUPD_SPARSE_ADAM;
        case SLK_OPT_ADAM_DENSE: return SLK_EUPD_ADAM_DENSE;
        case SLK_OPT_SGD: return SLK_EUPD_SGD;
        default: return SLK_EUPD_ADAGRAD_DENSE;
    }
}

// Workgroups of the persistent launch: one position per row group in the (2x longer) item phase when the chip allows, at most
// one workgroup per CU, and never more than the device can hold RESIDENT at once for this kernel (occupancy query x CUs: the
// grid barrier needs every workgroup running; on gfx950 one 64-thread workgroup per CU always fits an idle device, the
// query guards a build whose register / LDS footprint says otherwise).
static unsigned epoch_grid(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t bsz, unsigned np, int g,
                           slk_epoch_fn fn, size_t lds) {
    const unsigned gpb = (unsigned)SLK_EPOCH_TB / (unsigned)g;
    unsigned grid = (unsigned)((np * bsz + gpb - 1) / gpb);
    if (optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE) {
        // the dense optimizers also sweep every row of the larger table once per phase: one row per row group if the chip allows
        const int64_t rows = tables->num_users > tables->num_items ? tables->num_users : tables->num_items;
        const unsigned by_rows = (unsigned)((rows + gpb - 1) / gpb);
        if (by_rows > grid) grid = by_rows;
    }
    unsigned cap = (unsigned)ctx->num_cus < (unsigned)ctx->opt_epoch_max_grid ? (unsigned)ctx->num_cus : (unsigned)ctx->opt_epoch_max_grid;
    // measurement switch (round 5, the mid-size single-launch A/B: profiles/r05_*): "epoch_max_grid" above 1024 lifts the
    // one-workgroup-per-CU limit -- several one-wave workgroups per CU, still bounded by what the device holds resident
    if (ctx->opt_epoch_max_grid > 1024) cap = (unsigned)ctx->opt_epoch_max_grid;
    int per_cu = 0;
    if (fn && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, SLK_EPOCH_TB, lds) == hipSuccess && per_cu >= 0) {
        const unsigned resident = (unsigned)per_cu * (unsigned)ctx->num_cus;
        if (resident < cap) cap = resident;  // 0: the kernel cannot be resident at all -> the caller takes the launch path
    } else {
        (void)hipGetLastError();
        // residency unknown: the lifted cap is not trusted -- back to at most one workgroup per CU, which an idle
        // gfx950 always holds resident (a grid barrier over workgroups that cannot all run never completes)
        if (cap > (unsigned)ctx->num_cus) cap = (unsigned)ctx->num_cus;
    }
    if (grid > cap) grid = cap;
    return grid;
}

static slk_epoch_fn epoch_pick_fn(int vec, int g, int upd, int mode) {
    slk_epoch_fn fn = nullptr;
#define SLK_PICK_EPOCH(V_, G_) fn = epoch_fn<V_, G_>(upd, mode)
    SLK_FOR_LAYOUT(vec, g, SLK_PICK_EPOCH);
#undef SLK_PICK_EPOCH
    return fn;
}

static const size_t kEpochLds = 4 * sizeof(double) + 16;

// Scratch of the persistent route for chunks of up to n_mb minibatches: called by slk_bilinear_reserve (so that the training
// call allocates nothing) and again, idempotently, by every slk_epoch_run_chunk.
int slk_epoch_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, uint32_t n_mb, int64_t bsz, int loss, int NP) {
    int vec, g, rc;
    if (!slk_pick_layout(tables->dim, &vec, &g)) return slk_fail(ctx, SLK_EINVAL, "embedding dim %d unsupported", tables->dim);
    const unsigned grid = epoch_grid(ctx, tables, optim, bsz, (unsigned)NP, g, epoch_pick_fn(vec, g, epoch_upd_of(optim), epoch_mode_of(loss)),
                                     kEpochLds);
    if ((rc = slk_ensure(ctx, ctx->extra[EP_COEF], (size_t)n_mb * sizeof(slk_step_coef)))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EP_BAR], 2048))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EP_PARTIAL], (size_t)n_mb * (grid ? grid : 1) * 8))) return rc;
    if (optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE) {
        const size_t rows = (size_t)tables->num_users + (size_t)tables->num_items;
        if ((rc = slk_ensure(ctx, ctx->extra[EP_TOUCH], rows * 4))) return rc;
    }
    // pinned, double-buffered host staging of the per-minibatch coefficients (an asynchronous copy from pageable memory may
    // read the host buffer after the call returned -- and the next chunk's call rewrites it)
    for (int b = 0; b < 2; ++b) {
        if (ctx->h_coef_cap[b] >= n_mb) continue;
        if (ctx->ev_coef[b]) SLK_HIP(ctx, hipEventSynchronize(ctx->ev_coef[b]));
        if (ctx->h_coef[b]) SLK_HIP(ctx, hipHostFree(ctx->h_coef[b]));
        ctx->h_coef[b] = nullptr;
        ctx->h_coef_cap[b] = 0;
        const size_t cap = n_mb < 1024 ? 1024 : n_mb;
        SLK_HIP(ctx, hipHostMalloc(&ctx->h_coef[b], cap * sizeof(slk_step_coef), hipHostMallocDefault));
        ctx->h_coef_cap[b] = cap;
        if (!ctx->ev_coef[b]) SLK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_coef[b], hipEventDisableTiming));
    }
    return SLK_OK;
}

// All minibatches of one prepared chunk (sorted lists in pb, see slk_bilinear.hip) in one persistent launch.
int slk_epoch_run_chunk(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, const slk_prep_bufs &pb, uint32_t nc,
                        int64_t bsz, unsigned ubits, unsigned ibits, int loss, int NP, int RS, float *snap, float *gsn,
                        float *d_mb_loss, const float *d_ratings, hipStream_t s) {
    const int mode = epoch_mode_of(loss);
    const bool expl = mode == SLK_EMODE_EXPLICIT;  // d_ratings: the chunk's ratings, indexed by pb.uval[1] (see do_sort)
    const bool adp = mode == SLK_EMODE_ADAPTIVE;   // NP = 1 + negatives per interaction (pair losses 2, explicit 1)
    int vec, g, rc;
    if (!slk_pick_layout(tables->dim, &vec, &g)) return slk_fail(ctx, SLK_EINVAL, "embedding dim %d unsupported", tables->dim);
    const uint32_t n_mb = (uint32_t)((nc + bsz - 1) / bsz);
    const int upd = epoch_upd_of(optim);
    slk_epoch_fn fn = epoch_pick_fn(vec, g, upd, mode);
    const size_t lds = kEpochLds;
    const unsigned grid = epoch_grid(ctx, tables, optim, bsz, (unsigned)NP, g, fn, lds);
    if (grid == 0) {  // the occupancy query says no workgroup of this kernel fits a CU: nothing ran, take the launch path
        slk_fail(ctx, SLK_EIO, "k_bilinear_epoch cannot be resident on this device (occupancy 0)");
        ctx->epoch_refused = true;
        return SLK_EAGAIN_EPOCH;
    }
    if ((rc = slk_epoch_reserve(ctx, tables, optim, n_mb, bsz, loss, NP))) return rc;

    // per-step coefficients, in double like torch (slk_set_opt_coeffs / slk_dense_sweeps)
    const int cb = ctx->coef_flip;
    ctx->coef_flip ^= 1;
    SLK_HIP(ctx, hipEventSynchronize(ctx->ev_coef[cb]));  // the copy issued from this buffer two chunks ago (never recorded: returns at once)
    slk_step_coef *hc = (slk_step_coef *)ctx->h_coef[cb];
    for (uint32_t m = 0; m < n_mb; ++m) {
        const double step = (double)(optim->step + 1 + m);
        slk_step_coef &c = hc[m];
        c.c0 = c.c1 = 0.0f;
        if (upd == SLK_EUPD_SGD) {
            c.c0 = (float)optim->lr;
        } else if (upd == SLK_EUPD_ADAGRAD || upd == SLK_EUPD_ADAGRAD_DENSE) {
            c.c0 = (float)(optim->lr / (1.0 + (step - 1.0) * optim->lr_decay));
        } else {
            const double bc1 = 1.0 - pow(optim->beta1, step), bc2 = 1.0 - pow(optim->beta2, step);
            if (upd == SLK_EUPD_SPARSE_ADAM) {
                c.c0 = (float)(optim->lr * sqrt(bc2) / bc1);
            } else {
                c.c0 = (float)(optim->lr / bc1);
                c.c1 = (float)sqrt(bc2);
            }
        }
    }
    SLK_HIP(ctx, hipMemcpyAsync(ctx->extra[EP_COEF].p, hc, (size_t)n_mb * sizeof(slk_step_coef), hipMemcpyHostToDevice, s));
    SLK_HIP(ctx, hipEventRecord(ctx->ev_coef[cb], s));
    SLK_HIP(ctx, hipMemsetAsync(ctx->extra[EP_BAR].p, 0, 2048, s));
    const bool dense_opt = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (dense_opt) {
        const size_t rows = (size_t)tables->num_users + (size_t)tables->num_items;
        SLK_HIP(ctx, hipMemsetAsync(ctx->extra[EP_TOUCH].p, 0, rows * 4, s));
    }

    slk_epoch_args e;
    memset(&e, 0, sizeof(e));
    for (int t = 0; t < 4; ++t) {
        e.P[t] = tables->d_param[t];
        e.S1[t] = optim->d_state1[t];
        e.S2[t] = optim->d_state2[t];
    }
    e.D = tables->dim;
    e.n_users = (uint32_t)tables->num_users;
    e.n_items = (uint32_t)tables->num_items;
    e.nc = nc;
    e.bsz = (uint32_t)bsz;
    e.n_mb = n_mb;
    e.ukey = (const uint32_t *)pb.ukey[1].p;
    e.uit = (expl || adp) ? (const uint32_t *)pb.uit.p : (const uint32_t *)pb.uval[1].p;
    e.uk = (expl || adp) ? (const uint32_t *)pb.uval[1].p : nullptr;
    e.NP = NP;
    e.sk = adp ? (float *)ctx->sk.p : nullptr;
    ctx->epoch_bars_per_mb = adp ? 3 : 2;
    e.ratings = d_ratings;
    e.umask = (uint32_t)((1ull << ubits) - 1);
    e.ikey = (const uint32_t *)pb.ikey[1].p;
    e.ipay = (const uint32_t *)pb.ipay[1].p;
    e.imask = (uint32_t)((1ull << ibits) - 1);
    e.snap = snap;
    e.RS = RS;
    e.gsn = gsn;
    e.partial = (double *)ctx->extra[EP_PARTIAL].p;
    e.mb_loss = d_mb_loss;
    e.coef = (const slk_step_coef *)ctx->extra[EP_COEF].p;
    e.touch_u = dense_opt ? (uint32_t *)ctx->extra[EP_TOUCH].p : nullptr;
    e.touch_i = dense_opt ? e.touch_u + tables->num_users : nullptr;
    e.bar = (unsigned *)ctx->extra[EP_BAR].p;
    e.status = &ctx->d_rng->epoch_abort;
    e.loss_kind = loss;
    e.eps = (float)optim->eps;
    e.omb1 = (float)(1.0 - optim->beta1);
    e.omb2 = (float)(1.0 - optim->beta2);
    e.beta2 = (float)optim->beta2;
    e.wd = (float)optim->weight_decay;
    // two levels pay above ~128 arrivals (profiles/r02_d_epoch_kernel_anatomy.jsonl: 256 workgroups 9.3 -> 6.0 us per pair)
    e.bar_kind = ctx->opt_epoch_barrier >= 0 ? ctx->opt_epoch_barrier : (grid > 128 ? 1 : 0);
    e.debug = ctx->opt_epoch_debug;

    slk_prof_begin(ctx, SLK_K_EPOCH, s);
    void *kargs[1] = {&e};
    // The grid (<= one wavefront-sized workgroup per CU) is resident as a whole on any device that is not saturated by other
    // work for seconds, which is what the barrier's bounded spins tolerate.  A plain launch lets kernels of other streams
    // (the next epoch's shuffle and negatives, implicit.py's pipelined fit) run beside it; the cooperative form adds a
    // launch-time residency check but serialises the device.
    hipError_t le = hipSuccess;
    if (ctx->opt_epoch_cooperative) {
        le = hipLaunchCooperativeKernel(fn, dim3(grid), dim3(SLK_EPOCH_TB), kargs, lds, s);
    } else {
        SLK_RESIDENT_GRID_LAUNCH();
        hipLaunchKernelGGL(fn, dim3(grid), dim3(SLK_EPOCH_TB), lds, s, e);
        le = hipGetLastError();
    }
    slk_prof_end(ctx, s);
    if (le != hipSuccess) {
        // the grid cannot be resident at once (device shared with other work, cooperative launches unsupported): nothing
        // has run -- the caller takes the per-minibatch launches for this chunk and stops asking
        (void)hipGetLastError();
        slk_fail(ctx, SLK_EIO, "cooperative launch of k_bilinear_epoch (%u workgroups) refused: %s", grid, hipGetErrorString(le));
        ctx->epoch_refused = true;
        return SLK_EAGAIN_EPOCH;
    }
    optim->step += n_mb;
    return SLK_OK;
}
