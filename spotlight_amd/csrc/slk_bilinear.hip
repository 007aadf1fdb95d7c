// slk_bilinear.hip -- the fused negative-sampled BilinearNet training step for gfx950.
//
// Replaces the minibatch body of ImplicitFactorizationModel.fit()
// (spotlight/factorization/implicit.py:229-243): 2x BilinearNet.forward
// (factorization/representations.py:61-91), the loss (losses.py:18-166), autograd's
// embedding backward (duplicate rows SUMMED before the non-linear optimizer update) and
// torch.optim.{Adagrad,SparseAdam,Adam}.step().
//
// Exactness constraint that shapes the design (hard part 1): within a minibatch
// every forward reads pre-step parameters, and every row receives the SUM of its gradient
// contributions before one optimizer update.  So per minibatch:
//
//   prep (per chunk of minibatches, slk_prep): radix-sort the interactions by
//        (minibatch, user) and the (interaction, item) occurrences by (minibatch, item).
//        Sorted order makes each unique row the property of exactly ONE row group, turns
//        row traffic into ascending-address streams, and needs no atomics.
//   USER PASS  one row group (G = dim/4 lanes, 16 B per lane, a D=64 row = one 256-B line)
//        per unique user: gathers U[u], V[pos], V[neg] and the biases, wave-shuffle dot
//        products, loss + dL/dscore, accumulates the user-row gradient over the user's
//        occurrences, snapshots the pre-step user row (coalesced, position-indexed) for the
//        item pass, writes dL/dscore per occurrence, then applies the optimizer to U[u] in
//        place.
//   ITEM PASS  one row group per unique item: sums g * snapshot(user row) over the item's
//        occurrences (positive or negative), then applies the optimizer to V[i] in place.
//
// The path is HBM-bound (0.2 flop/B): no MFMA, no LDS tiling of rows -- each row is touched
// once per pass by the lanes that own it.
#include <math.h>

#include "slk_kernels.h"

// ---------------------------------------------------------------------------------------
// USER PASS
// ---------------------------------------------------------------------------------------
// BLOOM: the user and/or item embedding layer is a BloomEmbedding (slk_kernels.h).  Item vectors
// are then sums of hashed rows; a bloom USER vector is shared between users, so its gradient is
// not applied here but parked in a.urec for a ROW-mode owner pass over the hashed user rows.
// UMODE: 0 = the (positive, negative) pair's scores, loss and dL/dscore formed here; 1 = dL/dscore read from a.gk
// (adaptive hinge, staged explicit route); 2 = explicit feedback, fused: one pair, score + loss against a.ratings
// LAT (pair mode, plain tables): the latency-bound form for minibatches that do not fill the chip (a few thousand to ~10^5
// interactions: every row group then handles one or two positions and the pass is a chain of dependent round trips, not a
// stream).  The position's item pair is fetched WITH its key (before the head test: wasted for the few non-head positions)
// and the user row's optimizer state with the row, so a position costs two round trips -- (key, pair), then (user row +
// state, both item rows, the three biases) -- instead of four.  Same arithmetic in the same order: bit-identical results.
// ULONG (plain user table): HOT USERS.  A user's occurrences in the minibatch are walked by ONE row group; a power user
// with thousands of them is a serial tail longer than the whole pass.  As in the item pass, a run that wholly covers an
// aligned tile of SLK_USER_TILE positions is LONG: it is cut at the tile boundaries, every segment is walked by its own row
// group from the same pre-step user row (records, dL/dscore and loss terms as usual) and leaves its part of the user
// gradient as a PARTIAL (a.upart / upart_meta, at most two per tile); nobody updates U[u] here -- k_user_stitch, behind the
// pass, adds a long run's partials in tile order and applies the one update.  Every position of a run decides "long" the
// same way (a head looks at the first aligned tile behind it, a boundary position at the tile before and the tile behind
// it).  Short runs are walked and applied exactly as in the plain form; the host launches this form only for minibatches
// that hold a long run (k_user_long_flags).
// PP (pair mode, plain tables): inside a user-row ping-pong scope (slk_user_pingpong_begin; slk_pass_args::uflag / P0alt) --
// the user's current row is read from the copy its flag byte names, the updated row goes to the OTHER copy, the byte is
// flipped, and no record is written: the item pass (SLK_ITEM_SNAPPP) gathers the pre-step row from where it still stands.
#ifndef SLK_PP_PIPE
#define SLK_PP_PIPE 2  // software pipeline of the bandwidth-bound ping-pong user pass (below): 0 none, 1 keys + flag, 2 + item pair, 3 + early state.  Same box, C2 user pass: one-table form 0.3135 ms, 0: 0.306, 1: 0.289, 2: 0.284, 3: 0.295 (profiles/r06_p_*)
#endif
// (Holding the pipelined form to the one-table form's occupancy -- amdgpu_waves_per_eu 8: 64 VGPRs instead of 65, SparseAdam 7: 72
// instead of 81 -- spills two to eight dwords and LOSES: user pass 0.283 -> 0.289 ms, SparseAdam 0.422 -> 0.467, profiles/r06_t_ab_occ.txt.)
// SGL (the bandwidth-bound ping-pong form, Adagrad): the single-occurrence fast path (slk_pass_args::mflag) -- an item that occurs
// ONCE in the minibatch is updated HERE: the pass holds its pre-step row (read for the score), dL/dscore and the pre-step user row,
// so the item pass's re-read of the row and its gather of the user row are spared (catalogues far larger than a minibatch: 98 % of
// the occurrences at the C5 shard's shape).  The same operations in the same order as the item pass applies to a run of length
// one (0 + g * u, then slk_apply_vec_pre): bit-identical tables.  Nobody else reads or writes such a row in this minibatch.
// (The form compiles to 102 VGPRs = 4 waves per SIMD.  Held to 5 -- amdgpu_waves_per_eu: 96 VGPRs, 28 B of scratch -- it LOSES: C5 shard
// user pass 0.683-0.689 -> 0.721-0.724 ms, profiles/r06_z3_ab_sgl_waves.txt.)
template <int VEC, int G, int UPD, int UMODE, bool BLOOM, bool LAT = false, bool ULONG = false, bool PP = false, bool SGL = false>
__global__ __launch_bounds__(256) void k_user_pass(slk_pass_args a) {
    static_assert(!LAT || (UMODE == 0 && !BLOOM), "LAT is the pair mode over plain tables");
    static_assert(!PP || (UMODE == 0 && !BLOOM), "PP is the pair mode over plain tables");
    static_assert(!SGL || (PP && !LAT && !ULONG && UPD == SLK_UPD_ADAGRAD), "SGL: the plain ping-pong form, Adagrad");
    static_assert(!ULONG || !BLOOM, "long user runs: plain tables");
    constexpr uint32_t S = SLK_USER_TILE;
    constexpr bool PRE = UMODE != 0;
    constexpr bool EXPL = UMODE == 2;
    __shared__ double red[256];
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t stride = gridDim.x * GPB;
    float loss_acc = 0.0f;

    // Plain form: row group g takes positions g, g + stride, ... (one position per turn).  ULONG: row group g takes TILES g,
    // g + stride, ... and walks the positions of a tile itself, run piece by run piece.  A segment of a long run is one tile's
    // worth of occurrences walked by ONE row group; with one position per turn the segment heads (every SLK_USER_TILE-th
    // position) land on one row group of a wavefront while its other three idle behind the divergent walk -- measured,
    // profiles/r03_c_*, r03_d_*: Zipf(1.0) users 2.0-2.9 ms per pass.  With a tile per row group the four groups of a
    // wavefront each walk their own tile, hot or not, at the plain form's memory parallelism.
    const uint32_t n_turns = ULONG ? (a.end - a.begin + S - 1u) / S : a.end - a.begin;
    // PIPE (the bandwidth-bound ping-pong form): a position costs key -> flag byte -> row, one dependent round trip more than the
    // one-table form's key -> row.  The row group's turns are therefore software-pipelined: entering turn t it holds {key, previous
    // key, flag} of turn t and {key, previous key} of turn t + 1; it issues the flag load of turn t + 1 and the key loads of turn
    // t + 2 in front of turn t's own row loads.  (Only the user's owner -- this group, in turn t + 1 -- writes that flag byte.)
    // SLK_PP_PIPE >= 2: the turn's (positive, negative) item pair travels with its key, so that a head issues its user row, both
    // item rows and the biases in ONE round trip; >= 3: and the row's optimizer state with them (as the latency-bound form does).
    constexpr bool PIPE = PP && !LAT && !ULONG && SLK_PP_PIPE >= 1;
    constexpr bool PIPE_IDS = PIPE && SLK_PP_PIPE >= 2;
    uint32_t pk_key = 0u, pk_prev = 0u, pk_flag = 0u, pn_key = 0u, pn_prev = 0u;
    uint32_t pk_ip = 0u, pk_in = 0u, pn_ip = 0u, pn_in = 0u;
    uint32_t pk_mf = 0u, pn_mf = 0u;  // SGL: the turn's two "occurs more than once" bytes (positive | negative << 8)
    auto ld_mf = [&](uint32_t pos) -> uint32_t { return (uint32_t)*reinterpret_cast<const uint16_t *>(a.mflag + 2 * (size_t)pos); };
    if (PIPE) {
        const uint32_t t0 = blockIdx.x * GPB + grp;
        if (t0 < n_turns) {
            pk_key = a.ukey[a.begin + t0];
            pk_prev = t0 > 0u ? a.ukey[a.begin + t0 - 1u] : 0u;
            if (PIPE_IDS) {
                pk_ip = a.uit[2 * (size_t)(a.begin + t0)];
                pk_in = a.uit[2 * (size_t)(a.begin + t0) + 1];
            }
            if (SGL) pk_mf = ld_mf(a.begin + t0);
            pk_flag = (uint32_t)a.uflag[pk_key & a.umask];
        }
        if (t0 + stride < n_turns) {
            pn_key = a.ukey[a.begin + t0 + stride];
            pn_prev = a.ukey[a.begin + t0 + stride - 1u];
            if (PIPE_IDS) {
                pn_ip = a.uit[2 * (size_t)(a.begin + t0 + stride)];
                pn_in = a.uit[2 * (size_t)(a.begin + t0 + stride) + 1];
            }
            if (SGL) pn_mf = ld_mf(a.begin + t0 + stride);
        }
    }
    for (uint32_t turn = blockIdx.x * GPB + grp; turn < n_turns; turn += stride) {
    const uint32_t p_lo = a.begin + (ULONG ? turn * S : turn);
    const uint32_t p_hi = ULONG ? (p_lo + S < a.end ? p_lo + S : a.end) : p_lo + 1u;
    for (uint32_t p = p_lo; p < p_hi; ++p) {
        const bool nt_keys = (SLK_NT_OF(a) & 8) != 0;
        const uint32_t key = PIPE ? pk_key : slk_ld_u32(a.ukey + p, nt_keys);
        uint32_t lat_ip = 0u, lat_in = 0u;
        bool is_head;
        uint32_t cur_piped = 0u, mf_piped = 0u;
        if (PIPE) {
            is_head = !(p > a.begin && pk_prev == key);
            cur_piped = pk_flag;
            lat_ip = pk_ip;
            lat_in = pk_in;
            mf_piped = pk_mf;
            pk_mf = pn_mf;
            pk_key = pn_key;
            pk_prev = pn_prev;
            pk_ip = pn_ip;
            pk_in = pn_in;
            if (turn + stride < n_turns) pk_flag = (uint32_t)a.uflag[pn_key & a.umask];
            if (turn + 2u * stride < n_turns) {
                pn_key = a.ukey[p + 2u * stride];
                pn_prev = a.ukey[p + 2u * stride - 1u];
                if (PIPE_IDS) {
                    pn_ip = a.uit[2 * (size_t)(p + 2u * stride)];
                    pn_in = a.uit[2 * (size_t)(p + 2u * stride) + 1];
                }
                if (SGL) pn_mf = ld_mf(p + 2u * stride);
            }
        } else if (LAT) {
            const uint32_t prev = p > a.begin ? a.ukey[p - 1] : ~key;
            lat_ip = a.uit[2 * (size_t)p];
            lat_in = a.uit[2 * (size_t)p + 1];
            is_head = prev != key;
        } else {
            is_head = !(p > a.begin && a.ukey[p - 1] == key);
        }
        bool run_long = false;
        uint32_t seg_limit = a.end;  // the walk stops here at the latest
        if (ULONG) {
            const uint32_t rel = p - a.begin;
            if (is_head) {
                const uint32_t t0 = a.begin + (rel + S - 1u) / S * S;  // the first aligned tile behind the head
                run_long = t0 + S <= a.end && a.ukey[t0 + S - 1u] == key;
            } else if (rel % S == 0u) {
                run_long = (rel >= S && a.ukey[p - S] == key) || (p + S <= a.end && a.ukey[p + S - 1u] == key);
                is_head = false;
            }
            if (!is_head && !(run_long && rel % S == 0u)) continue;
            if (run_long) seg_limit = a.begin + (rel / S + 1u) * S < a.end ? a.begin + (rel / S + 1u) * S : a.end;
        } else if (!is_head) {
            continue;  // not the head of its user segment
        }
        const uint32_t user = key & a.umask;
        const size_t uoff = (size_t)user * D + d0;
        // user-row ping-pong (slk_pass_args::uflag): the copy that holds this user's current row; the updated row goes to the
        // other copy, the item pass gathers the pre-step row from this one -- no record is written
        constexpr bool pp = PP;
        const uint32_t cur = PIPE ? cur_piped : (pp ? (uint32_t)a.uflag[user] : 0u);
        float *const unew = pp ? (cur ? a.P[0] : a.P0alt) + uoff : nullptr;
        slk_vec<VEC> u;
        if (BLOOM)
            u = slk_emb_vec<VEC>(a.P[0], a.ub, user, D, d0, on);
        else
            u = on ? slk_vload_if_nt<VEC>((cur ? a.P0alt : a.P[0]) + uoff, (SLK_NT_OF(a) & 1) != 0) : slk_vzero<VEC>();
        const float bu = (UMODE == 0 && !BLOOM && a.ubz) ? 0.0f : a.P[2][user];  // (SLK_TABLES_USER_BIAS_ZERO: a line per interaction for a table of zeros)
        // explicit feedback: the pass is bound by its chain of dependent loads, so the Adagrad state of the
        // user row is fetched with the row instead of after the loss (the pair mode is bandwidth-bound:
        // fetching early there only lengthens register lifetimes)
        // SparseAdam (round 4): both moment rows (and the bias's moments) with the row in every mode -- its update otherwise
        // starts with a dependent round trip of TWO more rows per user behind the loss (the pass ran at 0.45 of the roofline
        // against Adagrad's 0.63, profiles/r03_final_bench_sparse_adam.json)
        constexpr bool EARLY_ADAM = !BLOOM && UPD == SLK_UPD_SPARSE_ADAM && !ULONG;
        // (SGL: the pass is one round trip per position -- user row, both item rows AND every optimizer-state row the position will
        // update -- then arithmetic and stores; loaded one after the other behind the loss the three state rows cost it three more)
        constexpr bool EARLY_STATE = ((EXPL || LAT || SGL || (PIPE && SLK_PP_PIPE >= 3)) && !BLOOM && UPD == SLK_UPD_ADAGRAD) || EARLY_ADAM;
        slk_vec<VEC> su = slk_vzero<VEC>(), su2 = slk_vzero<VEC>();
        if (EARLY_STATE && on) su = slk_vload_if_nt<VEC>(a.S1[0] + uoff, (SLK_NT_OF(a) & 1) != 0);
        if (EARLY_ADAM && on) su2 = slk_vload_if_nt<VEC>(a.S2[0] + uoff, (SLK_NT_OF(a) & 1) != 0);
        float sbu = 0.0f, sbu2 = 0.0f;
        if (EARLY_STATE && !(SGL && a.ubz)) sbu = a.S1[2][user];  // (SLK_TABLES_USER_BIAS_ZERO: the user-bias gradient is exactly zero, its state is never used)
        if (EARLY_ADAM) sbu2 = a.S2[2][user];
        slk_vec<VEC> gu = slk_vzero<VEC>();
        float gbu = 0.0f;
        uint32_t q = p;
        // the key after the current position travels one iteration ahead of its use (the segment-end test)
        // (latency-bound modes only: same-box A/B, +5 % on the adaptive pass, -1 % on the bandwidth-bound pair pass)
        constexpr bool NEXT_KEY = UMODE != 0;
        uint32_t next_key = 0;
        if (NEXT_KEY) next_key = (q + 1 < seg_limit) ? a.ukey[q + 1] : ~key;
        do {
            float *rec = a.snap + (size_t)(q - a.begin) * a.RS;
            // explicit feedback: the rating's gather is issued first so that it overlaps the item row's
            uint32_t e_item = 0;
            float e_rating = 0.0f;
            if (EXPL) {
                e_item = a.uit[q];
                e_rating = a.ratings[a.uk[q]];
            }
            if (on && !pp) slk_vstore_if_nt<VEC>(rec + d0, u, (SLK_NT_OF(a) & 16) != 0);
            if (!PRE) {
                uint32_t ip, in;
                if ((LAT || PIPE_IDS) && q == p) {
                    ip = lat_ip;
                    in = lat_in;
                } else {
                    ip = slk_ld_u32(a.uit + 2 * (size_t)q, nt_keys);
                    in = slk_ld_u32(a.uit + 2 * (size_t)q + 1, nt_keys);
                }
                slk_vec<VEC> vi, vj;
                if (BLOOM) {
                    vi = slk_emb_vec<VEC>(a.P[1], a.ib, ip, D, d0, on);
                    vj = slk_emb_vec<VEC>(a.P[1], a.ib, in, D, d0, on);
                } else {
                    vi = on ? slk_vload<VEC>(a.P[1] + (size_t)ip * D + d0) : slk_vzero<VEC>();
                    vj = on ? slk_vload<VEC>(a.P[1] + (size_t)in * D + d0) : slk_vzero<VEC>();
                }
                const float bip = a.P[3][SLK_B3(a, ip)], bin = a.P[3][SLK_B3(a, in)];  // issued with the rows, not behind the dots' shuffles
                // SGL: the optimizer state of the once-only items of this position, with their rows
                uint32_t mf = 0xffffu;
                slk_vec<VEC> svi = slk_vzero<VEC>(), svj = slk_vzero<VEC>();
                float sbi = 0.0f, sbj = 0.0f;
                if (SGL) {
                    mf = (q == p) ? mf_piped : ld_mf(q);
                    const bool nt_rows = (SLK_NT_OF(a) & 2) != 0;
                    if (!(mf & 0xffu)) {
                        if (on) svi = slk_vload_if_nt<VEC>(a.S1[1] + (size_t)ip * D + d0, nt_rows);
                        if (lane == 0) sbi = a.S1[3][SLK_B3(a, ip)];
                    }
                    if (!(mf >> 8)) {
                        if (on) svj = slk_vload_if_nt<VEC>(a.S1[1] + (size_t)in * D + d0, nt_rows);
                        if (lane == 0) sbj = a.S1[3][SLK_B3(a, in)];
                    }
                }
                const float sp = slk_group_sum<G>(slk_vdot<VEC>(u, vi)) + bu + bip;
                const float sn = slk_group_sum<G>(slk_vdot<VEC>(u, vj)) + bu + bin;
                float l, gp, gn;
                slk_pair_loss(a.loss_kind, sp, sn, a.inv_b, l, gp, gn);
#pragma unroll
                for (int i = 0; i < VEC; ++i) gu.v[i] += gp * vi.v[i] + gn * vj.v[i];
                gbu += gp + gn;
                if (lane == 0) {
                    if (pp) {  // {dL/dscore, src} per pair: one 16-B store
                        float srcf;
                        const uint32_t src = user | (cur << 31);
                        memcpy(&srcf, &src, 4);
                        reinterpret_cast<float4 *>(a.gsn)[q - a.begin] = make_float4(gp, srcf, gn, srcf);
                    } else {
                        float *gq = a.gsn + 2 * (size_t)(q - a.begin);
                        gq[0] = gp;
                        gq[1] = gn;
                    }
                    loss_acc += l;
                }
                if (SGL) {
                    // once-only items of this position: their whole update, as the item pass applies it to a run of length one --
                    // nothing for dL/dscore == 0 (Adagrad: the item pass leaves an untouched-gradient row alone), else the summed
                    // gradient 0 + g * u_old and 0 + g, then the row / bias update in place (the rows and biases are this pass's own
                    // pre-step loads)
                    const bool nt_rows = (SLK_NT_OF(a) & 2) != 0;
                    auto once = [&](uint32_t item, slk_vec<VEC> &v, slk_vec<VEC> &sv, float g, float b, float bs) {
                        const size_t voff = (size_t)item * D + d0;
                        if (on) {
                            slk_vec<VEC> gv;
#pragma unroll
                            for (int i = 0; i < VEC; ++i) gv.v[i] = 0.0f + g * u.v[i];
                            slk_apply_vec_pre<VEC, UPD>(a, 1, voff, v, sv, gv, nullptr, nt_rows);
                        }
                        if (lane == 0) {
                            slk_vec<1> bpv, bsv, gbv;
                            bpv.v[0] = b;
                            bsv.v[0] = bs;
                            gbv.v[0] = 0.0f + g;
                            slk_apply_vec_pre<1, UPD>(a, 3, SLK_B3(a, item), bpv, bsv, gbv);
                        }
                    };
                    if (!(mf & 0xffu) && gp != 0.0f) once(ip, vi, svi, gp, bip, sbi);
                    if (!(mf >> 8) && gn != 0.0f) once(in, vj, svj, gn, bin, sbj);
                }
            } else if (EXPL) {
                // explicit feedback (one pair per interaction): score, loss and dL/dscore formed here
                const uint32_t it = e_item;
                const float bi = a.P[3][SLK_B3(a, it)];  // issued with the row, not after the dot's shuffles
                slk_vec<VEC> v;
                if (BLOOM)
                    v = slk_emb_vec<VEC>(a.P[1], a.ib, it, D, d0, on);
                else
                    v = on ? slk_vload<VEC>(a.P[1] + (size_t)it * D + d0) : slk_vzero<VEC>();
                const float sc = slk_group_sum<G>(slk_vdot<VEC>(u, v)) + bu + bi;
                float l, g;
                slk_explicit_loss(a.loss_kind, sc, e_rating, a.inv_b, a.end - a.begin, l, g);
                if (lane == 0) {
                    a.gsn[q - a.begin] = g;
                    loss_acc += l;
                }
                if (g != 0.0f) {
                    slk_vaxpy<VEC>(gu, g, v);
                    gbu += g;
                }
            } else {
                // dL/dscore of the interaction's 1 + n pairs, one per lane (one round trip for all of them, with the pairs' item
                // ids), then the rows of the LIVE pairs -- adaptive hinge: the positive and the selected draw -- two at a time,
                // added in pair order (the order the one-pair-at-a-time loop this replaces added them in: same bits; a pair per
                // iteration cost a dependent dL/dscore -> id -> row chain each)
                const size_t kb = (size_t)a.uk[q] * a.NP, qb = (size_t)q * a.NP;
                for (int s0 = 0; s0 < a.NP; s0 += G) {
                    const int s = s0 + lane;
                    float gs = 0.0f;
                    uint32_t it_l = 0u;
                    if (s < a.NP) {
                        gs = a.gk[kb + s];
                        it_l = a.uit[qb + s];
                        a.gsn[(size_t)(q - a.begin) * a.NP + s] = gs;
                    }
                    unsigned long long live = slk_group_or<G>(gs != 0.0f ? 1ull << lane : 0ull);
                    while (live) {
                        const int j0 = __builtin_ctzll(live);
                        live &= live - 1ull;
                        const bool two = live != 0ull;
                        const int j1 = two ? __builtin_ctzll(live) : j0;
                        if (two) live &= live - 1ull;
                        const uint32_t i0 = __shfl(it_l, j0, G), i1 = __shfl(it_l, j1, G);
                        const float g0 = __shfl(gs, j0, G), g1 = __shfl(gs, j1, G);
                        slk_vec<VEC> v0, v1 = slk_vzero<VEC>();
                        if (BLOOM) {
                            v0 = slk_emb_vec<VEC>(a.P[1], a.ib, i0, D, d0, on);
                            if (two) v1 = slk_emb_vec<VEC>(a.P[1], a.ib, i1, D, d0, on);
                        } else {
                            v0 = on ? slk_vload<VEC>(a.P[1] + (size_t)i0 * D + d0) : slk_vzero<VEC>();
                            if (two && on) v1 = slk_vload<VEC>(a.P[1] + (size_t)i1 * D + d0);
                        }
                        slk_vaxpy<VEC>(gu, g0, v0);
                        gbu += g0;
                        if (two) {
                            slk_vaxpy<VEC>(gu, g1, v1);
                            gbu += g1;
                        }
                    }
                }
            }
            ++q;
            if (NEXT_KEY) {
                if (next_key != key) break;
                next_key = (q + 1 < seg_limit) ? a.ukey[q + 1] : ~key;
            } else if (!(q < seg_limit && a.ukey[q] == key)) {
                break;
            }
        } while (true);

        if (ULONG && run_long) {
            // a segment of a long run: its part of the user gradient, for k_user_stitch
            const bool ends = !(q < a.end && a.ukey[q] == key);
            const size_t slot = 2 * (size_t)((p - a.begin) / S) + (is_head ? 1u : 0u);
            float *pp = a.upart + slot * (size_t)a.UPS;
            if (on) slk_vstore<VEC>(pp + d0, gu);
            if (lane == 0) {
                pp[a.UPS - 1] = gbu;
                a.upart_meta[2 * slot] = key;
                a.upart_meta[2 * slot + 1] = (a.upart_gen << SLK_IPART_GEN_SHIFT) | (is_head ? (uint32_t)SLK_IPART_STARTS : 0u) |
                                             (ends ? (uint32_t)SLK_IPART_ENDS : 0u);
                if (is_head) atomicAdd(a.upart_count, 1u);
            }
            p = q - 1u;  // (a run that ends inside the tile: the walk goes on with the run behind it)
            continue;
        }
        if (BLOOM && a.ub.n_hash) {
            if (on) slk_vstore<VEC>(a.urec + (size_t)(p - a.begin) * a.RSU + d0, gu);
        } else if (on) {
            if (EARLY_ADAM)
                slk_apply_vec_pre<VEC, UPD, true>(a, 0, uoff, u, su, gu, &su2, (SLK_NT_OF(a) & 1) != 0, unew);
            else if (EARLY_STATE)
                slk_apply_vec_pre<VEC, UPD>(a, 0, uoff, u, su, gu, nullptr, (SLK_NT_OF(a) & 1) != 0, unew);
            else
                slk_apply_vec<VEC, UPD>(a, 0, uoff, u, gu, (SLK_NT_OF(a) & 1) != 0, unew);
        }
        if (pp && lane == 0) a.uflag[user] = (uint8_t)(cur ^ 1u);  // (this group owns the user: nobody else reads or writes the byte in this launch)
        if (EARLY_ADAM) {
            if (lane == 0) {  // (SparseAdam decays the moments of a looked-up row whatever its gradient is)
                slk_vec<1> bpv, bsv, bsv2, gbv;
                bpv.v[0] = bu;
                bsv.v[0] = sbu;
                bsv2.v[0] = sbu2;
                gbv.v[0] = gbu;
                slk_apply_vec_pre<1, UPD, true>(a, 2, user, bpv, bsv, gbv, &bsv2);
            }
        } else if (EARLY_STATE) {
            if (lane == 0 && gbu != 0.0f) {  // gbu == 0 is an exact no-op for Adagrad (slk_apply_bias)
                slk_vec<1> bpv, bsv, gbv;
                bpv.v[0] = bu;
                bsv.v[0] = sbu;
                gbv.v[0] = gbu;
                slk_apply_vec_pre<1, UPD>(a, 2, user, bpv, bsv, gbv);
            }
        } else if (lane == 0) {
            slk_apply_bias<UPD>(a, 2, user, gbu);
        }
        if (ULONG) p = q - 1u;  // the run's other positions (in this tile; beyond it the loop ends) are done
    }
    }
    if (!PRE || EXPL) {
        const double tot = slk_block_sum_256((double)loss_acc, red);
        if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = tot;
    }
}

// Behind k_user_pass<..., ULONG>: one row group per tile of SLK_USER_TILE positions.  The group of the tile in which a long
// run STARTS adds the run's partials in tile order -- its own tile's slot 1, then slot 0 of the following tiles up to the
// one in which the run ends (the next G tiles' metas are read at once, the lanes that still belong to the run are a
// prefix) -- and applies the one update of U[u] and its bias.
template <int VEC, int G, int UPD, bool PP = false>
__global__ __launch_bounds__(256) void k_user_stitch(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    constexpr uint32_t S = SLK_USER_TILE;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t ntiles = (a.end - a.begin + S - 1u) / S;
    if (*a.upart_count == 0u) return;
    const uint32_t gen = a.upart_gen;
    for (uint32_t tile = blockIdx.x * GPB + grp; tile < ntiles; tile += gridDim.x * GPB) {
        const size_t slot = 2 * (size_t)tile + 1;
        const uint32_t fl = a.upart_meta[2 * slot + 1];
        if ((fl >> SLK_IPART_GEN_SHIFT) != gen) continue;  // no long run starts in this tile
        const uint32_t key = a.upart_meta[2 * slot];
        const float *pp = a.upart + slot * (size_t)a.UPS;
        slk_vec<VEC> gu = on ? slk_vload<VEC>(pp + d0) : slk_vzero<VEC>();
        float gbu = pp[a.UPS - 1];
        uint32_t t2 = tile + 1;
        bool more = (fl & SLK_IPART_ENDS) == 0;
        // A round looks at the next TPR = R * G tiles: lane l reads the metas of tiles t2 + l + j * G ({key, flags}: one 8-B load;
        // the tiles that continue this run are a prefix), and the partials the round can consume are fetched WITH the metas --
        // their addresses do not depend on them -- so a run that fills thousands of tiles costs its owner one round trip per TPR
        // tiles (measured before: one per tile, then three per G tiles; profiles/r03_*).  What the run does not reach is
        // dropped.  The sums are added in tile order, as the persistent kernel's walk adds them.
        constexpr int R = G <= 16 ? 2 : 1;
        constexpr int TPR = R * G;
        constexpr bool SPEC = G <= 16;  // wider groups: partials fetched after the metas, SLK_STITCH_BATCH at a time
        while (more) {
            // every load of the round is issued before anything waits: the partials first, then the metas (the compiler keeps
            // program order, and a compare right behind a meta load would put a full round trip in front of the partial loads)
            slk_vec<VEC> sc[SPEC ? TPR : 1];
            float sb[SPEC ? TPR : 1];
            if (SPEC) {
#pragma unroll
                for (int e = 0; e < (SPEC ? TPR : 1); ++e) {
                    sc[e] = slk_vzero<VEC>();
                    sb[e] = 0.0f;
                    if (t2 + (uint32_t)e < ntiles) {
                        const float *q = a.upart + 2 * (size_t)(t2 + (uint32_t)e) * (size_t)a.UPS;
                        if (on) sc[e] = slk_vload<VEC>(q + d0);
                        sb[e] = q[a.UPS - 1];
                    }
                }
            }
            uint2 kf[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint32_t tl = t2 + (uint32_t)(lane + j * G);
                kf[j] = make_uint2(0u, 0u);
                if (tl < ntiles) kf[j] = *reinterpret_cast<const uint2 *>(a.upart_meta + 4 * (size_t)tl);
            }
            uint32_t f[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                f[j] = kf[j].y;
                if (t2 + (uint32_t)(lane + j * G) >= ntiles || (f[j] >> SLK_IPART_GEN_SHIFT) != gen || (f[j] & SLK_IPART_STARTS) ||
                    kf[j].x != key)
                    f[j] = 0u;
                else
                    f[j] |= 1u;  // bit 0: part of this run
            }
            // cnt = tiles of this round that belong to the run: up to and including the first one that ends it
            // (the lanes' flags as bit masks of the group, OR-reduced in log2(G) exchanges: a loop of G dependent
            // lane-to-lane reads per round was what the stitch of a run over thousands of tiles spent its time in)
            int cnt = 0;
            bool ended = false;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (!ended && cnt == j * G) {
                    const unsigned long long valid = slk_group_or<G>((f[j] & 1u) ? 1ull << lane : 0ull);
                    const unsigned long long ends = slk_group_or<G>((f[j] & SLK_IPART_ENDS) ? 1ull << lane : 0ull);
                    const unsigned long long full = G == 64 ? ~0ull : (1ull << G) - 1ull;
                    int c = (valid & full) == full ? G : __builtin_ctzll(~valid);      // the leading lanes that continue the run
                    const unsigned long long e_in = ends & (c == 64 ? ~0ull : (1ull << c) - 1ull);
                    if (e_in) {
                        c = __builtin_ctzll(e_in) + 1;                                 // ... up to and including the one that ends it
                        ended = true;
                    }
                    cnt += c;
                }
            }
            if (SPEC) {
#pragma unroll
                for (int e = 0; e < (SPEC ? TPR : 1); ++e) {
                    if (e < cnt) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gu.v[i] += sc[e].v[i];
                        gbu += sb[e];
                    }
                }
            }
            for (int l0 = 0; !SPEC && l0 < cnt; l0 += SLK_STITCH_BATCH) {
                slk_vec<VEC> cc[SLK_STITCH_BATCH];
                float cb[SLK_STITCH_BATCH];
#pragma unroll
                for (int e = 0; e < SLK_STITCH_BATCH; ++e) {
                    cc[e] = slk_vzero<VEC>();
                    cb[e] = 0.0f;
                    if (l0 + e < cnt) {
                        const float *q = a.upart + 2 * (size_t)(t2 + (uint32_t)(l0 + e)) * (size_t)a.UPS;
                        if (on) cc[e] = slk_vload<VEC>(q + d0);
                        cb[e] = q[a.UPS - 1];
                    }
                }
#pragma unroll
                for (int e = 0; e < SLK_STITCH_BATCH; ++e) {
                    if (l0 + e < cnt) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gu.v[i] += cc[e].v[i];
                        gbu += cb[e];
                    }
                }
            }
            more = !ended && cnt == TPR;
            t2 += (uint32_t)TPR;
        }
        const uint32_t user = key & a.umask;
        const size_t uoff = (size_t)user * D + d0;
        // (user-row ping-pong: the long run's segments read the row from the current copy and left the flag alone)
        constexpr bool pingpong = PP;
        // (read by lane 0 and handed round: nothing else separates the other lanes' read from lane 0's write below)
        uint32_t cur = 0u;
        if (pingpong) cur = __shfl(lane == 0 ? (uint32_t)a.uflag[user] : 0u, 0, G);
        if (on) {
            slk_vec<VEC> u = slk_vload_if_nt<VEC>((cur ? a.P0alt : a.P[0]) + uoff, (SLK_NT_OF(a) & 1) != 0);
            slk_apply_vec<VEC, UPD>(a, 0, uoff, u, gu, (SLK_NT_OF(a) & 1) != 0, pingpong ? (cur ? a.P[0] : a.P0alt) + uoff : nullptr);
        }
        if (pingpong && lane == 0) a.uflag[user] = (uint8_t)(cur ^ 1u);
        if (lane == 0) slk_apply_bias<UPD>(a, 2, user, gbu);
    }
}

// flags[mb] |= 1 iff some user run of minibatch mb's window of the user-sorted list wholly covers an aligned tile of
// SLK_USER_TILE positions (ids only: once per chunk, read back by the host with the item pass's flags)
__global__ __launch_bounds__(256) void k_user_long_flags(const uint32_t *ukey, uint32_t n, uint32_t per_mb, int *flags) {
    constexpr uint32_t S = SLK_USER_TILE;
    const uint32_t tiles_per_mb = (per_mb + S - 1) / S;
    const uint32_t n_mb = (n + per_mb - 1) / per_mb;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_mb * tiles_per_mb; i += gridDim.x * 256) {
        const uint32_t mb = i / tiles_per_mb, q = i - mb * tiles_per_mb;
        const uint32_t w0 = mb * per_mb, w1 = (n - w0 < per_mb) ? n : w0 + per_mb;
        const uint32_t t0 = w0 + q * S;
        if (t0 >= w1 || w1 - t0 < S) continue;  // only full tiles make a run long
        if (ukey[t0] == ukey[t0 + S - 1]) flags[mb] = 1;
    }
}

// ---------------------------------------------------------------------------------------
// adaptive hinge: scores for every (interaction, pair), then the per-column selection of
// _get_multiple_negative_predictions' view(n, B) layout (factorization/implicit.py:266-275)
// ---------------------------------------------------------------------------------------
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_score_pass(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t stride = gridDim.x * GPB;
    for (uint32_t q = a.begin + blockIdx.x * GPB + grp; q < a.end; q += stride) {
        const uint32_t user = a.ukey[q] & a.umask;
        const slk_vec<VEC> u = slk_emb_vec<VEC>(a.P[0], a.ub, user, D, d0, on);
        const float bu = a.P[2][user];
        const size_t kb = (size_t)a.uk[q] * a.NP, qb = (size_t)q * a.NP;
        constexpr int SB = 3;  // candidates whose rows are in flight together (1 + 5 draws: two batches)
        for (int s0 = 0; s0 < a.NP; s0 += SB) {
            slk_vec<VEC> v[SB];
            float bi[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                v[j] = slk_vzero<VEC>();
                bi[j] = 0.0f;
                if (s0 + j < a.NP) {
                    const uint32_t it = a.uit[qb + s0 + j];
                    v[j] = slk_emb_vec<VEC>(a.P[1], a.ib, it, D, d0, on);
                    bi[j] = a.P[3][SLK_B3(a, it)];
                }
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const float sc = slk_group_sum<G>(slk_vdot<VEC>(u, v[j])) + bu + bi[j];
                if (lane == 0 && s0 + j < a.NP) a.sk[kb + s0 + j] = sc;
            }
        }
    }
}

// (k_adaptive_select: slk_kernels.h -- the row-sharded path launches it too)

// ---------------------------------------------------------------------------------------
// explicit feedback (spotlight/factorization/explicit.py:223-234, losses.py:169-244): per interaction
// the loss of its ONE predicted score against the observed rating and dL/dscore, formed in fp32
// operation by operation as autograd forms them.  One thread per interaction of the minibatch.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_explicit_loss(const float *sk, const float *ratings, float *gk, uint32_t k0,
                                                       uint32_t bm, int loss_kind, float inv_b, double *loss_partial) {
    __shared__ double red[256];
    double lsum = 0.0;
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < bm; c += gridDim.x * 256) {
        const float sc = sk[k0 + c], r = ratings[k0 + c];
        float l, g;
        slk_explicit_loss(loss_kind, sc, r, inv_b, bm, l, g);
        gk[k0 + c] = g;
        lsum += (double)l;
    }
    const double tot = slk_block_sum_256(lsum, red);
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------
// dense sweeps (reference default optimizer: Adam with weight_decay = l2 over EVERY row)
// ---------------------------------------------------------------------------------------

// One launch sweeps up to four tables: block b belongs to the table t with first[t] <= b < first[t + 1]
// and strides over that table with the blocks of its own range (a minibatch of the reference's default
// optimizer is then user pass + item pass + ONE sweep launch instead of four).
struct slk_sweep_all_args {
    float *p[4], *s1[4], *s2[4], *g[4];
    size_t numel[4];
    unsigned first[5];
    int adam;
    float wd, w1, beta2, omb2, step_size, bc2_sqrt, eps, clr;
};

__global__ __launch_bounds__(256) void k_dense_sweep_all(slk_sweep_all_args a) {
    int t = 0;
    while (t < 3 && blockIdx.x >= a.first[t + 1]) ++t;
    const size_t local = blockIdx.x - a.first[t], nblk = a.first[t + 1] - a.first[t];
    float *p = a.p[t], *s1 = a.s1[t], *s2 = a.s2[t], *g = a.g[t];
    const size_t numel = a.numel[t];
    if (a.adam) {
        // torch/optim/adam.py:414-546 (single-tensor): g += wd*p; m.lerp_(g, 1-b1);
        // v = b2*v + (1-b2) g^2; p += -(lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
        for (size_t e = local * 256 + threadIdx.x; e < numel; e += nblk * 256) {
            const float gv = g[e] + a.wd * p[e];
            g[e] = 0.0f;
            const float m = s1[e] + a.w1 * (gv - s1[e]);
            const float v = s2[e] * a.beta2 + a.omb2 * (gv * gv);
            s1[e] = m;
            s2[e] = v;
            p[e] += -a.step_size * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
        }
    } else {
        // torch/optim/adagrad.py:350-385 dense branch with weight_decay
        for (size_t e = local * 256 + threadIdx.x; e < numel; e += nblk * 256) {
            const float gv = g[e] + a.wd * p[e];
            g[e] = 0.0f;
            const float s = s1[e] + gv * gv;
            s1[e] = s;
            p[e] += -a.clr * (gv / (sqrtf(s) + a.eps));
        }
    }
}

// ---------------------------------------------------------------------------------------
// prep kernels
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_i64_to_u32(const int64_t *in, uint32_t *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (uint32_t)in[i];
}

__global__ __launch_bounds__(256) void k_u32_to_i64(const uint32_t *in, int64_t *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (int64_t)in[i];
}

__global__ __launch_bounds__(256) void k_pack_items(const uint32_t *uk, const int64_t *items,
                                                    const uint32_t *neg32, uint32_t nc, int nn, uint32_t *uit) {
    const int NP = nn + 1;
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nc; q += gridDim.x * 256) {
        const uint32_t k = uk[q];
        uit[(size_t)q * NP] = (uint32_t)items[k];
        for (int r = 0; r < nn; ++r) uit[(size_t)q * NP + 1 + r] = neg32[(size_t)k * nn + r];
    }
}

// adaptive hinge, live occurrences only: qk[uk[q]] = q
__global__ __launch_bounds__(256) void k_invert_perm(const uint32_t *uk, uint32_t nc, uint32_t *qk) {
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nc; q += gridDim.x * 256) qk[uk[q]] = q;
}

// keys of one minibatch's live list (k_adaptive_select): entry e -> (item of occurrence live[e], live[e]);
// with a BloomEmbedding item layer (ib.n_hash > 0, hashed = true) entry e*H + h -> (row_h(item), live[e]).
// Dead entries get the all-ones key: they sort to the end and the item pass skips them.
__global__ __launch_bounds__(256) void k_build_live_keys(const uint32_t *live, const uint32_t *uit, uint32_t n_live,
                                                         slk_bloom_dev ib, bool hashed, uint32_t *key, uint32_t *val) {
    const uint32_t H = hashed ? (uint32_t)ib.n_hash : 1u;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n_live * H; e += gridDim.x * 256) {
        const uint32_t l = e / H, h = e - l * H;
        const uint32_t r = live[l];
        uint32_t k = 0xffffffffu;
        if (r != 0xffffffffu) k = hashed ? slk_bloom_row(ib, uit[r], (int)h) : uit[r];
        key[e] = k;
        val[e] = r;
    }
}

// BloomEmbedding item layer: occurrence r contributes to the n_hash hashed rows of its item:
// key[r*H + h] = (minibatch, row_h(item)), value = r (same record as the plain occurrence)
__global__ __launch_bounds__(256) void k_build_item_bloom_keys(const uint32_t *uit, uint32_t nocc, int NP,
                                                               uint32_t bsz, unsigned cbits, slk_bloom_dev ib,
                                                               uint32_t *key, uint32_t *val) {
    const uint32_t H = (uint32_t)ib.n_hash;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nocc * H; e += gridDim.x * 256) {
        const uint32_t r = e / H, h = e - r * H;
        const uint32_t q = r / (uint32_t)NP;
        key[e] = ((q / bsz) << cbits) | slk_bloom_row(ib, uit[r], (int)h);
        val[e] = r;
    }
}

// BloomEmbedding user layer: the head position p of every user segment contributes its summed
// gradient record to the n_hash hashed rows of the user; non-head positions get the sentinel
// row `ub.rows` (never updated).  value = p.
__global__ __launch_bounds__(256) void k_build_user_bloom_keys(const uint32_t *ukey, uint32_t umask, uint32_t nc,
                                                               uint32_t bsz, unsigned cbits, slk_bloom_dev ub,
                                                               uint32_t *key, uint32_t *val) {
    const uint32_t H = (uint32_t)ub.n_hash;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nc * H; e += gridDim.x * 256) {
        const uint32_t p = e / H, h = e - p * H;
        const uint32_t mb = p / bsz;
        const bool head = (p % bsz == 0) || ukey[p - 1] != ukey[p];
        key[e] = (mb << cbits) | (head ? slk_bloom_row(ub, ukey[p] & umask, (int)h) : ub.rows);
        val[e] = p;
    }
}

// ---------------------------------------------------------------------------------------
// predict
// ---------------------------------------------------------------------------------------
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_predict(const float *U, const float *V, const float *bu,
                                                 const float *bi, int D, slk_bloom_dev ub, slk_bloom_dev ib,
                                                 const int64_t *users, int64_t n_users, const int64_t *items,
                                                 int64_t n, float *out) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t k = (int64_t)blockIdx.x * GPB + grp; k < n; k += (int64_t)gridDim.x * GPB) {
        const int64_t u = users[n_users == 1 ? 0 : k];
        const int64_t i = items ? items[k] : k;
        const slk_vec<VEC> a = slk_emb_vec<VEC>(U, ub, (uint32_t)u, D, d0, on);
        const slk_vec<VEC> b = slk_emb_vec<VEC>(V, ib, (uint32_t)i, D, d0, on);
        const float s = (slk_chain_dot<VEC, G>(a, b) + bu[u]) + bi[i];  // the score's definition: slk_kernels.h, slk_eval.hip
        if (lane == 0) out[k] = s;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
typedef slk_pass_fn pass_fn;

template <int VEC, int G, bool BLOOM>
static pass_fn user_pass_fn2(int upd, int umode) {
    if (umode == 2) {
        if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 2, BLOOM>;
        if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, 2, BLOOM>;
        if (upd == SLK_UPD_SGD) return k_user_pass<VEC, G, SLK_UPD_SGD, 2, BLOOM>;
        return k_user_pass<VEC, G, SLK_UPD_GRAD_ONLY, 2, BLOOM>;
    }
    if (umode == 1) {
        if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 1, BLOOM>;
        if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, 1, BLOOM>;
        if (upd == SLK_UPD_SGD) return k_user_pass<VEC, G, SLK_UPD_SGD, 1, BLOOM>;
        return k_user_pass<VEC, G, SLK_UPD_GRAD_ONLY, 1, BLOOM>;
    }
    if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 0, BLOOM>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, 0, BLOOM>;
    if (upd == SLK_UPD_SGD) return k_user_pass<VEC, G, SLK_UPD_SGD, 0, BLOOM>;
    return k_user_pass<VEC, G, SLK_UPD_GRAD_ONLY, 0, BLOOM>;
}

// the latency-bound form of the pair-mode user pass over plain tables (k_user_pass<..., LAT = true>)
template <int VEC, int G, bool ULONG = false>
static pass_fn user_pass_lat_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 0, false, true, ULONG>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, 0, false, true, ULONG>;
    if (upd == SLK_UPD_SGD) return k_user_pass<VEC, G, SLK_UPD_SGD, 0, false, true, ULONG>;
    return k_user_pass<VEC, G, SLK_UPD_GRAD_ONLY, 0, false, true, ULONG>;
}

// inside a user-row ping-pong scope (pair mode, plain tables, row-sparse optimizers): the four forms of the pass and the stitch
template <int VEC, int G, bool LAT, bool ULONG>
static pass_fn user_pass_pp_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 0, false, LAT, ULONG, true>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, 0, false, LAT, ULONG, true>;
    return k_user_pass<VEC, G, SLK_UPD_SGD, 0, false, LAT, ULONG, true>;
}
// ... with the single-occurrence fast path (Adagrad, the bandwidth-bound form) and the item pass that skips what it covered
template <int VEC, int G>
static pass_fn user_pass_sgl_fn() {
    return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, 0, false, false, false, true, true>;
}
template <int VEC, int G>
static slk_item_fns item_pass_sgl_fn() {
    return {k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPPS>, k_item_stitch<VEC, G, SLK_UPD_ADAGRAD, SLK_PART_BOTH>,
            k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPPS, SLK_PART_BOTH, false>,
            k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPPS, SLK_PART_BOTH, false, 4>};
}
template <int VEC, int G>
static pass_fn user_stitch_pp_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_user_stitch<VEC, G, SLK_UPD_ADAGRAD, true>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_stitch<VEC, G, SLK_UPD_SPARSE_ADAM, true>;
    return k_user_stitch<VEC, G, SLK_UPD_SGD, true>;
}
template <int VEC, int G>
static slk_item_fns item_pass_pp_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD)
        return {k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPP>, k_item_stitch<VEC, G, SLK_UPD_ADAGRAD, SLK_PART_BOTH>,
                k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false>,
                k_item_pass<VEC, G, SLK_UPD_ADAGRAD, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false, 4>};
    if (upd == SLK_UPD_SPARSE_ADAM)
        return {k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, SLK_ITEM_SNAPPP>, k_item_stitch<VEC, G, SLK_UPD_SPARSE_ADAM, SLK_PART_BOTH>,
                k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false>,
                k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false, 4>};
    return {k_item_pass<VEC, G, SLK_UPD_SGD, SLK_ITEM_SNAPPP>, k_item_stitch<VEC, G, SLK_UPD_SGD, SLK_PART_BOTH>,
            k_item_pass<VEC, G, SLK_UPD_SGD, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false>,
            k_item_pass<VEC, G, SLK_UPD_SGD, SLK_ITEM_SNAPPP, SLK_PART_BOTH, false, 4>};
}

// plain tables, minibatches that hold a long user run (k_user_pass<..., ULONG = true>) and the stitch kernel behind it
template <int VEC, int G, int UMODE>
static pass_fn user_pass_long_fn1(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_user_pass<VEC, G, SLK_UPD_ADAGRAD, UMODE, false, false, true>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM, UMODE, false, false, true>;
    if (upd == SLK_UPD_SGD) return k_user_pass<VEC, G, SLK_UPD_SGD, UMODE, false, false, true>;
    return k_user_pass<VEC, G, SLK_UPD_GRAD_ONLY, UMODE, false, false, true>;
}
template <int VEC, int G>
static pass_fn user_pass_long_fn(int upd, int umode) {
    if (umode == 2) return user_pass_long_fn1<VEC, G, 2>(upd);
    if (umode == 1) return user_pass_long_fn1<VEC, G, 1>(upd);
    return user_pass_long_fn1<VEC, G, 0>(upd);
}
template <int VEC, int G>
static pass_fn user_stitch_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_user_stitch<VEC, G, SLK_UPD_ADAGRAD>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_user_stitch<VEC, G, SLK_UPD_SPARSE_ADAM>;
    if (upd == SLK_UPD_SGD) return k_user_stitch<VEC, G, SLK_UPD_SGD>;
    return k_user_stitch<VEC, G, SLK_UPD_GRAD_ONLY>;
}

template <int VEC, int G>
static pass_fn user_pass_fn(int upd, int umode, bool bloom) {
    return bloom ? user_pass_fn2<VEC, G, true>(upd, umode) : user_pass_fn2<VEC, G, false>(upd, umode);
}


int slk_check_tables(slk_ctx *ctx, const slk_tables *t, unsigned table_mask, int *vec, int *g, bool shadow_ok, bool pingpong_ok) {
    if (!t) return slk_fail(ctx, SLK_EINVAL, "tables is NULL");
    for (int i = 0; i < 4; ++i)
        if (((table_mask >> i) & 1u) && !t->d_param[i])
            return slk_fail(ctx, SLK_EINVAL, "tables->d_param[%d] is NULL", i);
    if (!shadow_ok && (table_mask & 8u) && ctx->shadow_active && t->d_param[3] == ctx->shadow_src_p)
        return slk_fail(ctx, SLK_EINVAL, "the item biases of these tables are shadowed (slk_bias_shadow_begin): the array is stale until "
                                         "slk_bias_shadow_end, and only slk_bilinear_train indexes the shadow");
    if (!pingpong_ok && (table_mask & 1u) && ctx->pp_active && t->d_param[0] == ctx->pp_src_u)
        return slk_fail(ctx, SLK_EINVAL, "the user rows of these tables are ping-ponged (slk_user_pingpong_begin): the array holds only "
                                         "some of the current rows until slk_user_pingpong_end, and only slk_bilinear_train reads both copies");
    if (((table_mask & 5u) && (t->num_users < 1 || t->num_users >= ((int64_t)1 << 31))) || t->num_items < 1 ||
        t->num_items >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "table rows must be in [1, 2^31): users %lld items %lld",
                        (long long)t->num_users, (long long)t->num_items);
    if (!slk_pick_layout(t->dim, vec, g))
        return slk_fail(ctx, SLK_EINVAL,
                        "embedding dim %d unsupported (need dim %% 4 == 0 and <= 256, or dim <= 64)", t->dim);
    const slk_bloom *bl[2] = {(table_mask & 1u) ? t->user_bloom : nullptr, (table_mask & 2u) ? t->item_bloom : nullptr};
    for (int side = 0; side < 2; ++side) {
        const slk_bloom *b = bl[side];
        if (!b) continue;
        if (b->n_hash < 1 || b->n_hash > 8 || b->rows < 1 || b->rows >= ((int64_t)1 << 30) || b->skip_row >= b->rows ||
            b->skip_row < -1 || b->padding_idx < -1)
            return slk_fail(ctx, SLK_EINVAL, "bad BloomEmbedding descriptor: rows %lld n_hash %d skip_row %lld",
                            (long long)b->rows, b->n_hash, (long long)b->skip_row);
    }
    return SLK_OK;
}

int slk_check_optim(slk_ctx *ctx, const slk_optim *optim, unsigned table_mask) {
    if (!optim) return slk_fail(ctx, SLK_EINVAL, "optim is NULL");
    if (optim->kind < SLK_OPT_ADAGRAD || optim->kind > SLK_OPT_SGD)
        return slk_fail(ctx, SLK_EINVAL, "unknown optimizer kind %d", optim->kind);
    if (optim->kind == SLK_OPT_SGD) {  // stateless
        if (optim->weight_decay != 0.0) return slk_fail(ctx, SLK_EINVAL, "SGD: weight_decay must be 0 (a full-table sweep per step is not built)");
        return SLK_OK;
    }
    const bool need_s2 = optim->kind == SLK_OPT_SPARSE_ADAM || optim->kind == SLK_OPT_ADAM_DENSE;
    for (int i = 0; i < 4; ++i) {
        if (!((table_mask >> i) & 1u)) continue;
        if (!optim->d_state1[i]) return slk_fail(ctx, SLK_EINVAL, "optim->d_state1[%d] is NULL", i);
        if (need_s2 && !optim->d_state2[i]) return slk_fail(ctx, SLK_EINVAL, "optim->d_state2[%d] is NULL", i);
    }
    if (optim->kind == SLK_OPT_ADAGRAD && optim->weight_decay != 0.0)
        return slk_fail(ctx, SLK_EINVAL, "row-sparse Adagrad requires weight_decay == 0 (use ADAGRAD_DENSE)");
    return SLK_OK;
}

int slk_ensure_dgrad(slk_ctx *ctx, const size_t elems[4], unsigned table_mask, hipStream_t s) {
    for (int t = 0; t < 4; ++t) {
        if (!((table_mask >> t) & 1u)) continue;
        if (ctx->dgrad_elems[t] != elems[t] || !ctx->dgrad[t].p) {
            int rc = slk_ensure(ctx, ctx->dgrad[t], elems[t] * 4);
            if (rc) return rc;
            SLK_HIP(ctx, hipMemsetAsync(ctx->dgrad[t].p, 0, elems[t] * 4, s));
            ctx->dgrad_elems[t] = elems[t];
        }
    }
    return SLK_OK;
}

// Reference default optimizer (Adam + l2) / dense Adagrad with weight decay: sweep every row of
// the tables in `table_mask`, consuming (and re-zeroing) the dense gradient buffers.
int slk_dense_sweeps(slk_ctx *ctx, float *const params[4], const slk_optim *optim, unsigned table_mask,
                     hipStream_t s) {
    const double step = (double)(optim->step + 1);
    slk_prof_begin(ctx, SLK_K_DENSE_SWEEP, s);
    slk_sweep_all_args w;
    memset(&w, 0, sizeof(w));
    w.wd = (float)optim->weight_decay;
    w.eps = (float)optim->eps;
    w.adam = optim->kind == SLK_OPT_ADAM_DENSE;
    if (w.adam) {
        const double bc1 = 1.0 - pow(optim->beta1, step), bc2 = 1.0 - pow(optim->beta2, step);
        w.w1 = (float)(1.0 - optim->beta1);
        w.beta2 = (float)optim->beta2;
        w.omb2 = (float)(1.0 - optim->beta2);
        w.step_size = (float)(optim->lr / bc1);
        w.bc2_sqrt = (float)sqrt(bc2);
    } else {
        w.clr = (float)(optim->lr / (1.0 + (step - 1.0) * optim->lr_decay));
    }
    unsigned blocks = 0;
    for (int t = 0; t < 4; ++t) {
        w.first[t] = blocks;
        if (!((table_mask >> t) & 1u)) continue;
        w.p[t] = params[t];
        w.s1[t] = optim->d_state1[t];
        w.s2[t] = optim->d_state2[t];
        w.g[t] = (float *)ctx->dgrad[t].p;
        w.numel[t] = ctx->dgrad_elems[t];
        blocks += slk_grid_for(ctx, w.numel[t], 256);
    }
    w.first[4] = blocks;
    if (blocks) {
        hipLaunchKernelGGL(k_dense_sweep_all, dim3(blocks), dim3(256), 0, s, w);
        SLK_LAUNCH_CHECK(ctx, "k_dense_sweep_all");
    }
    slk_prof_end(ctx, s);
    return SLK_OK;
}

int slk_launch_i64_to_u32(slk_ctx *ctx, const int64_t *in, uint32_t *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_i64_to_u32, dim3(slk_grid_for(ctx, n, 256)), dim3(256), 0, s, in, out, n);
    SLK_LAUNCH_CHECK(ctx, "k_i64_to_u32");
    return SLK_OK;
}

SLK_EXPORT int slk_bilinear_predict(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_users,
                                    int64_t n_users, const int64_t *d_items, int64_t n, float *d_out,
                                    void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 15u, &vec, &g))) return rc;
    if (n < 0 || !d_users || (n > 0 && !d_out)) return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_predict: bad arguments");
    if (n_users != 1 && n_users != n)
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_predict: n_users must be 1 or n (%lld vs %lld)",
                        (long long)n_users, (long long)n);
    if (n == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    if (n_users == 1 && !d_items) {
        // one user against EVERY item: the item table streams through the matrix cores (slk_eval.hip), same scores bit for bit
        if (n != tables->num_items)
            return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_predict: d_items == NULL scores all %lld items, n is %lld",
                            (long long)tables->num_items, (long long)n);
        const float *rep, *rbias;
        const int64_t *gmap;
        if ((rc = slk_eval_user_rep(ctx, tables, vec, g, d_users, &rep, &rbias, &gmap, s))) return rc;
        rc = slk_eval_predict_all(ctx, tables, rep, rbias, gmap, d_out, s);
        slk_prof_end(ctx, s);
        return rc;
    }
#define SLK_PREDICT(V_, G_)                                                                          \
    hipLaunchKernelGGL((k_predict<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n, 256 / G_)), dim3(256), 0, s, \
                       (const float *)tables->d_param[0], (const float *)tables->d_param[1],          \
                       (const float *)tables->d_param[2], (const float *)tables->d_param[3],          \
                       (int)tables->dim, ubd, ibd, d_users, n_users, d_items, n, d_out)
    slk_bloom_dev ubd, ibd;
    slk_bloom_to_dev(tables->user_bloom, &ubd);
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    SLK_FOR_LAYOUT(vec, g, SLK_PREDICT);
#undef SLK_PREDICT
    SLK_LAUNCH_CHECK(ctx, "k_predict");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

static int bilinear_train_impl(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                               const int64_t *d_users, const int64_t *d_items, int64_t n,
                               int64_t batch_size, int32_t loss, int32_t n_neg, const int64_t *d_neg_in,
                               int64_t *d_neg_out, float *d_mb_loss, void *stream, bool reserve_only,
                               const float *d_ratings = nullptr, bool prefetch_only = false);

SLK_EXPORT int slk_bilinear_train(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                                  const int64_t *d_users, const int64_t *d_items, int64_t n,
                                  int64_t batch_size, int32_t loss, int32_t n_neg, const int64_t *d_neg_in,
                                  int64_t *d_neg_out, float *d_mb_loss, void *stream) {
    return bilinear_train_impl(ctx, tables, optim, d_users, d_items, n, batch_size, loss, n_neg, d_neg_in,
                               d_neg_out, d_mb_loss, stream, false);
}

SLK_EXPORT int slk_bilinear_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t n,
                                    int64_t batch_size, int32_t loss, int32_t n_neg, void *stream) {
    slk_optim o;
    if (!optim) return slk_fail(ctx, SLK_EINVAL, "optim is NULL");
    o = *optim;
    return bilinear_train_impl(ctx, tables, &o, nullptr, nullptr, n, batch_size, loss, n_neg, nullptr, nullptr,
                               nullptr, stream, true);
}

// The FIRST chunk of the next slk_bilinear_train call with these very arguments, prepared now: its negatives and sorts go to
// the ctx's prep stream, beside whatever the caller's stream still holds (the passes of the epoch before).  fit() calls it as
// soon as the next epoch's shuffled ids exist; h_key / pos (optional): the MT19937 state the call's draws start from, written
// without waiting for the ctx's stream (the caller has waited for the last draw: slk_rng_get_state_sampled).  A no-op for
// calls that would not pipeline their prep (a single chunk, minibatches below "overlap_min_batch", the persistent route).
SLK_EXPORT int slk_bilinear_prefetch(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, const int64_t *d_users,
                                     const int64_t *d_items, int64_t n, int64_t batch_size, int32_t loss, int32_t n_neg,
                                     const uint32_t *h_key, int32_t pos, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!optim) return slk_fail(ctx, SLK_EINVAL, "optim is NULL");
    if (h_key) {
        if (pos < 0 || pos > 624) return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_prefetch: pos %d outside [0, 624]", pos);
        SLK_HIP(ctx, hipSetDevice(ctx->device));
        if (ctx->sampled_valid && ctx->ev_sampled) SLK_HIP(ctx, hipEventSynchronize(ctx->ev_sampled));
        int rc = slk_rng_write_state(ctx, h_key, pos);
        if (rc) return rc;
    }
    slk_optim o = *optim;
    return bilinear_train_impl(ctx, tables, &o, d_users, d_items, n, batch_size, loss, n_neg, nullptr, nullptr, nullptr, stream,
                               false, nullptr, true);
}

// ExplicitFactorizationModel.fit's minibatch loop (spotlight/factorization/explicit.py:213-236): the same
// passes with ONE pair per interaction, dL/dscore from k_explicit_loss (the PRE route of the user pass)
SLK_EXPORT int slk_bilinear_train_explicit(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                                           const int64_t *d_users, const int64_t *d_items, const float *d_ratings,
                                           int64_t n, int64_t batch_size, int32_t loss, float *d_mb_loss,
                                           void *stream) {
    if (ctx && (loss < SLK_LOSS_REGRESSION || loss > SLK_LOSS_LOGISTIC))
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train_explicit: loss kind %d is not regression/poisson/logistic", loss);
    if (ctx && n > 0 && !d_ratings) return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train_explicit: d_ratings is NULL");
    return bilinear_train_impl(ctx, tables, optim, d_users, d_items, n, batch_size, loss, 0, nullptr, nullptr,
                               d_mb_loss, stream, false, d_ratings);
}

static int bilinear_train_impl(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim,
                               const int64_t *d_users, const int64_t *d_items, int64_t n,
                               int64_t batch_size, int32_t loss, int32_t n_neg, const int64_t *d_neg_in,
                               int64_t *d_neg_out, float *d_mb_loss, void *stream, bool reserve_only,
                               const float *d_ratings, bool prefetch_only) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 15u, &vec, &g, /*shadow_ok=*/true, /*pingpong_ok=*/true))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if (n < 0 || batch_size < 1) return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: n %lld batch_size %lld",
                                                 (long long)n, (long long)batch_size);
    if (loss < SLK_LOSS_POINTWISE || loss > SLK_LOSS_LOGISTIC)
        return slk_fail(ctx, SLK_EINVAL, "unknown loss kind %d", loss);
    const bool adaptive = loss == SLK_LOSS_ADAPTIVE_HINGE;
    const bool expl = loss >= SLK_LOSS_REGRESSION;  // explicit feedback: ratings, no negatives
    if (expl != (d_ratings != nullptr) && !reserve_only)
        return slk_fail(ctx, SLK_EINVAL, "ratings go with (and only with) the regression/poisson/logistic losses");
    const bool pre = adaptive || expl;  // dL/dscore is computed before the user pass
    const int nn = adaptive ? n_neg : (expl ? 0 : 1);
    if (!expl && (nn < 1 || nn > 1024)) return slk_fail(ctx, SLK_EINVAL, "num_negative_samples %d outside [1, 1024]", nn);
    const int NP = nn + 1;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (n == 0) return SLK_OK;
    if (!reserve_only && (!d_users || !d_items || (!d_mb_loss && !prefetch_only)))
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: NULL id/loss pointer");
    if (batch_size * (int64_t)NP >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "batch_size * (1 + negatives) must be < 2^31");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;

    const int D = tables->dim;
    const unsigned ubits = slk_bits_for((uint64_t)tables->num_users - 1);
    const unsigned ibits = slk_bits_for((uint64_t)tables->num_items - 1);
    unsigned idbits = ubits > ibits ? ubits : ibits;
    // BloomEmbedding layers (layers.py:74-244): keys of the row-owner passes are hashed rows
    slk_bloom_dev ubd, ibd;
    slk_bloom_to_dev(tables->user_bloom, &ubd);
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    const bool bloom = ubd.n_hash || ibd.n_hash;
    const int Hu = ubd.n_hash, Hi = ibd.n_hash;
    const unsigned ucbits = Hu ? slk_bits_for((uint64_t)ubd.rows) : 0;  // + the non-head sentinel row
    const unsigned icbits = Hi ? slk_bits_for((uint64_t)ibd.rows - 1) : 0;
    if (ucbits > idbits) idbits = ucbits;
    if (icbits > idbits) idbits = icbits;
    const int64_t occ_mult = (int64_t)NP * (Hi ? Hi : 1) > (int64_t)(Hu ? Hu : 1) ? (int64_t)NP * (Hi ? Hi : 1)
                                                                                 : (int64_t)(Hu ? Hu : 1);
    // minibatches per chunk: keys must fit 32 bits, occurrences must fit 2^31, and scratch
    // stays bounded (~8M interactions).
    const int64_t bsz = batch_size < n ? batch_size : n;
    int64_t mb_per_chunk = (int64_t)1 << (32 - idbits);
    // ~8M interactions per chunk (option "chunk_interactions"): large enough for the sorts and the
    // MT19937 jump-ahead sampler to fill the GPU, small enough to bound the scratch
    const int64_t cap_inter = ctx->opt_chunk_interactions;
    if (mb_per_chunk * bsz > cap_inter) mb_per_chunk = cap_inter / bsz;
    while (mb_per_chunk > 1 && mb_per_chunk * bsz * occ_mult >= ((int64_t)1 << 31)) mb_per_chunk >>= 1;
    if (mb_per_chunk < 1) mb_per_chunk = 1;
    if (bsz * occ_mult >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "batch_size * lookups per interaction must be < 2^31");
    // the persistent route (slk_epoch.hip) keeps per-(minibatch, workgroup) loss partials and counts barriers in 32 bits:
    // at most 2^13 minibatches per launch (16 MB of partials at 256 workgroups)
    const bool maybe_epoch = ctx->opt_epoch_kernel && !ctx->epoch_refused && bsz <= ctx->opt_epoch_max_batch;
    if (maybe_epoch && mb_per_chunk > ((int64_t)1 << 13)) mb_per_chunk = (int64_t)1 << 13;
    const int64_t chunk_cap = mb_per_chunk * bsz;

    // Adaptive hinge: only the positive and the selected negative of a column carry a gradient, so
    // the item-side owner passes run over 2 live occurrences per interaction (sorted per minibatch,
    // after the selection) instead of all 1+n.  Not for SparseAdam, which also decays the moments
    // of looked-up rows whose gradient is zero.
    // Below `adaptive_late_min_batch` interactions per minibatch the per-minibatch sort of the live occurrences (half a dozen
    // small launches) costs more than the dead occurrences it removes from the item pass: all 1+n are sorted once per chunk.
    const bool late = adaptive && optim->kind != SLK_OPT_SPARSE_ADAM && (Hi > 0 || bsz >= ctx->opt_adaptive_late_min_batch);
    // scratch.  With the "overlap_prep" option the value-independent part of a chunk (negatives,
    // sort by user, sort by item) is prepared on a second HIP stream while the previous chunk's
    // passes run, and those buffers exist twice (ctx->pb[0|1]).  What fit() sets for its epochs (profiles/r03_a_*, r03_b_*: with
    // two of the user pass's eight workgroups per CU left to the prep stream the steady state of a run of training calls gains
    // 1.5-4 %; round 1 had measured no gain with the pass holding every wave slot); off on a bare ctx (slk_common.h).
    const size_t nc_max = (size_t)(chunk_cap < n ? chunk_cap : n);
    // chunk boundaries: chunks of chunk_cap.  Chunking is value-neutral: the negatives of a call are one contiguous draw however
    // it is cut.  (Ramped chunks and a short first chunk for the overlapped prep were measured in round 3 and lost at every
    // call length -- small chunks pay the sampler's jump-ahead and the sorts' fixed costs again: profiles/r03_c_*, r03_w_*.)
    std::vector<int64_t> cb;
    cb.push_back(0);
    while (cb.back() < n) {
        const int64_t next = cb.back() + mb_per_chunk * bsz;
        cb.push_back(next < n ? next : n);
    }
    const size_t n_chunks = cb.size() - 1;
    // Two buffer sets whenever a call has more than one chunk: chunk c + 1 is prepared (negatives, sorts) BEFORE chunk c's passes are
    // enqueued.  `side`: that preparation goes to the ctx's second stream and runs BESIDE the passes (option "overlap_prep", what
    // fit() sets).  Otherwise (a bare ctx; round 6) it goes to the caller's own stream, one chunk ahead: still one stream, every
    // kernel alone on the GPU -- but the host's one wait per chunk (the long-run flags of do_sort) now falls while the GPU works on
    // the next chunk's prep instead of idling (a bubble of ~40 us per chunk of 8 minibatches in rounds 1-5).
    const bool side = ctx->opt_overlap_prep && n_chunks > 1 && bsz >= ctx->opt_overlap_min_batch;
    const int nsets = n_chunks > 1 ? 2 : 1;
    // (the single-occurrence flags of a chunk, when the call may take that path: decided below, sized here)
    const bool sgl_maybe = ctx->pp_active && ctx->pp_src_u == tables->d_param[0] && !pre && !bloom && !dense &&
                           optim->kind == SLK_OPT_ADAGRAD && ctx->opt_item_single_min_items > 0 &&
                           tables->num_items >= ctx->opt_item_single_min_items;
    for (int st = 0; st < nsets; ++st) {
        slk_prep_bufs &pb = ctx->pb[st];
        if ((rc = slk_ensure(ctx, pb.neg32, nc_max * nn * 4))) return rc;
        for (int b = 0; b < 2; ++b) {
            if ((rc = slk_ensure(ctx, pb.ukey[b], nc_max * 4))) return rc;
            if ((rc = slk_ensure(ctx, pb.uval[b], nc_max * 8))) return rc;
            // `late`: the item-side lists are built per minibatch from the live occurrences only;
            // ipay[0] then holds the interaction -> sorted-position map of the chunk
            if (!late && (rc = slk_ensure(ctx, pb.ikey[b], nc_max * NP * 4))) return rc;
            if (!(late && b == 1) && (rc = slk_ensure(ctx, pb.ipay[b], late ? nc_max * 4 : nc_max * NP * 4))) return rc;
            if (!late && Hi && (rc = slk_ensure(ctx, pb.bik[b], nc_max * NP * Hi * 4))) return rc;
            if (!late && Hi && (rc = slk_ensure(ctx, pb.bip[b], nc_max * NP * Hi * 4))) return rc;
            if (Hu && (rc = slk_ensure(ctx, pb.buk[b], nc_max * Hu * 4))) return rc;
            if (Hu && (rc = slk_ensure(ctx, pb.bup[b], nc_max * Hu * 4))) return rc;
        }
        if (pre && (rc = slk_ensure(ctx, pb.uit, nc_max * NP * 4))) return rc;
        if (sgl_maybe && (rc = slk_ensure(ctx, pb.mflag, nc_max * NP + 16))) return rc;   // "occurs more than once" per occurrence: by payload,
        if (sgl_maybe && (rc = slk_ensure(ctx, pb.msorted, nc_max * NP + 16))) return rc; //   in item-sorted order
        if ((rc = slk_ensure(ctx, pb.lflags, 2 * (size_t)mb_per_chunk * 4))) return rc;  // long-run flags of a chunk: items, users
    }
    enum { BL_UREC = 16, BL_LIVE, BL_LK0, BL_LK1, BL_LV0, BL_LV1, BL_GSN, BL_UPART, BL_LATE_SORT = 38 };  // ctx->extra slots (24, 25: slk_eval.hip; the user-partial metas, which keep launch stamps across calls, have a buffer of their own: ctx->upart_meta)
    const int RS = (D + 3) / 4 * 4;  // record = the pre-step user row (16-B granular; D = 64: two aligned 128-B lines)
    if ((rc = slk_ensure(ctx, ctx->snap, (size_t)bsz * RS * 4))) return rc;
    // user-row ping-pong of a training scope (slk_user_pingpong_begin): pair mode over plain tables with a row-sparse optimizer on
    // the launch path -- the user pass writes the updated row to the other copy and no record, the item pass gathers the pre-step
    // row from the copy the user pass read.  A call the scope cannot serve is refused, not trained beside it.
    const bool pingpong = ctx->pp_active && ctx->pp_src_u == tables->d_param[0];
    if (ctx->pp_active && !pingpong && !reserve_only)
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: a user-row ping-pong (slk_user_pingpong_begin) is open on this ctx for OTHER "
                                         "tables; close it (slk_user_pingpong_end / _abort) before training another model");
    if (pingpong && (pre || bloom || dense || tables->num_users != ctx->pp_rows || D != ctx->pp_dim))
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: the user rows are ping-ponged (slk_user_pingpong_begin): only the pair losses "
                                         "(pointwise, bpr, hinge) over plain tables of the scope's shape with a row-sparse optimizer train inside the scope");
    // The single-occurrence fast path (k_user_pass<..., SGL>): inside a ping-pong scope, row-sparse Adagrad, item tables of at
    // least "item_single_min_items" rows (default 2^24: a catalogue far larger than a minibatch, where nearly every occurrence is
    // the only one of its item; at C2 -- two occurrences per item on average -- 12 % qualify and the flags would cost more)
    const bool sgl_on = pingpong && optim->kind == SLK_OPT_ADAGRAD && ctx->opt_item_single_min_items > 0 &&
                        tables->num_items >= ctx->opt_item_single_min_items;
    // dL/dscore per (position, pair); ping-pong: {dL/dscore, src} per (position, pair)
    if ((rc = slk_ensure(ctx, ctx->extra[BL_GSN], (size_t)bsz * NP * (pingpong ? 8 : 4)))) return rc;
    const unsigned max_grid = (unsigned)ctx->num_cus * (unsigned)(ctx->opt_user_grid_mult > 8 ? ctx->opt_user_grid_mult : 8);
    if ((rc = slk_ensure(ctx, ctx->losspart, (size_t)max_grid * 8))) return rc;
    if (pre) {
        if ((rc = slk_ensure(ctx, ctx->gk, nc_max * NP * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->sk, nc_max * NP * 4))) return rc;
    }
    if (late) {
        // (the live lists are sorted per minibatch on the PASSES' stream; with the overlapped prep another chunk's sorts run on
        // the prep stream at the same time, so these sorts have temporary storage of their own: BL_LATE_SORT, not ctx->sort_tmp)
        const size_t nl = 2 * (size_t)bsz, nlh = nl * (size_t)(Hi ? Hi : 1);
        if ((rc = slk_ensure(ctx, ctx->extra[BL_LIVE], nl * 4))) return rc;
        for (int b = 0; b < 4; ++b)
            if ((rc = slk_ensure(ctx, ctx->extra[BL_LK0 + b], nlh * 4))) return rc;
    }
    // hot users (k_user_pass<..., ULONG>): two partials per tile of SLK_USER_TILE positions, metas + the long-run counter
    const int UPS = RS + 4;
    const size_t utiles = ((size_t)bsz + SLK_USER_TILE - 1) / SLK_USER_TILE;
    if (!bloom) {
        if ((rc = slk_ensure(ctx, ctx->extra[BL_UPART], 2 * utiles * (size_t)UPS * 4))) return rc;
        const size_t meta_bytes = 2 * utiles * 8 + 64;
        if (meta_bytes > ctx->upart_meta.cap) {
            if ((rc = slk_ensure(ctx, ctx->upart_meta, meta_bytes))) return rc;
            SLK_HIP(ctx, hipMemsetAsync(ctx->upart_meta.p, 0, ctx->upart_meta.cap, s));  // stamp 0 is never used
        }
    }
    const int RSU = D + 4;  // user-bloom gradient record (+ an unused bias slot)
    if (Hu && (rc = slk_ensure(ctx, ctx->extra[BL_UREC], (size_t)bsz * RSU * 4))) return rc;
    // minibatches of a few thousand interactions: every minibatch of a chunk inside ONE persistent launch (slk_epoch.hip)
    // the item-bias shadow of a training scope (slk_bias_shadow_begin): the launch path's kernels index it, the persistent
    // kernel does not -- a call whose biases are shadowed takes the launches
    const bool shadowed = ctx->shadow_active && ctx->shadow_src_p == tables->d_param[3] && !dense &&
                          optim->kind == SLK_OPT_ADAGRAD && ctx->shadow_src_s == optim->d_state1[3];
    if (ctx->shadow_active && !shadowed && ctx->shadow_src_p == tables->d_param[3])
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: the item biases are shadowed (slk_bias_shadow_begin) for another optimizer state");
    // One scope per ctx, and the scope is a TRAINING scope of ONE model (ABI 11): a training call on other tables while it is
    // open is a caller that has lost track of the scope -- refused, not trained beside it (the scope's _end would otherwise
    // write through pointers whose owner may have moved on).  slk_bilinear_reserve allocates only and stays allowed.
    if (ctx->shadow_active && !shadowed && !reserve_only)
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: an item-bias shadow (slk_bias_shadow_begin) is open on this ctx for OTHER "
                                         "tables; close it (slk_bias_shadow_end / _abort) before training another model");
    if (shadowed && !prefetch_only && !reserve_only) ++ctx->stat_shadowed;
    if (pingpong && !prefetch_only && !reserve_only) ++ctx->stat_pingpong;
    bool epoch_route = !shadowed && !pingpong && (!pre || adaptive || (expl && ctx->opt_explicit_fused)) && slk_epoch_eligible(ctx, tables, optim, bsz, loss, bloom);
    auto ensure_dense_buffers = [&]() -> int {
        const size_t elems[4] = {(size_t)(Hu ? ubd.rows : tables->num_users) * D,
                                 (size_t)(Hi ? ibd.rows : tables->num_items) * D, (size_t)tables->num_users,
                                 (size_t)tables->num_items};
        return slk_ensure_dgrad(ctx, elems, 15u, s);
    };
    if (dense && !epoch_route && (rc = ensure_dense_buffers())) return rc;

    if (reserve_only) {
        if (side && (rc = slk_prep_stream_init(ctx))) return rc;
        // the pinned read-back buffers and events of the long-run flags, per buffer set: hipHostMalloc inside a training call
        // costs a device synchronisation (measured: the first overlapped call of a process 0.85 instead of 0.745 ms per step)
        for (int st = 0; st < nsets; ++st) {
            slk_prep_bufs &pb = ctx->pb[st];
            if ((rc = slk_ensure_lflags_host(ctx, pb, 2 * (size_t)mb_per_chunk))) return rc;
            pb.h_lflags_n = 0;
            if (!pb.ev_lflags) SLK_HIP(ctx, hipEventCreateWithFlags(&pb.ev_lflags, hipEventDisableTiming));
        }
        if (side && !ctx->prep_warmed && nn > 0 && nc_max >= 4096) {
            // One tiny prep on the prep stream, ordered against the caller's stream by the pipeline's own events: whatever the
            // runtime sets up the first time these kernels (sampler, sorts) run on a stream and the first time two
            // streams wait for each other's events then happens here and not inside the first overlapped training call
            // (measured: that call ran 0.81-0.87 instead of 0.75-0.77 ms per step; a prior overlapped call of two minibatches
            // removes it, profiles/r03_p_*).  The RNG state is put back afterwards.
            hipStream_t ps = ctx->prep_stream;
            slk_prep_bufs &pb = ctx->pb[1];
            if ((rc = slk_ensure(ctx, ctx->extra[BL_LIVE], sizeof(slk_rng_dev) + 64))) return rc;
            SLK_HIP(ctx, hipEventRecord(ctx->ev_start, s));
            SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_start, 0));
            SLK_HIP(ctx, hipMemcpyAsync(ctx->extra[BL_LIVE].p, ctx->d_rng, sizeof(slk_rng_dev), hipMemcpyDeviceToDevice, ps));
            if ((rc = slk_sample_u32(ctx, tables->num_items, 4096, (uint32_t *)pb.neg32.p, nullptr, ps))) return rc;
            SLK_HIP(ctx, hipMemcpyAsync(ctx->d_rng, ctx->extra[BL_LIVE].p, sizeof(slk_rng_dev), hipMemcpyDeviceToDevice, ps));
            SLK_HIP(ctx, hipMemsetAsync(pb.ukey[0].p, 0, 4096 * 4, ps));
            if ((rc = slk_sort_pairs_u32_u64(ctx, (const uint32_t *)pb.ukey[0].p, (uint32_t *)pb.ukey[1].p, (const uint64_t *)pb.uval[0].p,
                                             (uint64_t *)pb.uval[1].p, 4096, 20, ps)))
                return rc;
            if (!late) {
                SLK_HIP(ctx, hipMemsetAsync(pb.ikey[0].p, 0, 4096 * 4, ps));
                if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)pb.ikey[0].p, (uint32_t *)pb.ikey[1].p, (const uint32_t *)pb.ipay[0].p,
                                                 (uint32_t *)pb.ipay[1].p, 4096, 20, ps)))
                    return rc;
            }
            SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[1], ps));
            SLK_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_prep[1], 0));
            SLK_HIP(ctx, hipEventRecord(ctx->ev_done[1], s));
            SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_done[1], 0));
            SLK_HIP(ctx, hipStreamSynchronize(ps));
            SLK_HIP(ctx, hipStreamSynchronize(s));
            ctx->last_stream = s;
            ctx->sampled_valid = false;
            ctx->prep_warmed = true;
        }
        if (epoch_route && (rc = slk_epoch_reserve(ctx, tables, optim, (uint32_t)((nc_max + bsz - 1) / bsz), bsz, (int)loss, NP))) return rc;
        // sampler and sort scratch for the largest chunk (the whole call when its negatives are one draw), so that the training
        // call allocates nothing
        const bool all_r = nn > 0 && n_chunks > 1 && (uint64_t)n * (uint64_t)nn < ((uint64_t)1 << 30);
        if (all_r && (rc = slk_ensure(ctx, ctx->call_neg, (size_t)n * nn * 4))) return rc;
        if ((rc = slk_sample_reserve(ctx, tables->num_items, all_r ? n * (int64_t)nn : (int64_t)nc_max * nn))) return rc;
        return slk_sort_reserve(ctx, nc_max * (size_t)occ_mult);
    }

    const int upd = slk_upd_for(optim->kind);
    pass_fn upass = nullptr, spass = nullptr, upass_lat = nullptr, upass_long = nullptr, upass_lat_long = nullptr, ustitch = nullptr;
    pass_fn upass_sgl = nullptr;
    slk_item_fns ipass = {nullptr, nullptr}, ipass_rows = ipass, ipass_bias = ipass, rpass_rows = ipass, ipass_sgl = ipass;
    const int umode = (expl && ctx->opt_explicit_fused) ? 2 : (pre ? 1 : 0);
#define SLK_PICK(V_, G_)                                                                  \
    do {                                                                                  \
        upass = user_pass_fn<V_, G_>(upd, umode, bloom);                                  \
        if (umode == 0 && !bloom) upass_lat = user_pass_lat_fn<V_, G_>(upd);              \
        if (umode == 0 && !bloom) upass_lat_long = user_pass_lat_fn<V_, G_, true>(upd);   \
        if (!bloom) upass_long = user_pass_long_fn<V_, G_>(upd, umode);                   \
        if (!bloom) ustitch = user_stitch_fn<V_, G_>(upd);                                \
        ipass = slk_item_pass_fn<V_, G_, SLK_ITEM_SNAP>(upd);                             \
        spass = k_score_pass<V_, G_>;                                                     \
        if (pingpong) {                                                                   \
            upass = user_pass_pp_fn<V_, G_, false, false>(upd);                           \
            upass_lat = user_pass_pp_fn<V_, G_, true, false>(upd);                        \
            upass_long = user_pass_pp_fn<V_, G_, false, true>(upd);                       \
            upass_lat_long = user_pass_pp_fn<V_, G_, true, true>(upd);                    \
            ustitch = user_stitch_pp_fn<V_, G_>(upd);                                     \
            ipass = item_pass_pp_fn<V_, G_>(upd);                                         \
            if (sgl_on) {                                                                 \
                upass_sgl = user_pass_sgl_fn<V_, G_>();                                   \
                ipass_sgl = item_pass_sgl_fn<V_, G_>();                                   \
            }                                                                             \
        }                                                                                 \
        if (bloom) {                                                                      \
            ipass_rows = slk_item_pass_fn<V_, G_, SLK_ITEM_SNAP, SLK_PART_ROWS>(upd);     \
            ipass_bias = slk_item_pass_fn<V_, G_, SLK_ITEM_SNAP, SLK_PART_BIAS>(upd);     \
            rpass_rows = slk_item_pass_fn<V_, G_, SLK_ITEM_ROW, SLK_PART_ROWS>(upd);      \
        }                                                                                 \
    } while (0)
    SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
    const unsigned gpb = 256u / (unsigned)g;
    // The user pass is a grid-stride kernel: workgroups beyond what a CU holds RESIDENT start when others finish and unbalance
    // the pass.  Its grid is therefore capped at the kernel's own occupancy (slk_occupancy_of: the SparseAdam form needs 72
    // VGPRs = 7 workgroups per CU; launched 8 per CU it ran 0.58 ms, capped 0.44; Adagrad's 59 VGPRs hold 8).
    // (one cap for the forms a minibatch may take -- plain, latency-bound, long runs: the grid also counts the loss partials)
    int occ_user = slk_occupancy_of(ctx, upass);
    for (pass_fn f : {upass_lat, upass_long, upass_lat_long}) {
        const int o = slk_occupancy_of(ctx, f);
        if (o < occ_user) occ_user = o;
    }

    // ---- prep of one chunk into buffer set `pb` (value-independent: ids only), in two halves:
    // the negatives (ALU/latency-bound MT19937 generator), then the sorts (HBM-bound)
    // negatives of the whole call drawn ahead by slk_bilinear_prefetch (ctx->pf_neg): the chunks read theirs from it
    const uint32_t *neg_all = nullptr;
    auto do_sample = [&](size_t ck, slk_prep_bufs &pb, hipStream_t s) -> int {
        int rc;
        const int64_t c0 = cb[ck];
        const uint32_t nc = (uint32_t)(cb[ck + 1] - c0);
        uint32_t *neg32 = (uint32_t *)pb.neg32.p;
        if (neg_all) return SLK_OK;

        // ---- negatives (sampling.py:34, one randint per minibatch == one contiguous draw)
        if (nn == 0) return SLK_OK;  // explicit feedback draws none
        if (d_neg_in) {
            slk_prof_begin(ctx, SLK_K_SAMPLE, s);
            hipLaunchKernelGGL(k_i64_to_u32, dim3(slk_grid_for(ctx, (size_t)nc * nn, 256)), dim3(256), 0, s,
                               d_neg_in + c0 * nn, neg32, (size_t)nc * nn);
            SLK_LAUNCH_CHECK(ctx, "k_i64_to_u32");
            if (d_neg_out)
                SLK_HIP(ctx, hipMemcpyAsync(d_neg_out + c0 * nn, d_neg_in + c0 * nn, (size_t)nc * nn * 8,
                                            hipMemcpyDeviceToDevice, s));
            slk_prof_end(ctx, s);
        } else {
            if ((rc = slk_sample_u32(ctx, tables->num_items, (int64_t)nc * nn, neg32,
                                     d_neg_out ? d_neg_out + c0 * nn : nullptr, s)))
                return rc;
        }
        return SLK_OK;
    };
    auto do_sort = [&](size_t ck, slk_prep_bufs &pb, hipStream_t s) -> int {
        int rc;
        const int64_t c0 = cb[ck];
        const uint32_t nc = (uint32_t)(cb[ck + 1] - c0);
        const uint32_t nocc = nc * (uint32_t)NP;
        const int64_t *cu = d_users + c0, *ci = d_items + c0;
        const uint32_t *neg32 = neg_all ? neg_all + (size_t)c0 * nn : (const uint32_t *)pb.neg32.p;

        // ---- prep: sort interactions by (minibatch, user), occurrences by (minibatch, item)
        slk_prof_begin(ctx, SLK_K_PREP, s);
        const unsigned mbbits = slk_bits_for((uint64_t)((nc - 1) / (uint32_t)bsz));
        uint32_t *ukey_in = (uint32_t *)pb.ukey[0].p, *ukey = (uint32_t *)pb.ukey[1].p;
        const uint32_t *uit, *uk = nullptr;
        uint32_t *const ukeys[2] = {ukey_in, ukey};
        if (!pre) {
            // (keys and the fat (positive, negative) payload are formed by the sort's first pass from the id arrays)
            uint64_t *const uvals[2] = {(uint64_t *)pb.uval[0].p, (uint64_t *)pb.uval[1].p};
            if ((rc = slk_sort_user_fat(ctx, cu, ci, (const uint32_t *)neg32, nc, (size_t)bsz, ubits, mbbits, ukeys, uvals, s))) return rc;
            uit = (const uint32_t *)pb.uval[1].p;  // little-endian (pos, neg) pairs
        } else {
            uint32_t *const uvals[2] = {(uint32_t *)pb.uval[0].p, (uint32_t *)pb.uval[1].p};
            if ((rc = slk_sort_user_idx(ctx, cu, nc, (size_t)bsz, ubits, mbbits, ukeys, uvals, s))) return rc;
            uk = (const uint32_t *)pb.uval[1].p;
            hipLaunchKernelGGL(k_pack_items, dim3(slk_grid_for(ctx, nc, 256)), dim3(256), 0, s, uk, ci,
                               (const uint32_t *)neg32, nc, nn, (uint32_t *)pb.uit.p);
            SLK_LAUNCH_CHECK(ctx, "k_pack_items");
            uit = (const uint32_t *)pb.uit.p;
        }
        if (late) {
            hipLaunchKernelGGL(k_invert_perm, dim3(slk_grid_for(ctx, nc, 256)), dim3(256), 0, s, uk, nc,
                               (uint32_t *)pb.ipay[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_invert_perm");
        } else {
            uint32_t *const ikeys[2] = {(uint32_t *)pb.ikey[0].p, (uint32_t *)pb.ikey[1].p};
            uint32_t *const ivals[2] = {(uint32_t *)pb.ipay[0].p, (uint32_t *)pb.ipay[1].p};
            if ((rc = slk_sort_item_occ(ctx, uit, nocc, (size_t)bsz, NP, ibits, mbbits, ikeys, ivals, s))) return rc;
            if (sgl_on) {  // which occurrences are NOT the only one of their item in their minibatch (ids only)
                SLK_HIP(ctx, hipMemsetAsync(pb.mflag.p, 0, (size_t)nocc, s));
                hipLaunchKernelGGL(k_item_multi_flags, dim3(slk_grid_for(ctx, nocc, 256)), dim3(256), 0, s, (const uint32_t *)pb.ikey[1].p,
                                   (const uint32_t *)pb.ipay[1].p, nocc, (uint32_t)bsz * (uint32_t)NP, (uint8_t *)pb.msorted.p,
                                   (uint8_t *)pb.mflag.p);
                SLK_LAUNCH_CHECK(ctx, "k_item_multi_flags");
            }
        }
        // which minibatches hold a LONG run (an item run that wholly covers a tile of the item pass, [0, n_mb); a user run that
        // wholly covers a tile of the user pass, [n_mb, 2 n_mb)): ids only, so the answer is fetched once per chunk and the usual
        // minibatch (none) gets the plain passes with no stitch kernel behind them
        if (!epoch_route) {  // (the persistent route walks the runs itself; its fall-back takes the long forms, do_passes)
            const uint32_t n_mb_c = (nc + (uint32_t)bsz - 1) / (uint32_t)bsz;
            if ((rc = slk_ensure(ctx, pb.lflags, 2 * (size_t)n_mb_c * 4))) return rc;
            SLK_HIP(ctx, hipMemsetAsync(pb.lflags.p, 0, 2 * (size_t)n_mb_c * 4, s));
            if (!late) {
                hipLaunchKernelGGL(k_item_long_flags, dim3(slk_grid_for(ctx, nocc / (4 * gpb) + n_mb_c, 256)), dim3(256), 0, s,
                                   (const uint32_t *)pb.ikey[1].p, nocc, (uint32_t)bsz * (uint32_t)NP, 4u * gpb,
                                   (uint32_t)((1ull << ibits) - 1), 0xffffffffu, 0xffffffffu, (int *)pb.lflags.p);
                SLK_LAUNCH_CHECK(ctx, "k_item_long_flags");
            }
            if (!bloom) {
                hipLaunchKernelGGL(k_user_long_flags, dim3(slk_grid_for(ctx, nc / SLK_USER_TILE + n_mb_c, 256)), dim3(256), 0, s,
                                   (const uint32_t *)ukey, nc, (uint32_t)bsz, (int *)pb.lflags.p + n_mb_c);
                SLK_LAUNCH_CHECK(ctx, "k_user_long_flags");
            }
            if ((rc = slk_ensure_lflags_host(ctx, pb, 2 * (size_t)n_mb_c))) return rc;
            SLK_HIP(ctx, hipMemcpyAsync(pb.h_lflags, pb.lflags.p, 2 * (size_t)n_mb_c * 4, hipMemcpyDeviceToHost, s));
            if (!pb.ev_lflags) SLK_HIP(ctx, hipEventCreateWithFlags(&pb.ev_lflags, hipEventDisableTiming));
            SLK_HIP(ctx, hipEventRecord(pb.ev_lflags, s));
        } else {
            pb.h_lflags_n = 0;  // no flags for this chunk: every pass of a fall-back takes the partial-writing form
        }
        if (Hi && !late) {
            hipLaunchKernelGGL(k_build_item_bloom_keys, dim3(slk_grid_for(ctx, (size_t)nocc * Hi, 256)), dim3(256), 0, s,
                               uit, nocc, NP, (uint32_t)bsz, icbits, ibd, (uint32_t *)pb.bik[0].p,
                               (uint32_t *)pb.bip[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_build_item_bloom_keys");
            if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)pb.bik[0].p,
                                             (uint32_t *)pb.bik[1].p,
                                             (const uint32_t *)pb.bip[0].p,
                                             (uint32_t *)pb.bip[1].p, (size_t)nocc * Hi, icbits + mbbits, s, true)))
                return rc;
        }
        if (Hu) {
            hipLaunchKernelGGL(k_build_user_bloom_keys, dim3(slk_grid_for(ctx, (size_t)nc * Hu, 256)), dim3(256), 0, s,
                               (const uint32_t *)ukey, (uint32_t)((1ull << ubits) - 1), nc, (uint32_t)bsz, ucbits, ubd,
                               (uint32_t *)pb.buk[0].p, (uint32_t *)pb.bup[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_build_user_bloom_keys");
            if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)pb.buk[0].p,
                                             (uint32_t *)pb.buk[1].p,
                                             (const uint32_t *)pb.bup[0].p,
                                             (uint32_t *)pb.bup[1].p, (size_t)nc * Hu, ucbits + mbbits, s, true)))
                return rc;
        }
        slk_prof_end(ctx, s);

        return SLK_OK;
    };

    int64_t mb_global = 0;
    // ---- the minibatches of one prepared chunk, in order, on the caller's stream
    auto do_passes = [&](size_t ck, slk_prep_bufs &pb) -> int {
        int rc;
        const int64_t c0 = cb[ck];
        const uint32_t nc = (uint32_t)(cb[ck + 1] - c0);
        const uint32_t *ukey = (const uint32_t *)pb.ukey[1].p;
        const uint32_t *uit = pre ? (const uint32_t *)pb.uit.p : (const uint32_t *)pb.uval[1].p;
        const uint32_t *uk = pre ? (const uint32_t *)pb.uval[1].p : nullptr;
        bool lflags_ready = false;  // the chunk's long-run flags (do_sort): ONE host wait per chunk, before the first user pass
        const uint32_t n_mb_c = (nc + (uint32_t)bsz - 1) / (uint32_t)bsz;
        // ---- minibatches, in order
        for (uint32_t b0 = 0; b0 < nc; b0 += (uint32_t)bsz, ++mb_global) {
            const uint32_t b1 = (nc - b0 < (uint32_t)bsz) ? nc : b0 + (uint32_t)bsz;
            const uint32_t bm = b1 - b0;
            slk_pass_args a;
            memset(&a, 0, sizeof(a));
            for (int t = 0; t < 4; ++t) {
                a.P[t] = tables->d_param[t];
                a.S1[t] = dense ? (float *)ctx->dgrad[t].p : optim->d_state1[t];
                a.S2[t] = optim->d_state2[t];
            }
            if (shadowed) {
                a.P[3] = (float *)ctx->bias_shadow.p;
                a.S1[3] = (float *)ctx->bias_shadow.p + 1;
                a.bsh3 = 1u;
            }
            if (pingpong) {
                a.P0alt = (float *)ctx->pp_alt.p;
                a.uflag = (uint8_t *)ctx->pp_flags.p;
            }
            a.D = D;
            a.NP = NP;
            a.begin = b0;
            a.end = b1;
            a.ukey = ukey;
            a.umask = (uint32_t)((1ull << ubits) - 1);
            a.uit = uit;
            a.uk = uk;
            a.gk = (const float *)ctx->gk.p;
            a.sk = (float *)ctx->sk.p;
            a.snap = (float *)ctx->snap.p;
            a.RS = RS;
            a.gsn = (float *)ctx->extra[BL_GSN].p;
            a.ikey = (const uint32_t *)pb.ikey[1].p;
            a.imask = (uint32_t)((1ull << ibits) - 1);
            a.ipay = (const uint32_t *)pb.ipay[1].p;
            a.loss_partial = (double *)ctx->losspart.p;
            a.mb_loss_out = d_mb_loss + mb_global;
            a.loss_kind = loss;
            a.inv_b = 1.0f / (float)bm;
            a.ibegin = b0 * (uint32_t)NP;
            a.iend = b1 * (uint32_t)NP;
            a.pad_item = a.pad_item2 = 0xffffffffu;
            a.ub = ubd;
            a.ib = ibd;
            a.urec = (float *)ctx->extra[BL_UREC].p;
            a.RSU = RSU;
            slk_set_opt_coeffs(a, optim);
            // user biases the caller vouches are all zero, a loss whose user-bias gradient is exactly zero, an update for which a
            // zero gradient is an exact no-op: nothing of the table is read or written by the pair-mode user pass
            a.ubz = (tables->flags & SLK_TABLES_USER_BIAS_ZERO) && (loss == SLK_LOSS_BPR || loss == SLK_LOSS_HINGE) && !bloom &&
                    (optim->kind == SLK_OPT_ADAGRAD || optim->kind == SLK_OPT_SGD) && ctx->opt_user_bias_zero_hint;
            a.nt = ctx->opt_nt;
            // Round 6 (profiles/r06_mall_ab.jsonl): a minibatch whose records are as large as the Infinity Cache (2^20 x 256 B
            // = 256 MB) cannot keep them there until the item pass reads them -- written with plain stores they only evict the
            // item table, which the user pass re-reads twice per interaction.  Non-temporal record stores: C2 step -0.9 %
            // (item pass 0.3165 -> 0.3122 ms), C5 shard -0.2 %; a minibatch of 65 536 (16 MB of records: they DO stay) +4 %, so
            // only above "record_nt_min_bytes".
            if (ctx->opt_record_nt_min_bytes > 0 && (int64_t)bm * RS * 4 >= ctx->opt_record_nt_min_bytes) a.nt |= 16;
            // overlapped prep: the user pass is a grid-stride kernel whose workgroups hold their wave slots for the whole run; two
            // of the eight per CU are left to the prep stream's kernels (measured, profiles/r03_a_*: 8 -> 6 costs the pass
            // nothing by itself and gives the overlap 2 % more)
            const int ugm = side && ctx->opt_user_grid_mult > 6 ? 6 : ctx->opt_user_grid_mult;
            const unsigned ugrid = slk_grid_for(ctx, bm, gpb, ugm < occ_user ? ugm : occ_user);

            if (expl && ctx->opt_explicit_fused) {
                // the user pass forms score, loss and dL/dscore itself (one pair per interaction)
                a.ratings = d_ratings + c0;
                a.n_loss_partial = (int)ugrid;
            } else if (expl) {
                slk_prof_begin(ctx, SLK_K_SCORE, s);
                hipLaunchKernelGGL(spass, dim3(ugrid), dim3(256), 0, s, a);
                SLK_LAUNCH_CHECK(ctx, "k_score_pass");
                const unsigned sgrid = slk_grid_for(ctx, bm, 256);
                hipLaunchKernelGGL(k_explicit_loss, dim3(sgrid), dim3(256), 0, s, (const float *)ctx->sk.p,
                                   d_ratings + c0, (float *)ctx->gk.p, b0, bm, (int)loss, a.inv_b,
                                   (double *)ctx->losspart.p);
                SLK_LAUNCH_CHECK(ctx, "k_explicit_loss");
                a.n_loss_partial = (int)sgrid;
                slk_prof_end(ctx, s);
            } else if (adaptive) {
                slk_prof_begin(ctx, SLK_K_SCORE, s);
                hipLaunchKernelGGL(spass, dim3(ugrid), dim3(256), 0, s, a);
                SLK_LAUNCH_CHECK(ctx, "k_score_pass");
                const unsigned sgrid = slk_grid_for(ctx, bm, 256);
                hipLaunchKernelGGL(k_adaptive_select<0>, dim3(sgrid), dim3(256), 0, s, (const float *)ctx->sk.p,
                                   (float *)ctx->gk.p, b0, bm, nn, a.inv_b, (double *)ctx->losspart.p,
                                   late ? (const uint32_t *)pb.ipay[0].p : (const uint32_t *)nullptr,
                                   late ? (uint32_t *)ctx->extra[BL_LIVE].p : (uint32_t *)nullptr);
                SLK_LAUNCH_CHECK(ctx, "k_adaptive_select");
                a.n_loss_partial = (int)sgrid;
                slk_prof_end(ctx, s);
                if (late) {
                    // this minibatch's live occurrences, sorted by item (biases; rows of a plain table)
                    slk_prof_begin(ctx, SLK_K_PREP, s);
                    const uint32_t nl = 2 * bm;
                    hipLaunchKernelGGL(k_build_live_keys, dim3(slk_grid_for(ctx, nl, 256)), dim3(256), 0, s,
                                       (const uint32_t *)ctx->extra[BL_LIVE].p, uit, nl, ibd, false,
                                       (uint32_t *)ctx->extra[BL_LK0].p, (uint32_t *)ctx->extra[BL_LV0].p);
                    SLK_LAUNCH_CHECK(ctx, "k_build_live_keys");
                    if ((rc = slk_sort_pairs_u32_u32_in(ctx, ctx->extra[BL_LATE_SORT], (const uint32_t *)ctx->extra[BL_LK0].p,
                                                     (uint32_t *)ctx->extra[BL_LK1].p,
                                                     (const uint32_t *)ctx->extra[BL_LV0].p,
                                                     (uint32_t *)ctx->extra[BL_LV1].p, nl, ibits + 1, s, true)))  // + the dead entries' bit
                        return rc;
                    slk_prof_end(ctx, s);
                    a.ikey = (const uint32_t *)ctx->extra[BL_LK1].p;
                    a.ipay = (const uint32_t *)ctx->extra[BL_LV1].p;
                    a.imask = 0xffffffffu;
                    a.ibegin = 0;
                    a.iend = nl;
                    // a.pad_item2 == ~0u: the dead entries' all-ones key is never updated
                }
            } else {
                a.n_loss_partial = (int)ugrid;
            }

            if (!lflags_ready) {
                // the flags are produced by the chunk's sorts, long before the previous chunk's passes are through: the GPU has
                // work queued while the host waits for the read-back
                if (pb.ev_lflags && pb.h_lflags_n) SLK_HIP(ctx, hipEventSynchronize(pb.ev_lflags));
                lflags_ready = true;
            }
            const uint32_t mb_c = b0 / (uint32_t)bsz;
            const bool have_flags = pb.h_lflags_n >= 2 * (size_t)n_mb_c;
            const bool item_may_long = late || !ctx->opt_item_long_gate || !have_flags || pb.h_lflags[mb_c] != 0;
            const bool user_may_long = !bloom && (!ctx->opt_item_long_gate || !have_flags || pb.h_lflags[n_mb_c + mb_c] != 0);

            slk_prof_begin(ctx, SLK_K_USER_PASS, s);
            const bool lat = upass_lat && (int64_t)bm <= ctx->opt_user_lat_max_batch;
            // Option "user_grid_own_occ" (off): the grid of the form that is launched capped at THAT form's occupancy instead of the
            // smallest of the four forms' -- measured in round 6 and not kept (slk_common.h)
            // the single-occurrence fast path rides on the plain (bandwidth-bound) form only: a minibatch that takes the latency-bound
            // or the long-run form leaves every item to the item pass
            const bool sgl = sgl_on && upass_sgl && !user_may_long && !lat;
            if (sgl) {
                a.mflag = (const uint8_t *)pb.mflag.p;
                a.msorted = (const uint8_t *)pb.msorted.p;
                ++ctx->stat_single;
            }
            const pass_fn uform = sgl ? upass_sgl : (user_may_long ? (lat ? upass_lat_long : upass_long) : (lat ? upass_lat : upass));
            unsigned ugrid_form = ugrid;
            if (ctx->opt_user_grid_own_occ || sgl) {  // (the single-occurrence form holds more registers than the four whose smallest occupancy caps the others)
                const int occ_form = slk_occupancy_of(ctx, uform);
                ugrid_form = slk_grid_for(ctx, bm, gpb, ugm < occ_form ? ugm : occ_form);
                if (!pre || (expl && ctx->opt_explicit_fused)) a.n_loss_partial = (int)ugrid_form;  // (the pass writes one loss partial per workgroup)
            }
            if (user_may_long) {
                if (ctx->upart_gen >= 0x0ffffffeu) {  // the 28-bit stamp wraps: forget every old partial
                    SLK_HIP(ctx, hipMemsetAsync(ctx->upart_meta.p, 0, ctx->upart_meta.cap, s));
                    ctx->upart_gen = 0u;
                }
                ++ctx->upart_gen;
                a.upart = (float *)ctx->extra[BL_UPART].p;
                a.upart_meta = (uint32_t *)ctx->upart_meta.p;
                a.upart_count = a.upart_meta + 4 * utiles;
                a.upart_gen = ctx->upart_gen;
                a.UPS = UPS;
                SLK_HIP(ctx, hipMemsetAsync(a.upart_count, 0, 4, s));
                hipLaunchKernelGGL(uform, dim3(ugrid_form), dim3(256), 0, s, a);
                SLK_LAUNCH_CHECK(ctx, "k_user_pass<ULONG>");
                ++ctx->stat_user_long;
                hipLaunchKernelGGL(ustitch, dim3(slk_grid_for(ctx, ((size_t)bm + SLK_USER_TILE - 1) / SLK_USER_TILE, gpb)), dim3(256), 0, s, a);
                SLK_LAUNCH_CHECK(ctx, "k_user_stitch");
            } else {
                hipLaunchKernelGGL(uform, dim3(ugrid_form), dim3(256), 0, s, a);
                SLK_LAUNCH_CHECK(ctx, "k_user_pass");
            }
            slk_prof_end(ctx, s);
            slk_prof_begin(ctx, SLK_K_ITEM_PASS, s);
            if (!Hi) {
                if ((rc = slk_launch_item_pass(ctx, sgl ? ipass_sgl : ipass, a, g, s, "k_item_pass", item_may_long))) return rc;
            } else {
                // item biases are indexed by the item id: plain occurrence list, bias only ...
                if ((rc = slk_launch_item_pass(ctx, ipass_bias, a, g, s, "k_item_pass<BIAS>", item_may_long))) return rc;
                // ... while every occurrence feeds the n_hash hashed rows of the compressed table
                slk_pass_args r = a;
                r.mb_loss_out = nullptr;
                r.ikey = (const uint32_t *)pb.bik[1].p;
                r.ipay = (const uint32_t *)pb.bip[1].p;
                r.ibegin = a.ibegin * (uint32_t)Hi;
                r.iend = a.iend * (uint32_t)Hi;
                r.imask = (uint32_t)((1ull << icbits) - 1);
                if (late) {
                    // the live occurrences' hashed rows (the bias pass above has consumed LK1/LV1:
                    // same stream, so the buffers can be reused)
                    const uint32_t nlh = 2 * bm * (uint32_t)Hi;
                    hipLaunchKernelGGL(k_build_live_keys, dim3(slk_grid_for(ctx, nlh, 256)), dim3(256), 0, s,
                                       (const uint32_t *)ctx->extra[BL_LIVE].p, uit, 2 * bm, ibd, true,
                                       (uint32_t *)ctx->extra[BL_LK0].p, (uint32_t *)ctx->extra[BL_LV0].p);
                    SLK_LAUNCH_CHECK(ctx, "k_build_live_keys<hashed>");
                    if ((rc = slk_sort_pairs_u32_u32_in(ctx, ctx->extra[BL_LATE_SORT], (const uint32_t *)ctx->extra[BL_LK0].p,
                                                     (uint32_t *)ctx->extra[BL_LK1].p,
                                                     (const uint32_t *)ctx->extra[BL_LV0].p,
                                                     (uint32_t *)ctx->extra[BL_LV1].p, nlh, icbits + 1, s, true)))
                        return rc;
                    r.ikey = (const uint32_t *)ctx->extra[BL_LK1].p;
                    r.ipay = (const uint32_t *)ctx->extra[BL_LV1].p;
                    r.ibegin = 0;
                    r.iend = nlh;
                    r.imask = 0xffffffffu;
                }
                r.pad_item = tables->item_bloom->skip_row < 0 ? 0xffffffffu : (uint32_t)tables->item_bloom->skip_row;
                if ((rc = slk_launch_item_pass(ctx, ipass_rows, r, g, s, "k_item_pass<ROWS>"))) return rc;
            }
            if (Hu) {
                // owner pass over the hashed USER rows: table slot 1 of the pass is remapped onto
                // the user embedding table, records are the segment gradients parked by the user pass
                slk_pass_args r = a;
                r.mb_loss_out = nullptr;
                r.P[1] = a.P[0];
                r.S1[1] = a.S1[0];
                r.S2[1] = a.S2[0];
                r.snap = a.urec;
                r.RS = RSU;
                r.ikey = (const uint32_t *)pb.buk[1].p;
                r.ipay = (const uint32_t *)pb.bup[1].p;
                r.ibegin = b0 * (uint32_t)Hu;
                r.iend = b1 * (uint32_t)Hu;
                r.imask = (uint32_t)((1ull << ucbits) - 1);
                r.pad_item = tables->user_bloom->skip_row < 0 ? 0xffffffffu : (uint32_t)tables->user_bloom->skip_row;
                r.pad_item2 = ubd.rows;  // sentinel of non-head positions
                if ((rc = slk_launch_item_pass(ctx, rpass_rows, r, g, s, "k_item_pass<ROW,ROWS>"))) return rc;
            }
            slk_prof_end(ctx, s);

            if (dense && (rc = slk_dense_sweeps(ctx, tables->d_param, optim, 15u, s))) return rc;
            optim->step += 1;
        }
        return SLK_OK;
    };

    auto do_chunk = [&](size_t ck, slk_prep_bufs &pb) -> int {
        if (!epoch_route) return do_passes(ck, pb);
        const int64_t c0 = cb[ck];
        const uint32_t nc = (uint32_t)(cb[ck + 1] - c0);
        int rc = slk_epoch_run_chunk(ctx, tables, optim, pb, nc, bsz, ubits, ibits, (int)loss, NP, RS, (float *)ctx->snap.p,
                                     (float *)ctx->extra[BL_GSN].p, d_mb_loss + mb_global, expl ? d_ratings + c0 : nullptr, s);
        if (rc == SLK_EAGAIN_EPOCH) {  // cooperative launch refused: nothing ran; per-minibatch launches from here on
            epoch_route = false;
            if (dense && (rc = ensure_dense_buffers())) return rc;
            return do_passes(ck, pb);
        }
        mb_global += (nc + bsz - 1) / bsz;
        return rc;
    };

    // a chunk prepared ahead by slk_bilinear_prefetch: it must be THIS call's first chunk (its negatives are already drawn)
    // (a call that draws nothing -- negatives handed in, explicit feedback -- cannot be confused with the prepared one: it drops
    // the chunk instead of refusing; fit() of another model after one that left its loop by an exception)
    if (!prefetch_only && ctx->pf.valid && (d_neg_in || expl)) ctx->pf.valid = false;
    const bool have0 = !prefetch_only && ctx->pf.valid;
    if (have0 && (!side || epoch_route || ctx->pf.users != (const void *)d_users || ctx->pf.items != (const void *)d_items ||
                  ctx->pf.n != n || ctx->pf.bsz != bsz || ctx->pf.loss != (int)loss || ctx->pf.nn != nn || ctx->pf.nc0 != cb[1] || d_neg_in)) {
        ctx->pf.valid = false;
        return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_train: a chunk was prepared ahead (slk_bilinear_prefetch) for another call; its "
                                         "draws are consumed -- set the RNG state again");
    }
    if (!prefetch_only) ctx->pf.valid = false;
    if (prefetch_only) {
        ctx->pf.valid = false;
        if (!side || epoch_route || d_neg_in) return SLK_OK;  // such a call prepares in line: nothing to run ahead
        if ((rc = slk_prep_stream_init(ctx))) return rc;
        hipStream_t ps = ctx->prep_stream;
        // the set the call before did NOT finish on: its last reader were the passes two chunks back (ev_done of that set)
        const int set = ctx->last_pipe_set >= 0 ? (ctx->last_pipe_set ^ 1) : 0;
        // What the prepared chunk waits for (it used to wait for everything the caller's stream held -- every pass of
        // the epoch before -- and so ran BEHIND those passes, not beside them): the id arrays are complete when this call is made
        // (the header says so; fit() has read the shuffle's RNG state back, which synchronises the lane that wrote them); buffer
        // set `set` was last read by the passes two chunks back (ev_done); the sampler's and the sorts' scratch and, with
        // "overlap_prep" 1, the earlier readers of pf_neg (the sorts of the running call's later chunks) live on the prep stream
        // itself.  Only with "overlap_prep" 2 do sorts that read pf_neg run on the caller's stream: then, and only then, the
        // new draw waits for that stream's tail.
        //  That holds for a ctx whose last training call was a pipelined one.  After an IN-LINE call (one buffer
        // set, "overlap_prep" 0 / 2: sampler, sorts and buffer set 0 were used on the caller's stream -- ctx->last_pipe_set < 0)
        // the new draw waits for that stream's tail; after an in-line slk_sample_items it waits for that draw's own event.
        if (ctx->opt_overlap_prep != 1 || ctx->opt_prefetch_wait || ctx->last_pipe_set < 0) {
            SLK_HIP(ctx, hipEventRecord(ctx->ev_start, s));
            SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_start, 0));
        } else if (ctx->sampled_valid && ctx->ev_sampled) {
            SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_sampled, 0));  // (recorded on the prep stream itself in the steady state: free)
        }
        SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_done[set], 0));
        // The negatives of the WHOLE call in one draw (it is one contiguous draw however the call is chunked): the stream
        // position behind them -- where the NEXT epoch's shuffle starts -- is then known an epoch ahead
        // (slk_rng_get_state_sampled), and the training call itself draws nothing.  Calls of >= 2^30 draws keep drawing by chunk.
        const bool all = nn > 0 && (uint64_t)n * (uint64_t)nn < ((uint64_t)1 << 30);
        if (all) {
            if ((rc = slk_ensure(ctx, ctx->pf_neg, (size_t)n * nn * 4))) return rc;
            if ((rc = slk_sample_u32(ctx, tables->num_items, n * (int64_t)nn, (uint32_t *)ctx->pf_neg.p, nullptr, ps))) return rc;
            neg_all = (const uint32_t *)ctx->pf_neg.p;
        } else if ((rc = do_sample(0, ctx->pb[set], ps))) {
            return rc;
        }
        if (ctx->opt_overlap_prep == 1 && (rc = do_sort(0, ctx->pb[set], ps))) return rc;
        SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[set], ps));
        ctx->pf.all = all;
        ctx->pf.valid = true;
        ctx->pf.set = set;
        ctx->pf.users = d_users;
        ctx->pf.items = d_items;
        ctx->pf.n = n;
        ctx->pf.bsz = bsz;
        ctx->pf.nc0 = cb[1];
        ctx->pf.loss = (int)loss;
        ctx->pf.nn = nn;
        return SLK_OK;
    }
    if (nsets == 1) {
        // one chunk: everything in order on the caller's stream
        if ((rc = do_sample(0, ctx->pb[0], s))) return rc;
        if ((rc = do_sort(0, ctx->pb[0], s))) return rc;
        if ((rc = do_chunk(0, ctx->pb[0]))) return rc;
        ctx->last_stream = s;
        ctx->last_pipe_set = -1;
        return SLK_OK;
    }
    // pipeline: prep(c + 1) is enqueued before passes(c) -- on ctx->prep_stream, beside them (`side`), or on the caller's own
    // stream, ahead of them.
    // overlap_prep = 1: negatives and sorts; = 2: only the negatives (the generator is ALU-bound and
    // shares the chip with the HBM-bound passes; the sorts stay in line on the caller's stream)
    const bool sort_ahead = !side || ctx->opt_overlap_prep == 1;
    if (side && (rc = slk_prep_stream_init(ctx))) return rc;
    hipStream_t ps = side ? ctx->prep_stream : s;
    if (side) {
        SLK_HIP(ctx, hipEventRecord(ctx->ev_start, s));      // inputs produced on the caller's stream
        SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_start, 0));
    }
    int set = have0 ? ctx->pf.set : 0;
    if (have0) ++ctx->stat_prefetched;
    if (have0 && ctx->pf.all) neg_all = (const uint32_t *)ctx->pf_neg.p;
    if (!have0) {
        // The negatives of the WHOLE call in ONE draw (it is one contiguous draw however the call is chunked -- what
        // slk_bilinear_prefetch does for fit()'s epochs): the generator's jump-ahead, ~130 us whatever the draw's length, is
        // paid once per call instead of once per chunk (round 6: 0.0135 -> 0.0065 ms per C2 step at 20 minibatches per call).
        // Calls of >= 2^30 draws and calls that are handed their negatives keep drawing / converting by chunk.
        if (!d_neg_in && nn > 0 && (uint64_t)n * (uint64_t)nn < ((uint64_t)1 << 30)) {
            if ((rc = slk_ensure(ctx, ctx->call_neg, (size_t)n * nn * 4))) return rc;
            if ((rc = slk_sample_u32(ctx, tables->num_items, n * (int64_t)nn, (uint32_t *)ctx->call_neg.p, d_neg_out, ps))) return rc;
            neg_all = (const uint32_t *)ctx->call_neg.p;
        } else if ((rc = do_sample(0, ctx->pb[set], ps))) {
            return rc;
        }
        if (sort_ahead && (rc = do_sort(0, ctx->pb[set], ps))) return rc;
        if (side) SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[set], ps));
    }
    for (size_t ck = 0; ck < n_chunks; ++ck, set ^= 1) {
        if (side) SLK_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_prep[set], 0));
        if (ck == 0 && have0 && neg_all && d_neg_out) {  // the caller wants the draws: they were made ahead, as uint32
            hipLaunchKernelGGL(k_u32_to_i64, dim3(slk_grid_for(ctx, (size_t)n * nn, 256)), dim3(256), 0, s, neg_all, d_neg_out, (size_t)n * nn);
            SLK_LAUNCH_CHECK(ctx, "k_u32_to_i64");
        }
        if (!sort_ahead && (rc = do_sort(ck, ctx->pb[set], s))) return rc;
        if (ck + 1 < n_chunks) {
            // the other buffer set was last read by the passes of the previous chunk (with a chunk prepared ahead: by the last
            // passes of the call before this one)
            if (side && (ck > 0 || have0)) SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_done[set ^ 1], 0));
            if (!sort_ahead) {
                // "overlap_prep" 2: the next chunk's draw starts behind THIS chunk's sorts, i.e. beside its passes.  Started
                // together with the sorts (round 4) the generator's workgroups -- whole CUs by their LDS -- took the sorts'
                // CUs: the in-line sorts ran 17 % longer and the overlap gained nothing (profiles/r05_b_sweep_overlap_prep.jsonl)
                SLK_HIP(ctx, hipEventRecord(ctx->ev_start, s));
                SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_start, 0));
            }
            if ((rc = do_sample(ck + 1, ctx->pb[set ^ 1], ps))) return rc;
            if (sort_ahead && (rc = do_sort(ck + 1, ctx->pb[set ^ 1], ps))) return rc;
            if (side) {
                SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[set ^ 1], ps));
                ++ctx->stat_overlapped;
            }
        }
        if ((rc = do_chunk(ck, ctx->pb[set]))) return rc;
        if (side) SLK_HIP(ctx, hipEventRecord(ctx->ev_done[set], s));
    }
    if (!side) {  // everything ran on the caller's stream: in-line semantics for what follows (slk_bilinear_prefetch)
        ctx->last_pipe_set = -1;
        ctx->last_stream = s;
        return SLK_OK;
    }
    ctx->last_pipe_set = set ^ 1;  // (the loop's last increment undone: the set of the last chunk)
    ctx->last_stream = s;  // every prep is ordered before the tail of the caller's stream
    return SLK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Item-bias shadow (round 5).  On tables of 10^8 rows a minibatch's item biases share no cache line: the owner pass pays a
// 128-byte read and a 64-byte write for the 4-byte bias AND again for its 4-byte Adagrad accumulator -- a quarter of the C5
// shard's item-pass requests (profiles/r04_ea_c5_pmc.md).  For the duration of a training scope (a fit(): thousands of
// minibatches) the two arrays are held interleaved, {bias, sum} per item, so an occurrence touches ONE line; torch's own
// tensors are stale inside the scope and rewritten by slk_bias_shadow_end.  The reference has no counterpart (its biases are
// an nn.Embedding of dim 1: spotlight/factorization/representations.py:80-91).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bias_shadow_pack(const float *p, const float *s1, float2 *sh, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) sh[i] = make_float2(p[i], s1[i]);
}

__global__ __launch_bounds__(256) void k_bias_shadow_unpack(const float2 *sh, float *p, float *s1, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 v = sh[i];
        p[i] = v.x;
        s1[i] = v.y;
    }
}

SLK_EXPORT int slk_bias_shadow_begin(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, void *stream) {
    if (!ctx || !tables || !optim) return SLK_EINVAL;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->shadow_active) return slk_fail(ctx, SLK_EINVAL, "slk_bias_shadow_begin: a shadow is already active on this ctx");
    if (optim->kind != SLK_OPT_ADAGRAD || !tables->d_param[3] || !optim->d_state1[3] || tables->item_bloom || tables->num_items < 1)
        return slk_fail(ctx, SLK_EINVAL, "slk_bias_shadow_begin: row-sparse Adagrad over a plain item table only");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = slk_ensure(ctx, ctx->bias_shadow, (size_t)tables->num_items * 8))) return rc;
    hipLaunchKernelGGL(k_bias_shadow_pack, dim3(slk_grid_for(ctx, (size_t)tables->num_items, 256)), dim3(256), 0, s,
                       (const float *)tables->d_param[3], (const float *)optim->d_state1[3], (float2 *)ctx->bias_shadow.p,
                       (size_t)tables->num_items);
    SLK_LAUNCH_CHECK(ctx, "k_bias_shadow_pack");
    ctx->shadow_src_p = tables->d_param[3];
    ctx->shadow_src_s = optim->d_state1[3];
    ctx->shadow_rows = tables->num_items;
    ctx->shadow_active = true;
    ctx->last_stream = s;
    return SLK_OK;
}

SLK_EXPORT int slk_bias_shadow_end(slk_ctx *ctx, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!ctx->shadow_active) return SLK_OK;
    // Whatever happens below the scope is CLOSED when this call returns (an error return used to leave it open and
    // every later _begin / predict on that ctx refused); on an error the caller's arrays hold what they held at _begin.
    float *const dst_p = ctx->shadow_src_p, *const dst_s = ctx->shadow_src_s;
    const int64_t rows = ctx->shadow_rows;
    ctx->shadow_active = false;
    ctx->shadow_src_p = ctx->shadow_src_s = nullptr;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bias_shadow_unpack, dim3(slk_grid_for(ctx, (size_t)rows, 256)), dim3(256), 0, s,
                       (const float2 *)ctx->bias_shadow.p, dst_p, dst_s, (size_t)rows);
    SLK_LAUNCH_CHECK(ctx, "k_bias_shadow_unpack");
    ctx->last_stream = s;
    return SLK_OK;
}

// Closes the scope WITHOUT writing back (ABI 11): for a caller whose bias / accumulator arrays no longer exist (the model was
// deleted inside the scope).  What the scope trained is lost; the arrays -- if they still exist -- keep their _begin values.
SLK_EXPORT int slk_bias_shadow_abort(slk_ctx *ctx) {
    if (!ctx) return SLK_EINVAL;
    ctx->shadow_active = false;
    ctx->shadow_src_p = ctx->shadow_src_s = nullptr;
    return SLK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// User-row ping-pong (round 6).  The two ownership passes of a minibatch each need the OTHER side's pre-step rows.  Rounds 1-6
// saved the user side: the user pass wrote every position's pre-step user row into a record (4D bytes per interaction: 13 % of
// the pass's real traffic at C2) and the item pass gathered it from there.  With 288 GB of HBM the user table can simply exist
// twice: a user's updated row is written to the copy that does NOT hold its current row, one byte per user says which copy is
// current, and the pre-step row stays where it was for the item pass to gather -- the same 4D-byte random read the record cost,
// and no record write.  The caller's table is the union of both copies until _end copies the rows whose current copy is the
// ctx's back.  Same arithmetic in the same order: tables bit-identical to training without it.  The reference has no
// counterpart (autograd keeps the pre-step rows alive as saved tensors: spotlight/factorization/representations.py:61-91).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pingpong_merge(float *P0, const float *alt, const uint8_t *flag, size_t rows, int D) {
    if (D % 4 == 0) {
        const size_t per = (size_t)D / 4, n = rows * per;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
            if (flag[e / per]) reinterpret_cast<float4 *>(P0)[e] = reinterpret_cast<const float4 *>(alt)[e];
    } else {
        const size_t n = rows * (size_t)D;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
            if (flag[e / (size_t)D]) P0[e] = alt[e];
    }
}

SLK_EXPORT int slk_user_pingpong_begin(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!tables || !optim) return slk_fail(ctx, SLK_EINVAL, "slk_user_pingpong_begin: NULL tables / optim");
    if (ctx->pp_active) return slk_fail(ctx, SLK_EINVAL, "slk_user_pingpong_begin: a ping-pong is already active on this ctx");
    if (tables->user_bloom || tables->item_bloom || !tables->d_param[0] ||
        (optim->kind != SLK_OPT_ADAGRAD && optim->kind != SLK_OPT_SPARSE_ADAM && optim->kind != SLK_OPT_SGD))
        return slk_fail(ctx, SLK_EINVAL, "slk_user_pingpong_begin: plain tables and a row-sparse optimizer (Adagrad, SparseAdam, SGD) only");
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)tables->num_users;
    if ((rc = slk_ensure(ctx, ctx->pp_alt, rows * (size_t)tables->dim * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->pp_flags, (rows + 15) / 16 * 16))) return rc;
    SLK_HIP(ctx, hipMemsetAsync(ctx->pp_flags.p, 0, (rows + 15) / 16 * 16, s));  // every current row is the caller's
    ctx->pp_src_u = tables->d_param[0];
    ctx->pp_rows = tables->num_users;
    ctx->pp_dim = tables->dim;
    ctx->pp_active = true;
    ctx->last_stream = s;
    return SLK_OK;
}

SLK_EXPORT int slk_user_pingpong_end(slk_ctx *ctx, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!ctx->pp_active) return SLK_OK;
    // closed on every return (as slk_bias_shadow_end): a failed launch leaves the caller's table a mix of rows, and says so
    float *const dst = ctx->pp_src_u;
    const int64_t rows = ctx->pp_rows;
    const int D = ctx->pp_dim;
    ctx->pp_active = false;
    ctx->pp_src_u = nullptr;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pingpong_merge, dim3(slk_grid_for(ctx, (size_t)rows * (size_t)((D + 3) / 4), 256)), dim3(256), 0, s, dst,
                       (const float *)ctx->pp_alt.p, (const uint8_t *)ctx->pp_flags.p, (size_t)rows, D);
    SLK_LAUNCH_CHECK(ctx, "k_pingpong_merge");
    ctx->last_stream = s;
    return SLK_OK;
}

SLK_EXPORT int slk_user_pingpong_abort(slk_ctx *ctx) {
    if (!ctx) return SLK_EINVAL;
    ctx->pp_active = false;
    ctx->pp_src_u = nullptr;
    return SLK_OK;
}
