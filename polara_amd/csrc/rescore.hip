// K3 (second half): exact fp64 re-scoring of the fp32 candidates, final ordering, and the exact
// brute-force path for the (rare) users the candidate pass cannot certify.
//
// Why: the reference scores in fp64 (models.py:860) and returns index lists, and the north-star
// contract is "top-k index sets identical".  fp32 MFMA scores can flip near-ties at the k-th
// boundary, so the candidate kernel keeps KC > topk items per user and this pass (a) recomputes
// those KC scores in fp64 from the fp64 factors, (b) orders them (score desc, item asc), and
// (c) certifies the result: every non-candidate has fp32 score <= tau32 (the KC-th candidate), so
// its fp64 score is <= tau32 + err; if the k-th exact score clears that bound the top-k set is
// provably the exact one, else the user is flagged and re-done by score_exact_rows (all items,
// fp64, two-class key = the reference's downvote_seen_items semantics, models.py:510-519).
#include "pk_common.h"
#include <math.h>
#include <stdlib.h>

#define PK_IDX_NONE 0x7fffffff

__device__ __forceinline__ bool pk_before64(double ka, int va, double kb, int vb) {
    return (ka > kb) || (ka == kb && va < vb);
}

// bitonic sort (descending by pk_before64) inside aligned segments of SEG lanes, one (key, val) per lane
// LPC: lanes per element (the LPC adjacent lanes of a candidate hold identical copies and perform identical
// exchanges, so the network runs on all lanes with partners LPC * J lanes away)
template <int SEG, int K, int J, int LPC>
__device__ __forceinline__ void pk_bitonic_seg_merge(double &key, int &val, int t) {
    const double ok = pk_lane_xor<J * LPC>(key);
    const int ov = pk_lane_xor<J * LPC>(val);
    const bool lower = (t & J) == 0;
    const bool desc = (t & K) == 0;   // t < SEG: the last level (K == SEG) is descending everywhere
    const bool want_first = (lower == desc);
    const bool other_first = pk_before64(ok, ov, key, val);
    if (want_first == other_first) {
        key = ok;
        val = ov;
    }
    if constexpr (J > 1) pk_bitonic_seg_merge<SEG, K, (J >> 1), LPC>(key, val, t);
}
template <int SEG, int K, int LPC>
__device__ __forceinline__ void pk_bitonic_seg_levels(double &key, int &val, int t) {
    if constexpr (K > 2) pk_bitonic_seg_levels<SEG, (K >> 1), LPC>(key, val, t);
    pk_bitonic_seg_merge<SEG, K, (K >> 1), LPC>(key, val, t);
}
template <int SEG, int LPC>
__device__ __forceinline__ void pk_bitonic_seg(double &key, int &val, int t) {
    pk_bitonic_seg_levels<SEG, SEG, LPC>(key, val, t);
}

// ---- the ONE summation order of every exact fp64 score in this file ---------------------------------
// The K products are dealt to four chains by element PAIR: pair p = (2p, 2p+1) goes to chain p & 3, each
// chain is a serial fma chain in increasing p, and the total is (c0 + c2) + (c1 + c3).  A thread working
// alone (LPC = 1, and the exact-row kernel) keeps four accumulators; with LPC = 2 lane q owns chains q and
// q + 2; with LPC = 4 lane q owns chain {0, 2, 1, 3}[q], so that the lane^1 exchange forms (c0 + c2) and
// (c1 + c3) and the lane^2 exchange the total.  Whatever the lane layout — and whether a user goes
// through the re-scoring kernel or the exact-row kernel — a given (user, item) score has the same bits.
template <int LPC>
__device__ __forceinline__ double pk_dot_chains(const double *__restrict__ vr, const double *er, int K, int q,
                                                bool vvec2, bool evec2, double *e_norm2 = nullptr) {
    double n2 = 0.0;   // sum of squares of the e elements this lane touches (any order: only feeds a bound)
    constexpr int NLOC = 4 / LPC;
    double acc[NLOC];
#pragma unroll
    for (int j = 0; j < NLOC; ++j) acc[j] = 0.0;
    const int np = (K + 1) >> 1, nfull = K >> 1;
    const int start = (LPC == 4) ? (((q & 1) << 1) | (q >> 1)) : q;
    int p = start;
    // all NLOC pairs of an iteration are complete: no bounds tests, loads of the iteration issued together
    if (vvec2 && evec2) {
        const double2 *v2 = reinterpret_cast<const double2 *>(vr);
        const double2 *e2 = reinterpret_cast<const double2 *>(er);
#pragma unroll 2
        for (; p + (NLOC - 1) * LPC < nfull; p += 4) {
            double2 a[NLOC], e[NLOC];
#pragma unroll
            for (int j = 0; j < NLOC; ++j) {
                a[j] = v2[p + j * LPC];
                e[j] = e2[p + j * LPC];
            }
#pragma unroll
            for (int j = 0; j < NLOC; ++j) {
                acc[j] = fma(e[j].x, a[j].x, acc[j]);
                acc[j] = fma(e[j].y, a[j].y, acc[j]);
                n2 = fma(e[j].x, e[j].x, fma(e[j].y, e[j].y, n2));
            }
        }
    } else {
        for (; p + (NLOC - 1) * LPC < nfull; p += 4) {
#pragma unroll
            for (int j = 0; j < NLOC; ++j) {
                const int pj = p + j * LPC;
                const double e0 = er[2 * pj], e1 = er[2 * pj + 1];
                acc[j] = fma(e0, vr[2 * pj], acc[j]);
                acc[j] = fma(e1, vr[2 * pj + 1], acc[j]);
                n2 = fma(e0, e0, fma(e1, e1, n2));
            }
        }
    }
    // tail of the row (at most one iteration), incl. the half-filled last pair of an odd K
#pragma unroll
    for (int j = 0; j < NLOC; ++j) {
        const int pj = p + j * LPC;
        if (pj < nfull) {
            const double e0 = er[2 * pj], e1 = er[2 * pj + 1];
            acc[j] = fma(e0, vr[2 * pj], acc[j]);
            acc[j] = fma(e1, vr[2 * pj + 1], acc[j]);
            n2 = fma(e0, e0, fma(e1, e1, n2));
        } else if (pj < np) {
            const double e0 = er[2 * pj];
            acc[j] = fma(e0, vr[2 * pj], acc[j]);
            n2 = fma(e0, e0, n2);
        }
    }
    if (e_norm2) {
        if constexpr (LPC >= 2) n2 += pk_lane_xor<1>(n2);
        if constexpr (LPC >= 4) n2 += pk_lane_xor<2>(n2);
        *e_norm2 = n2;
    }
    double s;
    if constexpr (LPC == 1) s = (acc[0] + acc[2]) + (acc[1] + acc[3]);
    else if constexpr (LPC == 2) s = acc[0] + acc[1];
    else s = acc[0];
    if constexpr (LPC >= 2) s += pk_lane_xor<1>(s);
    if constexpr (LPC >= 4) s += pk_lane_xor<2>(s);
    return s;
}

// Approximate score against the fp32 image of an item row (pk_rescore_topk_rows_f64 with V32_dev): lane q of
// LPC takes the 16-byte pieces (four elements) q, q + LPC, ... of the row, fp64 fma chains, any order — these
// scores are only trusted to within the delta the caller adds for them, never returned as exact ones.
template <int LPC>
__device__ __forceinline__ double pk_dot_f32row(const float *__restrict__ vr, const double *er, int K, int q,
                                                bool vec, double *e_norm2) {
    double a0 = 0.0, a1 = 0.0, n2 = 0.0;
    const int nq = K >> 2;
    if (vec) {
        const float4 *v4 = reinterpret_cast<const float4 *>(vr);
        const double2 *e2 = reinterpret_cast<const double2 *>(er);
#pragma unroll 2
        for (int p = q; p < nq; p += LPC) {
            const float4 v = v4[p];
            const double2 ea = e2[2 * p], eb = e2[2 * p + 1];
            a0 = fma(ea.x, (double)v.x, a0);
            a1 = fma(ea.y, (double)v.y, a1);
            a0 = fma(eb.x, (double)v.z, a0);
            a1 = fma(eb.y, (double)v.w, a1);
            n2 = fma(ea.x, ea.x, fma(ea.y, ea.y, fma(eb.x, eb.x, fma(eb.y, eb.y, n2))));
        }
    } else {
        for (int p = q; p < nq; p += LPC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double e = er[4 * p + j];
                a0 = fma(e, (double)vr[4 * p + j], a0);
                n2 = fma(e, e, n2);
            }
        }
    }
    if (q == 0)
        for (int j = 4 * nq; j < K; ++j) {   // the last K mod 4 elements
            const double e = er[j];
            a1 = fma(e, (double)vr[j], a1);
            n2 = fma(e, e, n2);
        }
    double s = a0 + a1;
    if constexpr (LPC >= 2) {
        s += pk_lane_xor<1>(s);
        n2 += pk_lane_xor<1>(n2);
    }
    if constexpr (LPC >= 4) {
        s += pk_lane_xor<2>(s);
        n2 += pk_lane_xor<2>(n2);
    }
    *e_norm2 = n2;
    return s;
}

// LPC adjacent lanes own one candidate: they walk the candidate's item row in interleaved 16-byte pieces
// (the candidates of a user are popular items, their rows sit in L2) against the user's E row with serial
// fp64 FMA chains and add their LPC partial sums at the end (one or two lane exchanges); a segment of
// SEG >= KC*splits candidates owns one user, 64/(SEG*LPC) users share a wave.  LPC trades divergent-row
// loads (each instruction touches 64/LPC different rows: the texture path handles them one line at a
// time) against the instructions of the segment sort, which serve fewer users per wave.
// (The first version gave a whole wave to each user and paid a 6-step fp64 wave reduction per candidate
// plus a 64-lane sort: 2.7 ms per 1M users; LPC = 1: 1.04 ms.)
// SCORE4 (segments that fill a wave, SEG * LPC == 64): the scores are computed 16 candidates at a time by FOUR lanes
// each (pk_dot_*<4>: the same bits) and handed to the lanes that sort them — a gather instruction then touches 16 item
// rows of 64 contiguous bytes instead of 64 rows of 16 (LPC = 1, KC = 64) or 32 of 32: the texture path serves one line
// per clock, and at rank 200 / 64 candidates the re-scoring was 11 ms of a 53 ms pass (S-50M shard).
template <int SEG, int LPC, bool SCORE4>
__global__ __launch_bounds__(256) void rescore_topk_kernel(
    int64_t n_rows, const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows_dev, int64_t n_users,
    int64_t n_items, int K, const double *__restrict__ V, int64_t ldv, const float *__restrict__ V32, int64_t ldv32,
    const double *__restrict__ E, int64_t lde, const double *__restrict__ e_err, int64_t e_err_ld, int e_exact,
    const int64_t *__restrict__ seen_ptr, int KC, int splits,
    const float *__restrict__ cand_score, const int32_t *__restrict__ cand_idx, int topk, double vmax,
    int64_t *__restrict__ out_idx, double *__restrict__ out_score, int32_t *__restrict__ flags,
    int32_t *__restrict__ flagged_list, int32_t *__restrict__ flagged_count, int32_t flagged_offset,
    const float *__restrict__ item_norm) {
    constexpr int UPW = 64 / (SEG * LPC);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ul = lane / (SEG * LPC), t = (lane / LPC) % SEG, q = lane % LPC;
    // the r-th row to process is user rows[r] (a list of users to re-do) or user r (everybody)
    const int64_t r = ((int64_t)blockIdx.x * 4 + wave) * UPW + ul;
    // the list length may only be known on the device (pk_flag_compact): the grid then covers its capacity
    const int64_t n_eff = n_rows_dev ? ((int64_t)*n_rows_dev < n_rows ? (int64_t)*n_rows_dev : n_rows) : n_rows;
    if (((int64_t)blockIdx.x * 4 + wave) * UPW >= n_eff) return;   // whole wave beyond the list
    const bool live = r < n_eff;
    const int64_t user = live ? (rows ? (int64_t)rows[r] : r) : 0;
    const int64_t urow = user;

    // candidates: the union of the `splits` per-item-range top-KC lists of this user (<= SEG entries);
    // list h of user u lives at ((h * n_pad + u) * KC)
    const int64_t n_pad = ((n_users + 31) / 32) * 32;
    int idx = -1;
    if (live && t < KC * splits) idx = cand_idx[((int64_t)(t / KC) * n_pad + user) * KC + (t % KC)];
    double tau32 = -INFINITY;   // bound on the fp32 score of every NON-candidate item
    bool unbounded = false;     // a sweep that started from a bootstrapped threshold ended without a full list (score.hip:
                                // PK_IDX_FLOOR = -2 in the last slot): no bound on the items it left out -> exact path
    for (int h = 0; h < splits; ++h) {
        // a full list may have left items of its range out: they score at most its KC-th entry
        const int64_t last = ((int64_t)h * n_pad + urow) * KC + KC - 1;
        const int li = cand_idx[last];
        if (li >= 0) tau32 = fmax(tau32, (double)cand_score[last]);
        else if (li == -2) unbounded = true;
    }

    const double *vr = V + (int64_t)(idx >= 0 ? idx : 0) * ldv;
    const double *er = E + urow * lde;
    const bool vvec2 = (ldv & 1) == 0 && (((uintptr_t)V) & 15) == 0;
    const bool evec2 = (lde & 1) == 0 && (((uintptr_t)E) & 15) == 0;
    // first pass over an approximate E: the fp32 image of the item rows will do (half the lines per gathered
    // row), its rounding joins delta below; exact E rows (second pass, or no approximation at all): fp64 rows
    const bool use32 = (V32 != nullptr) && (e_err != nullptr) && !e_exact;
    double e2, s;
    if constexpr (SCORE4) {
        static_assert(SEG * LPC == 64, "SCORE4: one user per wave");
        const bool v32vec = evec2 && (ldv32 & 3) == 0 && (((uintptr_t)V32) & 15) == 0;
        const int q4 = lane & 3;
        s = 0.0;
#pragma unroll
        for (int j = 0; j < SEG / 16; ++j) {
            const int c = 16 * j + (lane >> 2);                    // the candidate this lane quad scores in pass j
            const int cidx = __shfl(idx, c * LPC, 64);
            const int64_t row = cidx >= 0 ? cidx : 0;
            const double sj = use32 ? pk_dot_f32row<4>(V32 + row * ldv32, er, K, q4, v32vec, &e2)
                                    : pk_dot_chains<4>(V + row * ldv, er, K, q4, vvec2, evec2, &e2);
            const double got = __shfl(sj, 4 * (t & 15), 64);       // candidate t was scored in pass t / 16
            if ((t >> 4) == j) s = got;
        }
    } else if (use32) {
        const bool v32vec = evec2 && (ldv32 & 3) == 0 && (((uintptr_t)V32) & 15) == 0;
        s = pk_dot_f32row<LPC>(V32 + (int64_t)(idx >= 0 ? idx : 0) * ldv32, er, K, q, v32vec, &e2);
    } else {
        s = pk_dot_chains<LPC>(vr, er, K, q, vvec2, evec2, &e2);
    }
    const double enorm = sqrt(e2);
    double my_s = (idx >= 0) ? s : -INFINITY;
    int my_i = (idx >= 0) ? idx : PK_IDX_NONE;
    pk_bitonic_seg<SEG, LPC>(my_s, my_i, t);

    // certification
    int flag = 0;
    const int64_t n_seen = seen_ptr ? (seen_ptr[urow + 1] - seen_ptr[urow]) : 0;
    const double s_k = __shfl(my_s, (ul * SEG + topk - 1) * LPC, 64);
    // E given only approximately (fold-in against the fp32 image of V, scoring.py): ||E' - E|| <= 2^-24 e_err[u],
    // so every score here is within delta of the exact one and the ORDER is the exact order wherever
    // consecutive scores are more than 2 delta apart (the k-th against the (k+1)-th too)
    // e_exact: the E rows given now ARE exact, but the candidates were selected by a sweep over the approximate
    // ones — only the bound on the non-candidates keeps a delta, there is nothing approximate left to order
    // scored against fl32(V): |sum_j E'_j (fl32(V_ij) - V_ij)| <= 2^-24 ||E'_u|| ||V_i|| on top of that
    // item by item: |s'_i - s_i| <= ||E' - E|| ||V_i|| + 2^-24 ||E'|| ||V_i|| = cu * ||V_i||, so two neighbours of the list
    // are in their exact order once they are further apart than cu (||V_i|| + ||V_i'||) — with the items' OWN norms
    // (item_norm: fp32 upper bounds) where the caller has them, with the catalogue's largest norm otherwise.  The
    // candidates of a user are rarely the catalogue's heaviest rows: their own norms certify about twice as many
    // users as vmax does, which matters since the packed fold-in image (csrc/foldq.hip) made cu ~40 times larger.
    const double cu = (e_err ? e_err[urow * e_err_ld] * 5.9604644775390625e-08 * (1.0 + 1e-6) : 0.0) +
                      (use32 ? enorm * 5.9604644775390625e-08 * (1.0 + 1e-6) : 0.0);
    const double delta = cu * vmax;
    double d_k = delta;              // the k-th entry's own error bound (the threshold test below)
    if (delta > 0.0 && !e_exact) {
        double d_me = delta;
        if (item_norm && my_i != PK_IDX_NONE && my_i >= 0) d_me = cu * fmin(vmax, (double)item_norm[my_i] * (1.0 + 1e-6));
        const int nxt_lane = lane + LPC;
        const double s_next = (t + 1 < SEG) ? __shfl(my_s, nxt_lane & 63, 64) : -INFINITY;
        const double d_next = (t + 1 < SEG) ? __shfl(d_me, nxt_lane & 63, 64) : delta;
        const bool close = (t < topk) && !(my_s - s_next > d_me + d_next);
        const unsigned long long cb = __ballot(close);
        const unsigned long long seg_mask = (SEG * LPC == 64) ? ~0ull : (((1ull << (SEG * LPC)) - 1ull) << (ul * SEG * LPC));
        if (cb & seg_mask) flag |= 4;
        d_k = __shfl(d_me, (ul * SEG + topk - 1) * LPC, 64);
    }
    if (n_items - n_seen < topk) {
        flag |= 2;  // seen items must re-enter the list: exact path
    } else if (unbounded) {
        flag |= 1;
    } else if (tau32 > -INFINITY) {
        // the sweep's split-bf16 product (score.hip): |s32 - e.v| <= (3 * 2^-16 + (4 K + 10) * 2^-23) |e||v| — operands
        // split into two bf16 each, the lo.lo term dropped, fp32 conversion of the inputs, accumulation roundings
        const double bound = (3.0 * 1.52587890625e-05 + (double)(4 * K + 10) * 1.1920928955078125e-07) * enorm * vmax;
        // the candidate sweep orders scores that agree to 2^-16 relative arbitrarily (key-only flush sorts,
        // score.hip): a non-candidate may exceed the KC-th candidate by that much
        const double tau_cert = tau32 + fabs(tau32) * 3.0517578125e-05;
        // the k-th entry's own error + what the approximate E can hide of an item the sweep left out (any norm up to vmax)
        const double slack = e_exact ? delta : d_k + delta;
        if (bound > 0.0 && !(s_k - tau_cert > bound + slack))
            flag |= (!e_exact && delta > 0.0 && s_k - tau_cert > bound + delta) ? 4 : 1;
    }
    if (live && q == 0 && t < topk) {
        out_idx[user * topk + t] = (my_i == PK_IDX_NONE) ? -1 : (int64_t)my_i;
        if (out_score) out_score[user * topk + t] = my_s;
    }
    if (live && q == 0 && t == 0) {
        flags[user] = flag;
        // the list of the users to re-do, built where the flag is: what a separate compaction pass over the flags (two
        // more launches per list) did.  The order of the list is arbitrary either way — every listed user is re-done on
        // its own rows — so the atomic counter costs no determinism of the results.
        if (flag && flagged_list) flagged_list[atomicAdd(flagged_count, 1)] = flagged_offset + (int32_t)user;
    }
}

extern "C" int pk_rescore_topk_rows_norms_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                                             const int32_t *n_rows_dev, int64_t n_users,
                                             int64_t n_items, int32_t K, const double *V_dev, int64_t ldv,
                                             const float *V32_dev, int64_t ldv32,
                                             const double *E_dev, int64_t lde, const double *e_err_dev,
                                             int64_t e_err_ld, int32_t e_exact,
                                             const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                                             const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                                             double v_row_norm_max, int64_t *out_idx_dev, double *out_score_dev,
                                             int32_t *flags_dev, int32_t *flagged_list_dev, int32_t *flagged_count_dev,
                                             int32_t flagged_offset, const float *item_norm_dev) {
    PK_REQUIRE((flagged_list_dev == nullptr) == (flagged_count_dev == nullptr), "pk_rescore_topk_f64: flagged list without its counter");
    PK_REQUIRE(n_users >= 1 && K >= 1 && K <= 256 && ldv >= K && lde >= K, "pk_rescore_topk_f64: bad sizes");
    PK_REQUIRE(n_rows >= 0 && n_rows <= n_users, "pk_rescore_topk_f64: bad row count");
    PK_REQUIRE(V32_dev == nullptr || ldv32 >= K, "pk_rescore_topk_f64: ldv32 < K");
    PK_REQUIRE(KC >= 1 && splits >= 1 && KC * splits <= 64 && topk >= 1 && topk <= KC,
               "pk_rescore_topk_f64: need topk <= KC and KC*splits <= 64");
    if (n_rows == 0) return PK_OK;
    const int seg = (KC * splits <= 16) ? 16 : (KC * splits <= 32) ? 32 : 64;
    const int lpc_req = 0;        // lanes per candidate: by the segment width (below)
    const bool score4 = true;     // two-step scoring (four lanes per candidate first)
#define PK_RESCORE(SEGV, LPCV) PK_RESCORE_X(SEGV, LPCV, false)
#define PK_RESCORE_X(SEGV, LPCV, S4)                                                                                    \
    hipLaunchKernelGGL((rescore_topk_kernel<SEGV, LPCV, S4>), dim3((unsigned)pk_ceil_div(n_rows, 4 * (64 / (SEGV * LPCV)))), \
                       dim3(256), 0, pk_stream(stream), n_rows, rows_dev, n_rows_dev, n_users, n_items, K, V_dev, ldv, V32_dev, ldv32, E_dev, lde, \
                       e_err_dev, e_err_ld, e_exact, seen_ptr_dev, KC, splits, cand_score_dev, cand_idx_dev, topk, v_row_norm_max,   \
                       out_idx_dev, out_score_dev, flags_dev, flagged_list_dev, flagged_count_dev, flagged_offset, item_norm_dev)
    if (seg == 16) {
        if (lpc_req == 1) PK_RESCORE(16, 1);
        else if (lpc_req == 4) PK_RESCORE(16, 4);
        else PK_RESCORE(16, 2);    // two users per wave: scoring them one after the other with four lanes per candidate
                                   // was measured 3-8 % SLOWER (0.076 -> 0.083 ms ML-20M-shaped, 0.65 -> 0.67 ms S-1M)
    } else if (seg == 32) {
        if (lpc_req == 1) PK_RESCORE(32, 1);
        else if (score4) PK_RESCORE_X(32, 2, true);
        else PK_RESCORE(32, 2);
    } else {
        if (score4) PK_RESCORE_X(64, 1, true);
        else PK_RESCORE(64, 1);
    }
#undef PK_RESCORE
#undef PK_RESCORE_X
    PK_CHECK_LAUNCH("rescore_topk_kernel");
    return PK_OK;
}

extern "C" int pk_rescore_topk_rows_list_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                                             const int32_t *n_rows_dev, int64_t n_users,
                                             int64_t n_items, int32_t K, const double *V_dev, int64_t ldv,
                                             const float *V32_dev, int64_t ldv32,
                                             const double *E_dev, int64_t lde, const double *e_err_dev,
                                             int64_t e_err_ld, int32_t e_exact,
                                             const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                                             const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                                             double v_row_norm_max, int64_t *out_idx_dev, double *out_score_dev,
                                             int32_t *flags_dev, int32_t *flagged_list_dev, int32_t *flagged_count_dev,
                                             int32_t flagged_offset) {
    return pk_rescore_topk_rows_norms_f64(stream, n_rows, rows_dev, n_rows_dev, n_users, n_items, K, V_dev, ldv, V32_dev, ldv32, E_dev,
                                          lde, e_err_dev, e_err_ld, e_exact, seen_ptr_dev, KC, splits, cand_score_dev, cand_idx_dev, topk,
                                          v_row_norm_max, out_idx_dev, out_score_dev, flags_dev, flagged_list_dev, flagged_count_dev,
                                          flagged_offset, nullptr);
}

extern "C" int pk_rescore_topk_rows_f64(void *stream, int64_t n_rows, const int32_t *rows_dev,
                                        const int32_t *n_rows_dev, int64_t n_users,
                                        int64_t n_items, int32_t K, const double *V_dev, int64_t ldv,
                                        const float *V32_dev, int64_t ldv32,
                                        const double *E_dev, int64_t lde, const double *e_err_dev,
                                        int64_t e_err_ld, int32_t e_exact,
                                        const int64_t *seen_ptr_dev, int32_t KC, int32_t splits,
                                        const float *cand_score_dev, const int32_t *cand_idx_dev, int32_t topk,
                                        double v_row_norm_max, int64_t *out_idx_dev, double *out_score_dev,
                                        int32_t *flags_dev) {
    return pk_rescore_topk_rows_list_f64(stream, n_rows, rows_dev, n_rows_dev, n_users, n_items, K, V_dev, ldv, V32_dev, ldv32, E_dev,
                                         lde, e_err_dev, e_err_ld, e_exact, seen_ptr_dev, KC, splits, cand_score_dev, cand_idx_dev, topk,
                                         v_row_norm_max, out_idx_dev, out_score_dev, flags_dev, nullptr, nullptr, 0);
}

extern "C" int pk_rescore_topk_f64(void *stream, int64_t n_users, int64_t n_items, int32_t K, const double *V_dev,
                                   int64_t ldv, const double *E_dev, int64_t lde, const int64_t *seen_ptr_dev,
                                   int32_t KC, int32_t splits, const float *cand_score_dev,
                                   const int32_t *cand_idx_dev,
                                   int32_t topk, double v_row_norm_max, int64_t *out_idx_dev,
                                   double *out_score_dev, int32_t *flags_dev) {
    return pk_rescore_topk_rows_f64(stream, n_users, nullptr, nullptr, n_users, n_items, K, V_dev, ldv, nullptr, 0, E_dev, lde,
                                    nullptr, 0, 0, seen_ptr_dev, KC, splits, cand_score_dev, cand_idx_dev, topk, v_row_norm_max,
                                    out_idx_dev, out_score_dev, flags_dev);
}

// ---- re-doing flagged users without the host -------------------------------------------------------
// list[0 .. count) = users u in [0, n) with (flags[u] & mask) != 0, in no particular order (count is zeroed by
// a one-thread kernel in pk_flag_compact).  cap = n: the list cannot overflow.
__global__ __launch_bounds__(256) void flag_compact_kernel(int64_t n, const int32_t *__restrict__ flags, int mask,
                                                           int32_t *__restrict__ list, int32_t *__restrict__ count) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = u < n && (flags[u] & mask) != 0;
    // one atomic per wave: lanes take consecutive slots behind the wave's base
    const unsigned long long b = __ballot(hit);
    if (b == 0ull) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __builtin_ctzll(b)) base = atomicAdd(count, __popcll(b));
    base = __shfl(base, __builtin_ctzll(b), 64);
    if (hit) list[base + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)u;
}

__global__ void zero_i32_kernel(int32_t *p) { *p = 0; }

__global__ void zero_i32_n_kernel(int32_t *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
// zeroes n device counters with a kernel (not a memset node: see pk_flag_compact); one workgroup loops over them — a
// handful per pass, a few thousand when a caller asks for that many user batches
extern "C" int pk_zero_i32(void *stream, int32_t *p_dev, int32_t n) {
    PK_REQUIRE(p_dev && n >= 1 && n <= (1 << 20), "pk_zero_i32: bad arguments");
    hipLaunchKernelGGL(zero_i32_n_kernel, dim3(1), dim3(64), 0, pk_stream(stream), p_dev, n);
    PK_CHECK_LAUNCH("zero_i32_n_kernel");
    return PK_OK;
}

extern "C" int pk_flag_compact(void *stream, int64_t n, const int32_t *flags_dev, int32_t mask, int32_t *list_dev,
                               int32_t *count_dev) {
    PK_REQUIRE(n >= 1 && flags_dev && list_dev && count_dev, "pk_flag_compact: bad arguments");
    hipStream_t st = pk_stream(stream);
    // the counter is zeroed by a kernel, not by hipMemsetAsync: a 4-byte memset NODE of a captured hipGraph faults on
    // replay once another kernel has run in between (ROCm 7.2, MI355X; tools/probes/graph_debug4.py bisects the pass
    // to this call) — and a scoring pass must stay capturable (scoring.CapturedPass)
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(1), 0, st, count_dev);
    hipLaunchKernelGGL(flag_compact_kernel, dim3((unsigned)pk_ceil_div(n, 256)), dim3(256), 0, st, n, flags_dev, mask,
                       list_dev, count_dev);
    PK_CHECK_LAUNCH("flag_compact_kernel");
    return PK_OK;
}

// E[row_offset + list[r], 0:K] = sum_p vals[p] * V[indices[p], 0:K] in fp64 for the listed rows of a CSR.
// A fixed grid of workgroups strides over the list (its length is only known on the device); the sixteen waves
// of a workgroup take every sixteenth 64-pair chunk of the row, eight independent row gathers in flight each
// (the rows that need re-folding are the long ones — thousands of entries — and the kernel lasts as long as the longest:
// with four waves 54 us for 140 users of an ML-20M-shaped pass), and their partial sums are added in wave order.
#define PK_FOLD_BLOCKS 2048
#define PK_FOLD_WAVES 16
template <typename VT, int CPL>
__global__ __launch_bounds__(64 * PK_FOLD_WAVES) void fold_rows_kernel(int64_t cap, const int32_t *__restrict__ list,
                                                        const int32_t *__restrict__ count, int64_t row_offset,
                                                        const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const VT *__restrict__ vals, const double *__restrict__ V,
                                                        int64_t ldv, int K, double *__restrict__ E, int64_t lde) {
    __shared__ double part[PK_FOLD_WAVES][64 * CPL];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n = (*count < cap) ? *count : cap;
    int col[CPL];
#pragma unroll
    for (int g = 0; g < CPL; ++g) {
        const int c = lane + 64 * g;
        col[g] = c < K ? c : K - 1;     // clamp: out-of-range lanes read a valid column, discarded at the end
    }
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const int64_t row = row_offset + list[r];
        const int64_t p0 = indptr[row], p1 = indptr[row + 1];
        double acc[CPL];
#pragma unroll
        for (int g = 0; g < CPL; ++g) acc[g] = 0.0;
        for (int64_t base = p0 + 64 * wave; base < p1; base += 64 * PK_FOLD_WAVES) {
            int j = 0;
            double a = 0.0;             // padded lanes hold (0, 0.0): they add 0 * V[0, :]
            if (base + lane < p1) {
                j = indices[base + lane];
                a = (double)vals[base + lane];
            }
            const int cnt = (int)((p1 - base < 64) ? (p1 - base) : 64);
            for (int t0 = 0; t0 < cnt; t0 += 8) {
                double x[8][CPL];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jt = __builtin_amdgcn_readlane(j, (t0 + u) & 63);
                    const double *vr = V + (int64_t)jt * ldv;
#pragma unroll
                    for (int g = 0; g < CPL; ++g) x[u][g] = vr[col[g]];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(a), (t0 + u) & 63);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(a), (t0 + u) & 63);
                    const double at = __hiloint2double(hi, lo);
#pragma unroll
                    for (int g = 0; g < CPL; ++g) acc[g] = fma(at, x[u][g], acc[g]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < CPL; ++g) part[wave][lane + 64 * g] = acc[g];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int g = 0; g < CPL; ++g) {
                const int c = lane + 64 * g;
                if (c < K) {
                    double tot = part[0][c];
#pragma unroll
                    for (int w = 1; w < PK_FOLD_WAVES; ++w) tot += part[w][c];
                    E[row * lde + c] = tot;
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int pk_fold_rows_f64(void *stream, int64_t cap, const int32_t *list_dev, const int32_t *count_dev,
                                int64_t row_offset, const int64_t *indptr_dev, const int32_t *indices_dev,
                                const void *vals_dev, int val_kind, const double *V_dev, int64_t ldv, int32_t K,
                                double *E_dev, int64_t lde) {
    PK_REQUIRE(cap >= 1 && K >= 1 && K <= 256 && ldv >= K && lde >= K, "pk_fold_rows_f64: bad sizes");
    // indices_dev / vals_dev may be NULL for a matrix without entries (never dereferenced then)
    PK_REQUIRE(list_dev && count_dev && indptr_dev, "pk_fold_rows_f64: bad pointers");
    dim3 grid((unsigned)(cap < PK_FOLD_BLOCKS ? cap : PK_FOLD_BLOCKS)), block(64 * PK_FOLD_WAVES);
    PK_REQUIRE(val_kind == PK_VAL_F32 || val_kind == PK_VAL_F64, "pk_fold_rows_f64: bad val_kind %d", val_kind);
    const int cpl = (K + 63) / 64;
#define PK_FOLD(VT, C)                                                                                           \
    hipLaunchKernelGGL((fold_rows_kernel<VT, C>), grid, block, 0, pk_stream(stream), cap, list_dev, count_dev,   \
                       row_offset, indptr_dev, indices_dev, static_cast<const VT *>(vals_dev), V_dev, ldv, K,    \
                       E_dev, lde)
#define PK_FOLD_C(VT)                                                                                            \
    switch (cpl) {                                                                                               \
        case 1: PK_FOLD(VT, 1); break;                                                                           \
        case 2: PK_FOLD(VT, 2); break;                                                                           \
        case 3: PK_FOLD(VT, 3); break;                                                                           \
        default: PK_FOLD(VT, 4); break;                                                                          \
    }
    if (val_kind == PK_VAL_F32) { PK_FOLD_C(float) } else { PK_FOLD_C(double) }
#undef PK_FOLD_C
#undef PK_FOLD
    PK_CHECK_LAUNCH("fold_rows_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// exact rows: one workgroup per listed user
// ------------------------------------------------------------------------------------------
#define PK_EXACT_CHUNK 1024
#define PK_EXACT_TOPK_MAX 256
#define PK_EXACT_CHUNKS_MAX 8192
#define PK_EXACT_FAST_ROWS 1024
static int64_t exact_per_row(int64_t n_items) {
    // one-workgroup kernel: fp64 score + 1 class byte per item; chunk kernels: per chunk of PK_EXACT_CHUNK items at most
    // min(PK_EXACT_TOPK_MAX, n_items) candidates of 13 bytes (score, index, class); padded to 16 bytes
    const int64_t slow = n_items * 8 + ((n_items + 15) / 16) * 16;
    const int64_t kmax = n_items < PK_EXACT_TOPK_MAX ? n_items : PK_EXACT_TOPK_MAX;
    const int64_t fast = ((pk_ceil_div(n_items, PK_EXACT_CHUNK) * kmax * 13 + 15) / 16) * 16;
    return slow > fast ? slow : fast;
}
extern "C" int64_t pk_exact_work_bytes(int32_t n_rows, int64_t n_items) { return (int64_t)n_rows * exact_per_row(n_items); }

struct Best {
    int cls;  // 0 = unseen (ranks first), 1 = seen, 2 = taken / none
    double s;
    int idx;
};
__device__ __forceinline__ bool best_before(const Best &a, const Best &b) {
    if (a.cls != b.cls) return a.cls < b.cls;
    if (a.s != b.s) return a.s > b.s;
    return a.idx < b.idx;
}

// Every workgroup walks the list from `first_row` with stride gridDim.x (one work slice per workgroup): the number of
// listed users may live on the device (n_rows_dev), so that the caller never has to read it back before launching.
// by_user: results go to row `user` of the outputs instead of row r of the list.
__global__ __launch_bounds__(256) void score_exact_rows_kernel(
    int32_t n_rows_host, const int32_t *__restrict__ n_rows_dev, int32_t first_row, const int32_t *__restrict__ rows,
    int by_user, int64_t n_items, int K, const double *__restrict__ V, int64_t ldv,
    const double *__restrict__ E, int64_t lde, const int64_t *__restrict__ seen_ptr,
    const int32_t *__restrict__ seen_idx, int topk, int64_t *__restrict__ out_idx,
    double *__restrict__ out_score, unsigned char *__restrict__ work, int64_t per_row) {
    extern __shared__ double s_e[];   // the user's E row: K doubles (dynamic: any rank fits — the fused sweep stops at 256)
    __shared__ int s_cls[256];
    __shared__ double s_s[256];
    __shared__ int s_i[256];
    const int tid = threadIdx.x;
    const int32_t n_rows = n_rows_dev ? *n_rows_dev : n_rows_host;
    double *score = reinterpret_cast<double *>(work + (int64_t)blockIdx.x * per_row);
    unsigned char *cls = work + (int64_t)blockIdx.x * per_row + n_items * 8;
  for (int32_t r = first_row + blockIdx.x; r < n_rows; r += gridDim.x) {
    const int64_t user = rows[r];
    const int64_t orow = by_user ? user : (int64_t)r;
    __syncthreads();   // the previous user's s_e / cls are no longer read

    for (int c = tid; c < K; c += 256) s_e[c] = E[user * lde + c];
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += 256) {
        const double *vr = V + i * ldv;
        score[i] = pk_dot_chains<1>(vr, s_e, K, 0, false, false);   // same bits as the re-scoring kernel
        cls[i] = 0;
    }
    __syncthreads();
    if (seen_ptr) {
        const int64_t p0 = seen_ptr[user], p1 = seen_ptr[user + 1];
        for (int64_t p = p0 + tid; p < p1; p += 256) {
            const int j = seen_idx[p];
            if (j >= 0 && j < n_items) cls[j] = 1;
        }
    }
    __syncthreads();
    for (int t = 0; t < topk; ++t) {
        Best b;
        b.cls = 2;
        b.s = -INFINITY;
        b.idx = PK_IDX_NONE;
        for (int64_t i = tid; i < n_items; i += 256) {
            Best c;
            c.cls = cls[i];
            c.s = score[i];
            c.idx = (int)i;
            if (c.cls < 2 && best_before(c, b)) b = c;
        }
        s_cls[tid] = b.cls;
        s_s[tid] = b.s;
        s_i[tid] = b.idx;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                Best x{s_cls[tid], s_s[tid], s_i[tid]}, y{s_cls[tid + st], s_s[tid + st], s_i[tid + st]};
                if (best_before(y, x)) {
                    s_cls[tid] = y.cls;
                    s_s[tid] = y.s;
                    s_i[tid] = y.idx;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const bool ok = s_cls[0] < 2;
            out_idx[orow * topk + t] = ok ? (int64_t)s_i[0] : -1;
            if (out_score) out_score[orow * topk + t] = ok ? s_s[0] : -INFINITY;
            if (ok) cls[s_i[0]] = 2;
        }
        __syncthreads();
    }
  }
}

// ---- the same result from the whole chip -------------------------------------------------------------------------
// One workgroup per listed user streams the whole of V through one CU (and reads a V row per THREAD: 64 cache lines per
// load instruction) and then scans the n_items scores once per list position: 7.5 ms per user at 500K items x rank 200 —
// eleven flagged users of a million took longer than the rest of the pass (82 of 135 ms, S-50M shard).  The fast path
// cuts the catalogue into chunks of PK_EXACT_CHUNK items; a workgroup owns (chunk, a slice of the listed users): four
// lanes per item compute the score (pk_dot_chains<4>: the bits of every other exact score in this file), the user's
// seen items falling into the chunk are marked, and the chunk's best min(topk, chunk) entries — same total order:
// class, score descending, index ascending — go to the work buffer in that order; a V chunk (1.6 MB at rank 200) stays
// in L2 while the listed users take their turns.  A second kernel, one workgroup per user, merges the sorted chunk
// lists by their heads (a cursor per chunk in LDS).  Users beyond the work buffer's row slots, lists longer than 256 and
// catalogues beyond 8 M items take the one-workgroup kernel above.
struct ExactLayout {   // candidate arrays of one row slot inside its per_row bytes of the work buffer
    int64_t idx_off, cls_off;
};
static inline ExactLayout exact_layout(int64_t n_chunks, int ksel) {
    ExactLayout l;
    l.idx_off = n_chunks * ksel * 8;
    l.cls_off = l.idx_off + n_chunks * ksel * 4;
    return l;
}

__device__ __forceinline__ Best best_wave_min(Best b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.cls = __shfl_xor(b.cls, off, 64);
        o.s = __shfl_xor(b.s, off, 64);
        o.idx = __shfl_xor(b.idx, off, 64);
        if (best_before(o, b)) b = o;
    }
    return b;
}

__global__ __launch_bounds__(256) void exact_chunk_kernel(
    int32_t n_rows_host, const int32_t *__restrict__ n_rows_dev, int32_t row_slots, const int32_t *__restrict__ rows,
    int64_t n_items, int K, const double *__restrict__ V, int64_t ldv, const double *__restrict__ E, int64_t lde,
    const int64_t *__restrict__ seen_ptr, const int32_t *__restrict__ seen_idx, int ksel,
    unsigned char *__restrict__ work, int64_t per_row, int64_t idx_off, int64_t cls_off) {
    extern __shared__ __attribute__((aligned(16))) double s_e[];   // K doubles
    __shared__ double s_score[PK_EXACT_CHUNK];
    __shared__ unsigned char s_cls[PK_EXACT_CHUNK];
    __shared__ int s_wc[4];
    __shared__ double s_ws[4];
    __shared__ int s_wi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t n_rows = n_rows_dev ? *n_rows_dev : n_rows_host;
    if (n_rows > row_slots) n_rows = row_slots;
    const int64_t chunk = blockIdx.x;
    const int64_t i0 = chunk * PK_EXACT_CHUNK;
    const int cn = (int)((n_items - i0 < PK_EXACT_CHUNK) ? (n_items - i0) : PK_EXACT_CHUNK);
    const bool vvec2 = ((ldv & 1) == 0) && ((((uintptr_t)V) & 15) == 0);
    for (int32_t r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const int64_t user = rows[r];
        __syncthreads();   // the previous user's s_e / s_score / s_cls are no longer read
        for (int c = tid; c < K; c += 256) s_e[c] = E[user * lde + c];
        for (int c = tid; c < PK_EXACT_CHUNK; c += 256) s_cls[c] = (c < cn) ? 0 : 2;
        __syncthreads();
        const int q = tid & 3;
        for (int l0 = 0; l0 < cn; l0 += 64) {
            const int li = l0 + (tid >> 2);
            const int64_t item = (li < cn) ? i0 + li : i0;
            const double sc = pk_dot_chains<4>(V + item * ldv, s_e, K, q, vvec2, true);
            if (q == 0 && li < cn) s_score[li] = sc;
        }
        if (seen_ptr) {
            const int64_t p0 = seen_ptr[user], p1 = seen_ptr[user + 1];
            for (int64_t p = p0 + tid; p < p1; p += 256) {
                const int64_t j = (int64_t)seen_idx[p] - i0;
                if (j >= 0 && j < cn) s_cls[j] = 1;
            }
        }
        __syncthreads();
        unsigned char *base = work + (int64_t)r * per_row;
        double *c_s = reinterpret_cast<double *>(base) + chunk * ksel;
        int32_t *c_i = reinterpret_cast<int32_t *>(base + idx_off) + chunk * ksel;
        unsigned char *c_c = base + cls_off + chunk * ksel;
        for (int t = 0; t < ksel; ++t) {
            Best b;
            b.cls = 2;
            b.s = -INFINITY;
            b.idx = PK_IDX_NONE;
#pragma unroll
            for (int j = 0; j < PK_EXACT_CHUNK / 256; ++j) {
                const int li = tid + 256 * j;
                Best c;
                c.cls = s_cls[li];
                c.s = s_score[li];
                c.idx = (int)(i0 + li);
                if (c.cls < 2 && best_before(c, b)) b = c;
            }
            b = best_wave_min(b);
            if (lane == 0) {
                s_wc[wave] = b.cls;
                s_ws[wave] = b.s;
                s_wi[wave] = b.idx;
            }
            __syncthreads();
            Best w{s_wc[0], s_ws[0], s_wi[0]};
#pragma unroll
            for (int x = 1; x < 4; ++x) {
                Best o{s_wc[x], s_ws[x], s_wi[x]};
                if (best_before(o, w)) w = o;
            }
            if (tid == 0) {
                c_s[t] = w.s;
                c_i[t] = w.idx;
                c_c[t] = (unsigned char)w.cls;
                if (w.cls < 2) s_cls[w.idx - i0] = 2;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void exact_merge_kernel(
    int32_t n_rows_host, const int32_t *__restrict__ n_rows_dev, int32_t row_slots, const int32_t *__restrict__ rows,
    int by_user, int n_chunks, int ksel, int topk, int64_t *__restrict__ out_idx, double *__restrict__ out_score,
    const unsigned char *__restrict__ work, int64_t per_row, int64_t idx_off, int64_t cls_off) {
    extern __shared__ int s_cur[];   // n_chunks cursors into the (sorted) chunk lists
    __shared__ int s_wc[4];
    __shared__ double s_ws[4];
    __shared__ int s_wi[4];
    __shared__ int s_wk[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t n_rows = n_rows_dev ? *n_rows_dev : n_rows_host;
    if (n_rows > row_slots) n_rows = row_slots;
    for (int32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int64_t orow = by_user ? (int64_t)rows[r] : (int64_t)r;
        const unsigned char *base = work + (int64_t)r * per_row;
        const double *c_s = reinterpret_cast<const double *>(base);
        const int32_t *c_i = reinterpret_cast<const int32_t *>(base + idx_off);
        const unsigned char *c_c = base + cls_off;
        __syncthreads();
        for (int c = tid; c < n_chunks; c += 256) s_cur[c] = 0;
        __syncthreads();
        for (int t = 0; t < topk; ++t) {
            Best b;
            b.cls = 2;
            b.s = -INFINITY;
            b.idx = PK_IDX_NONE;
            int bk = -1;
            for (int c = tid; c < n_chunks; c += 256) {
                const int cur = s_cur[c];
                if (cur >= ksel) continue;
                const int64_t e = (int64_t)c * ksel + cur;
                Best h;
                h.cls = c_c[e];
                h.s = c_s[e];
                h.idx = c_i[e];
                if (h.cls < 2 && best_before(h, b)) {
                    b = h;
                    bk = c;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                Best o;
                o.cls = __shfl_xor(b.cls, off, 64);
                o.s = __shfl_xor(b.s, off, 64);
                o.idx = __shfl_xor(b.idx, off, 64);
                const int ok = __shfl_xor(bk, off, 64);
                if (best_before(o, b)) {
                    b = o;
                    bk = ok;
                }
            }
            if (lane == 0) {
                s_wc[wave] = b.cls;
                s_ws[wave] = b.s;
                s_wi[wave] = b.idx;
                s_wk[wave] = bk;
            }
            __syncthreads();
            if (tid == 0) {
                Best w{s_wc[0], s_ws[0], s_wi[0]};
                int wk = s_wk[0];
                for (int x = 1; x < 4; ++x) {
                    Best o{s_wc[x], s_ws[x], s_wi[x]};
                    if (best_before(o, w)) {
                        w = o;
                        wk = s_wk[x];
                    }
                }
                const bool ok = w.cls < 2;
                out_idx[orow * topk + t] = ok ? (int64_t)w.idx : -1;
                if (out_score) out_score[orow * topk + t] = ok ? w.s : -INFINITY;
                if (ok) s_cur[wk] += 1;
            }
            __syncthreads();
        }
    }
}

// rows [0, min(count, row_slots)) through the chunk kernels when they apply, the rest (or everything) through the
// one-workgroup kernel; `count` on the host (n_rows_dev == nullptr) or on the device
static int exact_launch(hipStream_t st, int32_t n_rows_host, const int32_t *n_rows_dev, int32_t row_slots,
                        const int32_t *rows_dev, int by_user, int64_t n_items, int32_t K, const double *V_dev, int64_t ldv,
                        const double *E_dev, int64_t lde, const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev,
                        int32_t topk, int64_t *out_idx_dev, double *out_score_dev, unsigned char *work, int32_t n_wg_slow) {
    const int64_t per_row = exact_per_row(n_items);
    const int64_t n_chunks = pk_ceil_div(n_items, PK_EXACT_CHUNK);
    const bool fast = row_slots > 0 && topk <= PK_EXACT_TOPK_MAX && n_chunks <= PK_EXACT_CHUNKS_MAX && (size_t)K * 8 <= 48 * 1024;
    int32_t first_slow = 0;
    if (fast) {
        const int ksel = (int)(topk < n_items ? topk : n_items);   // <= PK_EXACT_TOPK_MAX: what exact_per_row provides for
        const ExactLayout lay = exact_layout(n_chunks, ksel);
        int64_t gy = 2048 / n_chunks;
        if (gy < 1) gy = 1;
        if (gy > row_slots) gy = row_slots;
        hipLaunchKernelGGL(exact_chunk_kernel, dim3((unsigned)n_chunks, (unsigned)gy), dim3(256), (size_t)K * 8, st, n_rows_host,
                           n_rows_dev, row_slots, rows_dev, n_items, K, V_dev, ldv, E_dev, lde, seen_ptr_dev, seen_idx_dev, ksel,
                           work, per_row, lay.idx_off, lay.cls_off);
        PK_CHECK_LAUNCH("exact_chunk_kernel");
        hipLaunchKernelGGL(exact_merge_kernel, dim3((unsigned)row_slots), dim3(256), (size_t)n_chunks * 4, st, n_rows_host,
                           n_rows_dev, row_slots, rows_dev, by_user, (int)n_chunks, ksel, topk, out_idx_dev, out_score_dev, work,
                           per_row, lay.idx_off, lay.cls_off);
        PK_CHECK_LAUNCH("exact_merge_kernel");
        first_slow = row_slots;
        if (!n_rows_dev && n_rows_host <= row_slots) return PK_OK;
    }
    hipLaunchKernelGGL(score_exact_rows_kernel, dim3((unsigned)n_wg_slow), dim3(256), (size_t)K * 8, st, n_rows_host, n_rows_dev,
                       first_slow, rows_dev, by_user, n_items, K, V_dev, ldv, E_dev, lde, seen_ptr_dev, seen_idx_dev, topk,
                       out_idx_dev, out_score_dev, work, per_row);
    PK_CHECK_LAUNCH("score_exact_rows_kernel");
    return PK_OK;
}

extern "C" int pk_score_exact_rows_f64(void *stream, int32_t n_rows, const int32_t *rows_dev, int64_t n_items,
                                       int32_t K, const double *V_dev, int64_t ldv, const double *E_dev,
                                       int64_t lde, const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev,
                                       int32_t topk, int64_t *out_idx_dev, double *out_score_dev, void *work_dev) {
    PK_REQUIRE(n_rows >= 0 && n_items >= 1 && K >= 1 && K <= 8192 && topk >= 1, "pk_score_exact_rows_f64: bad sizes");
    PK_REQUIRE(ldv >= K && lde >= K && work_dev, "pk_score_exact_rows_f64: bad arguments");
    if (n_rows == 0) return PK_OK;
    // the work buffer has a row slot per listed user (pk_exact_work_bytes(n_rows, n_items)).  The chunk kernels select
    // min(topk, chunk) entries per (user, chunk) one at a time: they win when a FEW users need the whole chip (the flagged
    // users of a pass), and lose by an order of magnitude when every user of a large set comes this way (topk > 52 or
    // rank > 256 for 1e5 users: there the one-workgroup kernel already fills the chip with users)
    const int32_t fast_rows = n_rows <= PK_EXACT_FAST_ROWS ? n_rows : 0;
    return exact_launch(pk_stream(stream), n_rows, nullptr, fast_rows, rows_dev, 0, n_items, K, V_dev, ldv, E_dev, lde,
                        seen_ptr_dev, seen_idx_dev, topk, out_idx_dev, out_score_dev, static_cast<unsigned char *>(work_dev),
                        n_rows);
}

/* The same over a DEVICE-side list (pk_flag_compact): users list_dev[0 .. *count_dev), results written to the
 * rows of those users in the [n_users x topk] outputs; the work buffer has n_wg row slots (work >= pk_exact_work_bytes(n_wg,
 * n_items)): the first n_wg listed users go through the chunk kernels, any further ones share n_wg workgroups of the
 * one-workgroup kernel.  Nothing about the list visits the host, so a scoring pass needs no synchronisation. */
extern "C" int pk_score_exact_list_f64(void *stream, int32_t n_wg, const int32_t *list_dev, const int32_t *count_dev,
                                       int64_t n_items, int32_t K, const double *V_dev, int64_t ldv, const double *E_dev,
                                       int64_t lde, const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev, int32_t topk,
                                       int64_t *out_idx_dev, double *out_score_dev, void *work_dev) {
    PK_REQUIRE(n_wg >= 1 && n_items >= 1 && K >= 1 && K <= 8192 && topk >= 1 && list_dev && count_dev,
               "pk_score_exact_list_f64: bad sizes");
    PK_REQUIRE(ldv >= K && lde >= K && work_dev && out_idx_dev, "pk_score_exact_list_f64: bad arguments");
    return exact_launch(pk_stream(stream), 0, count_dev, n_wg, list_dev, 1, n_items, K, V_dev, ldv, E_dev, lde, seen_ptr_dev,
                        seen_idx_dev, topk, out_idx_dev, out_score_dev, static_cast<unsigned char *>(work_dev), n_wg);
}

// ------------------------------------------------------------------------------------------
// rows of a result back in the caller's order: dst[perm[r], :] = src[r, :] (perm == NULL: a plain copy).  The scoring
// pass groups its users by activity and scatters the lists back at the end (scoring.recommend); `dst` may be device
// memory or MAPPED PINNED HOST memory — the [n_users x topk] array the reference returns on the host
// (models.py:400-405) is then written by this kernel over PCIe and no copy-engine transfer follows the pass.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_rows_i64_kernel(int64_t n_rows, int width, const int64_t *__restrict__ src,
                                                               const int64_t *__restrict__ perm, int64_t *__restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rows * width) return;
    const int64_t r = e / width;
    const int c = (int)(e - r * width);
    const int64_t to = perm ? perm[r] : r;
    dst[to * width + c] = src[e];
}

extern "C" int pk_scatter_rows_i64(void *stream, int64_t n_rows, int32_t width, const int64_t *src_dev, const int64_t *perm_dev,
                                   int64_t *dst) {
    PK_REQUIRE(n_rows >= 0 && width >= 1 && src_dev && dst, "pk_scatter_rows_i64: bad arguments");
    if (n_rows == 0) return PK_OK;
    hipLaunchKernelGGL(scatter_rows_i64_kernel, dim3((unsigned)pk_ceil_div(n_rows * width, 256)), dim3(256), 0, pk_stream(stream),
                       n_rows, width, src_dev, perm_dev, dst);
    PK_CHECK_LAUNCH("scatter_rows_i64_kernel");
    return PK_OK;
}

// internal item positions -> the caller's item ids, on the way out: dst[e] = src[e] >= 0 ? table[src[e]] : -1 (the padding
// of a list shorter than topk stays -1).  The host-side renaming of a [138K x 10] result cost 4 ms in NumPy against a
// 0.9 ms pass (models.get_recommendations); `dst` may be device or mapped pinned host memory.
__global__ __launch_bounds__(256) void map_ids_i64_kernel(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ table,
                                                          int64_t n_table, int64_t *__restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int64_t v = src[e];
    dst[e] = (v >= 0 && v < n_table) ? table[v] : -1;
}

extern "C" int pk_map_ids_i64(void *stream, int64_t n, const int64_t *src_dev, const int64_t *table_dev, int64_t n_table,
                              int64_t *dst) {
    PK_REQUIRE(n >= 0 && n_table >= 0 && src_dev && dst && (table_dev || n_table == 0), "pk_map_ids_i64: bad arguments");
    if (n == 0) return PK_OK;
    hipLaunchKernelGGL(map_ids_i64_kernel, dim3((unsigned)pk_ceil_div(n, 256)), dim3(256), 0, pk_stream(stream), n, src_dev,
                       table_dev, n_table, dst);
    PK_CHECK_LAUNCH("map_ids_i64_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// dense fp64 score rows (slice_recommendations / _user_scores support, models.py:277-291, 857-861)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_scores_kernel(int n_rows, int64_t n_items, int K,
                                                           const double *__restrict__ V, int64_t ldv,
                                                           const double *__restrict__ E, int64_t lde,
                                                           double *__restrict__ out, int64_t ldo) {
    extern __shared__ double s_e[];   // K doubles
    const int r = blockIdx.y;
    for (int c = threadIdx.x; c < K; c += 256) s_e[c] = E[(int64_t)r * lde + c];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    const double *vr = V + i * ldv;
    out[(int64_t)r * ldo + i] = pk_dot_chains<1>(vr, s_e, K, 0, false, false);
}

extern "C" int pk_dense_scores_f64(void *stream, int32_t n_rows, int64_t n_items, int32_t K, const double *V_dev,
                                   int64_t ldv, const double *E_dev, int64_t lde, double *out_dev, int64_t ldo) {
    PK_REQUIRE(n_rows >= 1 && n_rows <= 65535 && n_items >= 1 && K >= 1 && K <= 8192 && ldo >= n_items,
               "pk_dense_scores_f64: bad sizes");
    hipLaunchKernelGGL(dense_scores_kernel, dim3((unsigned)pk_ceil_div(n_items, 256), (unsigned)n_rows), dim3(256), (size_t)K * 8,
                       pk_stream(stream), n_rows, n_items, K, V_dev, ldv, E_dev, lde, out_dev, ldo);
    PK_CHECK_LAUNCH("dense_scores_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// top-k columns of dense score rows: the array form of get_topk_elements / topsort (models.py:488-491, 561-563), for the
// host-array conveniences of the model layer (_user_scores, show_recommendations).  One workgroup per row; position t
// of the list is the best element that comes AFTER position t - 1 in the total order (score descending, column ascending)
// — k block-wide selections over the row, nothing is modified.  NaN scores sort last.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_rows_kernel(int64_t n_cols, const double *__restrict__ scores, int64_t ld, int topk,
                                                        int64_t *__restrict__ out) {
    __shared__ double s_val[4];
    __shared__ long long s_idx[4];
    const double *row = scores + (int64_t)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double prev_s = INFINITY;
    long long prev_i = -1;
    for (int t = 0; t < topk; ++t) {
        double best = -INFINITY;
        long long bi = 0x7fffffffffffffffLL;
        for (int64_t c = threadIdx.x; c < n_cols; c += 256) {
            double v = row[c];
            if (v != v) v = -INFINITY;                                  // NaN last
            const bool after = (v < prev_s) || (v == prev_s && (long long)c > prev_i);
            if (after && (v > best || (v == best && (long long)c < bi))) {
                best = v;
                bi = c;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const long long oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_val[wave] = best;
            s_idx[wave] = bi;
        }
        __syncthreads();
        best = s_val[0];
        bi = s_idx[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) {
                best = s_val[w];
                bi = s_idx[w];
            }
        __syncthreads();
        if (threadIdx.x == 0) out[(int64_t)blockIdx.x * topk + t] = (bi == 0x7fffffffffffffffLL) ? -1 : bi;
        prev_s = best;
        prev_i = bi;
    }
}

extern "C" int pk_topk_rows_f64(void *stream, int64_t n_rows, int64_t n_cols, const double *scores_dev, int64_t ld, int32_t topk,
                                int64_t *out_idx_dev) {
    PK_REQUIRE(n_rows >= 0 && n_cols >= 1 && ld >= n_cols && topk >= 1 && topk <= n_cols && scores_dev && out_idx_dev,
               "pk_topk_rows_f64: bad arguments (1 <= topk <= n_cols)");
    if (n_rows == 0) return PK_OK;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, pk_stream(stream), n_cols, scores_dev, ld, topk,
                       out_idx_dev);
    PK_CHECK_LAUNCH("topk_rows_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// evaluation support (models.py:408-485 consume the [n_users x topk] array): the rank (1-based, 0 = absent) at
// which every holdout item was recommended to its user, straight from the device-resident top-k buffer —
// only this holdout-sized vector travels to the host, not the recommendation array (20 GB at 50M users x top-50).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eval_ranks_kernel(int64_t n_holdout, const int64_t *__restrict__ recs,
                                                         int topk, const int64_t *__restrict__ hold_row,
                                                         const int64_t *__restrict__ hold_item,
                                                         int32_t *__restrict__ rank_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_holdout) return;
    const int64_t *row = recs + hold_row[e] * topk;
    const int64_t item = hold_item[e];
    int rank = 0;
    for (int t = 0; t < topk; ++t)
        if (row[t] == item) {
            rank = t + 1;
            break;
        }
    rank_out[e] = rank;
}

extern "C" int pk_eval_ranks(void *stream, int64_t n_holdout, const int64_t *recs_dev, int32_t topk,
                             const int64_t *hold_row_dev, const int64_t *hold_item_dev, int32_t *rank_out_dev) {
    PK_REQUIRE(n_holdout >= 0 && topk >= 1 && recs_dev && hold_row_dev && hold_item_dev && rank_out_dev,
               "pk_eval_ranks: bad arguments");
    if (n_holdout == 0) return PK_OK;
    hipLaunchKernelGGL(eval_ranks_kernel, dim3((unsigned)pk_ceil_div(n_holdout, 256)), dim3(256), 0, pk_stream(stream),
                       n_holdout, recs_dev, topk, hold_row_dev, hold_item_dev, rank_out_dev);
    PK_CHECK_LAUNCH("eval_ranks_kernel");
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_rescore() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&zero_i32_kernel));
}
