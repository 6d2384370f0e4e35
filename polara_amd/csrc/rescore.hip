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

#define PK_IDX_NONE 0x7fffffff

__device__ __forceinline__ bool pk_before64(double ka, int va, double kb, int vb) {
    return (ka > kb) || (ka == kb && va < vb);
}

// bitonic sort (descending by pk_before64) inside aligned segments of SEG lanes, one (key, val) per lane
template <int SEG, int K, int J>
__device__ __forceinline__ void pk_bitonic_seg_merge(double &key, int &val, int t) {
    const double ok = pk_lane_xor<J>(key);
    const int ov = pk_lane_xor<J>(val);
    const bool lower = (t & J) == 0;
    const bool desc = (t & K) == 0;   // t < SEG: the last level (K == SEG) is descending everywhere
    const bool want_first = (lower == desc);
    const bool other_first = pk_before64(ok, ov, key, val);
    if (want_first == other_first) {
        key = ok;
        val = ov;
    }
    if constexpr (J > 1) pk_bitonic_seg_merge<SEG, K, (J >> 1)>(key, val, t);
}
template <int SEG, int K>
__device__ __forceinline__ void pk_bitonic_seg_levels(double &key, int &val, int t) {
    if constexpr (K > 2) pk_bitonic_seg_levels<SEG, (K >> 1)>(key, val, t);
    pk_bitonic_seg_merge<SEG, K, (K >> 1)>(key, val, t);
}
template <int SEG>
__device__ __forceinline__ void pk_bitonic_seg(double &key, int &val, int t) {
    pk_bitonic_seg_levels<SEG, SEG>(key, val, t);
}

// A LANE owns one candidate: it walks its own item row (the candidates of a user are popular items,
// their rows sit in L2) against the user's E row with a serial fp64 FMA chain — no cross-lane
// reduction at all; a segment of SEG >= KC*splits lanes owns one user, 64/SEG users share a wave.
// (The first version gave a whole wave to each user and paid a 6-step fp64 wave reduction per
// candidate plus a 64-lane sort: 2.7 ms per 1M users, instruction-bound.)
template <int SEG>
__global__ __launch_bounds__(256) void rescore_topk_kernel(
    int64_t n_users, int64_t n_items, int K, const double *__restrict__ V, int64_t ldv,
    const double *__restrict__ E, int64_t lde, const int64_t *__restrict__ seen_ptr, int KC, int splits,
    const float *__restrict__ cand_score, const int32_t *__restrict__ cand_idx, int topk, double vmax,
    int64_t *__restrict__ out_idx, double *__restrict__ out_score, int32_t *__restrict__ flags) {
    constexpr int UPW = 64 / SEG;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ul = lane / SEG, t = lane % SEG;
    const int64_t user = ((int64_t)blockIdx.x * 4 + wave) * UPW + ul;
    const bool live = user < n_users;
    const int64_t urow = live ? user : 0;

    // candidates: the union of the `splits` per-item-range top-KC lists of this user (<= SEG entries);
    // list h of user u lives at ((h * n_pad + u) * KC)
    const int64_t n_pad = ((n_users + 31) / 32) * 32;
    int idx = -1;
    if (live && t < KC * splits) idx = cand_idx[((int64_t)(t / KC) * n_pad + user) * KC + (t % KC)];
    double tau32 = -INFINITY;   // bound on the fp32 score of every NON-candidate item
    for (int h = 0; h < splits; ++h) {
        // a full list may have left items of its range out: they score at most its KC-th entry
        const int64_t last = ((int64_t)h * n_pad + urow) * KC + KC - 1;
        if (cand_idx[last] >= 0) tau32 = fmax(tau32, (double)cand_score[last]);
    }

    const double *vr = V + (int64_t)(idx >= 0 ? idx : 0) * ldv;
    const double *er = E + urow * lde;
    double s = 0.0, e2 = 0.0;
    const bool vec2 = ((ldv | lde) & 1) == 0 && ((((uintptr_t)V) | ((uintptr_t)E)) & 15) == 0;
    int k = 0;
    if (vec2) {
        const double2 *v2 = reinterpret_cast<const double2 *>(vr);
        const double2 *e2p = reinterpret_cast<const double2 *>(er);
#pragma unroll 4
        for (; k + 1 < K; k += 2) {
            const double2 a = v2[k >> 1], b = e2p[k >> 1];
            s = fma(b.x, a.x, s);
            s = fma(b.y, a.y, s);
            e2 = fma(b.x, b.x, e2);
            e2 = fma(b.y, b.y, e2);
        }
    }
    for (; k < K; ++k) {
        const double b = er[k];
        s = fma(b, vr[k], s);
        e2 = fma(b, b, e2);
    }
    const double enorm = sqrt(e2);
    double my_s = (idx >= 0) ? s : -INFINITY;
    int my_i = (idx >= 0) ? idx : PK_IDX_NONE;
    pk_bitonic_seg<SEG>(my_s, my_i, t);

    // certification
    int flag = 0;
    const int64_t n_seen = seen_ptr ? (seen_ptr[urow + 1] - seen_ptr[urow]) : 0;
    const double s_k = __shfl(my_s, ul * SEG + topk - 1, 64);
    if (n_items - n_seen < topk) {
        flag |= 2;  // seen items must re-enter the list: exact path
    } else if (tau32 > -INFINITY) {
        // |fl32(e.v) - e.v| <= (K + 3) u32 |e||v|  (input rounding + K-term fmaf chain), u32 = 2^-24
        const double bound = (double)(K + 3) * 5.9604644775390625e-08 * enorm * vmax;
        // the candidate sweep orders scores that agree to 2^-18 relative arbitrarily (key-only flush sort,
        // score.hip): a non-candidate may exceed the KC-th candidate by that much
        const double tau_cert = tau32 + fabs(tau32) * 7.62939453125e-06;
        if (bound > 0.0 && !(s_k - tau_cert > bound)) flag |= 1;
    }
    if (live && t < topk) {
        out_idx[user * topk + t] = (my_i == PK_IDX_NONE) ? -1 : (int64_t)my_i;
        if (out_score) out_score[user * topk + t] = my_s;
    }
    if (live && t == 0) flags[user] = flag;
}

extern "C" int pk_rescore_topk_f64(void *stream, int64_t n_users, int64_t n_items, int32_t K, const double *V_dev,
                                   int64_t ldv, const double *E_dev, int64_t lde, const int64_t *seen_ptr_dev,
                                   int32_t KC, int32_t splits, const float *cand_score_dev,
                                   const int32_t *cand_idx_dev,
                                   int32_t topk, double v_row_norm_max, int64_t *out_idx_dev,
                                   double *out_score_dev, int32_t *flags_dev) {
    PK_REQUIRE(n_users >= 1 && K >= 1 && K <= 256 && ldv >= K && lde >= K, "pk_rescore_topk_f64: bad sizes");
    PK_REQUIRE(KC >= 1 && splits >= 1 && KC * splits <= 64 && topk >= 1 && topk <= KC,
               "pk_rescore_topk_f64: need topk <= KC and KC*splits <= 64");
    const int seg = (KC * splits <= 16) ? 16 : (KC * splits <= 32) ? 32 : 64;
#define PK_RESCORE(SEGV)                                                                                        \
    hipLaunchKernelGGL((rescore_topk_kernel<SEGV>), dim3((unsigned)pk_ceil_div(n_users, 4 * (64 / SEGV))),      \
                       dim3(256), 0, pk_stream(stream), n_users, n_items, K, V_dev, ldv, E_dev, lde,            \
                       seen_ptr_dev, KC, splits, cand_score_dev, cand_idx_dev, topk, v_row_norm_max,            \
                       out_idx_dev, out_score_dev, flags_dev)
    if (seg == 16) PK_RESCORE(16);
    else if (seg == 32) PK_RESCORE(32);
    else PK_RESCORE(64);
#undef PK_RESCORE
    PK_CHECK_LAUNCH("rescore_topk_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// exact rows: one workgroup per listed user
// ------------------------------------------------------------------------------------------
extern "C" int64_t pk_exact_work_bytes(int32_t n_rows, int64_t n_items) {
    // fp64 score + 1 class byte per item, rows padded to 16 bytes
    const int64_t per_row = n_items * 8 + ((n_items + 15) / 16) * 16;
    return (int64_t)n_rows * per_row;
}

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

__global__ __launch_bounds__(256) void score_exact_rows_kernel(
    const int32_t *__restrict__ rows, int64_t n_items, int K, const double *__restrict__ V, int64_t ldv,
    const double *__restrict__ E, int64_t lde, const int64_t *__restrict__ seen_ptr,
    const int32_t *__restrict__ seen_idx, int topk, int64_t *__restrict__ out_idx,
    double *__restrict__ out_score, unsigned char *__restrict__ work, int64_t per_row) {
    __shared__ double s_e[256];
    __shared__ int s_cls[256];
    __shared__ double s_s[256];
    __shared__ int s_i[256];
    const int tid = threadIdx.x;
    const int64_t user = rows[blockIdx.x];
    double *score = reinterpret_cast<double *>(work + (int64_t)blockIdx.x * per_row);
    unsigned char *cls = work + (int64_t)blockIdx.x * per_row + n_items * 8;

    if (tid < K) s_e[tid] = E[user * lde + tid];
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += 256) {
        const double *vr = V + i * ldv;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma(s_e[k], vr[k], s);
        score[i] = s;
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
            out_idx[(int64_t)blockIdx.x * topk + t] = ok ? (int64_t)s_i[0] : -1;
            if (out_score) out_score[(int64_t)blockIdx.x * topk + t] = ok ? s_s[0] : -INFINITY;
            if (ok) cls[s_i[0]] = 2;
        }
        __syncthreads();
    }
}

extern "C" int pk_score_exact_rows_f64(void *stream, int32_t n_rows, const int32_t *rows_dev, int64_t n_items,
                                       int32_t K, const double *V_dev, int64_t ldv, const double *E_dev,
                                       int64_t lde, const int64_t *seen_ptr_dev, const int32_t *seen_idx_dev,
                                       int32_t topk, int64_t *out_idx_dev, double *out_score_dev, void *work_dev) {
    PK_REQUIRE(n_rows >= 0 && n_items >= 1 && K >= 1 && K <= 256 && topk >= 1, "pk_score_exact_rows_f64: bad sizes");
    PK_REQUIRE(ldv >= K && lde >= K && work_dev, "pk_score_exact_rows_f64: bad arguments");
    if (n_rows == 0) return PK_OK;
    const int64_t per_row = n_items * 8 + ((n_items + 15) / 16) * 16;
    hipLaunchKernelGGL(score_exact_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, pk_stream(stream), rows_dev,
                       n_items, K, V_dev, ldv, E_dev, lde, seen_ptr_dev, seen_idx_dev, topk, out_idx_dev,
                       out_score_dev, static_cast<unsigned char *>(work_dev), per_row);
    PK_CHECK_LAUNCH("score_exact_rows_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// dense fp64 score rows (slice_recommendations / _user_scores support, models.py:277-291, 857-861)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_scores_kernel(int n_rows, int64_t n_items, int K,
                                                           const double *__restrict__ V, int64_t ldv,
                                                           const double *__restrict__ E, int64_t lde,
                                                           double *__restrict__ out, int64_t ldo) {
    __shared__ double s_e[256];
    const int r = blockIdx.y;
    if (threadIdx.x < K) s_e[threadIdx.x] = E[(int64_t)r * lde + threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    const double *vr = V + i * ldv;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = fma(s_e[k], vr[k], s);
    out[(int64_t)r * ldo + i] = s;
}

extern "C" int pk_dense_scores_f64(void *stream, int32_t n_rows, int64_t n_items, int32_t K, const double *V_dev,
                                   int64_t ldv, const double *E_dev, int64_t lde, double *out_dev, int64_t ldo) {
    PK_REQUIRE(n_rows >= 1 && n_rows <= 65535 && n_items >= 1 && K >= 1 && K <= 256 && ldo >= n_items,
               "pk_dense_scores_f64: bad sizes");
    hipLaunchKernelGGL(dense_scores_kernel, dim3((unsigned)pk_ceil_div(n_items, 256), (unsigned)n_rows), dim3(256), 0,
                       pk_stream(stream), n_rows, n_items, K, V_dev, ldv, E_dev, lde, out_dev, ldo);
    PK_CHECK_LAUNCH("dense_scores_kernel");
    return PK_OK;
}
