// evaluate() on the device: the metric formulas of recommender/evaluation.py:90-253 as reductions
// over the device-resident [n_users x topk] recommendation array and the holdout (sorted by user), so that only a
// handful of sums visit the host — not the recommendation array (20 GB at 50M users x top-50) — and the rank sweep of
// evaluation/pipelines.py:81-116 can run without leaving the device.
//
// One thread per test user (holdouts are a few items per user): the ranks of its holdout items in its list, the
// hit / miss counts of evaluation.py:176-205, the ratios of :208-236, ARHR / MRR (:108-118), MAP (:120-133) and
// NDCG / NDCL (:136-173; the ideal order is "holdout items by descending relevance", ties cannot change the sums).
// Per-user values go to a [n_users x PK_EVAL_COLS] table that pk_colsum_f64 adds up in a fixed order.
#include "pk_common.h"
#include <math.h>

// columns of the per-user table
enum { EV_TP = 0, EV_FP, EV_TN, EV_FN, EV_PRECISION, EV_RECALL, EV_FALLOUT, EV_SPECIFITY, EV_MISS_RATE, EV_ARHR, EV_MRR,
       EV_MAP, EV_NDCG, EV_NDCL, EV_NRECS, EV_NHOLD };

__device__ __forceinline__ double ev_gain(double r, int alternative) { return alternative ? exp2(r) - 1.0 : r; }

__global__ __launch_bounds__(256) void eval_user_metrics_kernel(
    int64_t n_users, int topk, const int64_t *__restrict__ recs, int64_t ld, const int64_t *__restrict__ hold_ptr,
    const int64_t *__restrict__ hold_item, const double *__restrict__ hold_rel, const unsigned char *__restrict__ hold_pos,
    double penalty, double switch_positive, int alternative, double *__restrict__ out) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    const int64_t *row = recs + u * ld;
    const int64_t h0 = hold_ptr[u], h1 = hold_ptr[u + 1];
    const bool split = hold_pos != nullptr;
    int n_recs = 0;
    for (int t = 0; t < topk; ++t) n_recs += row[t] >= 0;
    double tp = 0, fp_cnt = 0, n_pos = 0, n_neg = 0, arhr = 0, mrr = 0, ap = 0;
    double dcg = 0, idcg = 0, dcl = 0, idcl = 0;
    for (int64_t e = h0; e < h1; ++e) {
        const int64_t item = hold_item[e];
        const double rel = hold_rel ? hold_rel[e] : 1.0;
        const bool pos = split ? hold_pos[e] != 0 : true;
        const bool nz = rel != 0.0;
        int rank = 0;
        for (int t = 0; t < topk; ++t)
            if (row[t] == item) {
                rank = t + 1;
                break;
            }
        // position of this entry in `np.argsort(relevance)[::-1]` (evaluation.py:147): descending relevance, equal
        // values in REVERSED holdout order (numpy sorts the few items of a user with a stable insertion sort, then the
        // list is reversed).  The tie rule only matters when a tie spans the positive / negative split, i.e. with
        // ignore_feedback (all relevances 1).
        int ideal = 0;
        for (int64_t f = h0; f < h1; ++f) {
            const double rf = hold_rel ? hold_rel[f] : 1.0;
            ideal += (rf > rel) || (rf == rel && f > e);
        }
        const double disc = rank > 0 ? 1.0 / log2(1.0 + (double)rank) : 0.0;
        const double ideal_disc = 1.0 / log2(2.0 + (double)ideal);
        if (pos) {
            n_pos += nz;
            const double g = ev_gain(rel, alternative);
            dcg += g * disc;
            idcg += g * ideal_disc;
            if (rank > 0 && nz) {
                tp += 1.0;
                const double rr = 1.0 / (double)rank;
                arhr += rr;
                mrr = rr > mrr ? rr : mrr;
                // precision at this hit: hits of the user ranked at or above it / rank
                int above = 0;
                for (int64_t f = h0; f < h1; ++f) {
                    const double rf = hold_rel ? hold_rel[f] : 1.0;
                    const bool pf = split ? hold_pos[f] != 0 : true;
                    if (!pf || rf == 0.0) continue;
                    const int64_t it = hold_item[f];
                    for (int t = 0; t < rank; ++t)
                        if (row[t] == it) {
                            ++above;
                            break;
                        }
                }
                ap += (double)above * rr;
            }
        } else {
            n_neg += nz;
            const double g = -ev_gain(rel - switch_positive, alternative);
            dcl += g * disc;
            idcl += g * ideal_disc;
            if (rank > 0 && nz) fp_cnt += 1.0;
        }
    }
    const double n_hold = (double)(h1 - h0);
    double fp, tn = 0.0, fn;
    if (!split) {
        fp = penalty > 0 ? penalty * ((double)n_recs - tp) : 0.0;
        fn = n_hold - tp;
    } else {
        fp = fp_cnt;
        tn = n_neg - fp_cnt;
        fn = n_pos - tp;
        if (penalty > 0) fp += penalty * ((double)n_recs - tp - fp_cnt);
    }
    double *o = out + u * 16;
    o[EV_TP] = tp;
    o[EV_FP] = fp;
    o[EV_TN] = tn;
    o[EV_FN] = fn;
    o[EV_PRECISION] = tp > 0 ? tp / (tp + fp) : 0.0;
    o[EV_RECALL] = tp > 0 ? tp / (tp + fn) : 0.0;
    o[EV_FALLOUT] = (split && fp > 0) ? fp / (fp + tn) : 0.0;
    o[EV_SPECIFITY] = (split && tn > 0) ? tn / (fp + tn) : 0.0;
    o[EV_MISS_RATE] = fn > 0 ? fn / (fn + tp) : 0.0;
    o[EV_ARHR] = arhr;
    o[EV_MRR] = mrr;
    const double n_rel_adj = n_hold < (double)topk ? n_hold : (double)topk;
    o[EV_MAP] = n_rel_adj > 0 ? ap / n_rel_adj : 0.0;
    o[EV_NDCG] = dcg > 0 ? dcg / idcg : 0.0;
    o[EV_NDCL] = (split && dcl > 0) ? dcl / idcl : 0.0;
    o[EV_NRECS] = (double)n_recs;
    o[EV_NHOLD] = n_hold;
}

extern "C" int32_t pk_eval_cols(void) { return 16; }

extern "C" int pk_eval_user_metrics(void *stream, int64_t n_users, int32_t topk, const int64_t *recs_dev, int64_t ld,
                                    const int64_t *hold_ptr_dev, const int64_t *hold_item_dev, const double *hold_rel_dev,
                                    const unsigned char *hold_pos_dev, double not_rated_penalty, double switch_positive,
                                    int32_t alternative, double *out_dev) {
    PK_REQUIRE(n_users >= 1 && topk >= 1 && ld >= topk && recs_dev && hold_ptr_dev && hold_item_dev && out_dev,
               "pk_eval_user_metrics: bad arguments");
    hipLaunchKernelGGL(eval_user_metrics_kernel, dim3((unsigned)pk_ceil_div(n_users, 256)), dim3(256), 0, pk_stream(stream),
                       n_users, topk, recs_dev, ld, hold_ptr_dev, hold_item_dev, hold_rel_dev, hold_pos_dev, not_rated_penalty,
                       switch_positive, alternative, out_dev);
    PK_CHECK_LAUNCH("eval_user_metrics_kernel");
    return PK_OK;
}

// ---- sums of the 16 columns of the per-user table in a fixed order (blocks of 4096 users, then the block sums) ----------
#define PK_EVAL_RED_ROWS 4096
__global__ __launch_bounds__(256) void eval_reduce_partial_kernel(int64_t n, const double *__restrict__ table,
                                                                  double *__restrict__ part) {
    __shared__ double s_red[256];
    const int c = threadIdx.x & 15, lane = threadIdx.x >> 4;          // 16 columns x 16 row lanes
    const int64_t r0 = (int64_t)blockIdx.x * PK_EVAL_RED_ROWS;
    int64_t r1 = r0 + PK_EVAL_RED_ROWS;
    if (r1 > n) r1 = n;
    double acc = 0.0;
    for (int64_t r = r0 + lane; r < r1; r += 16) acc += table[r * 16 + c];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0) {
        double t = 0.0;
        for (int l = 0; l < 16; ++l) t += s_red[l * 16 + c];
        part[(int64_t)blockIdx.x * 16 + c] = t;
    }
}

__global__ __launch_bounds__(64) void eval_reduce_final_kernel(int64_t n_blocks, const double *__restrict__ part,
                                                               double *__restrict__ sums) {
    const int c = threadIdx.x;
    if (c >= 16) return;
    double t = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) t += part[b * 16 + c];
    sums[c] = t;
}

extern "C" int64_t pk_eval_reduce_work_bytes(int64_t n_users) {
    return pk_ceil_div(n_users > 0 ? n_users : 1, PK_EVAL_RED_ROWS) * 16 * 8;
}

extern "C" int pk_eval_reduce(void *stream, int64_t n_users, const double *table_dev, double *sums_dev, void *work_dev) {
    PK_REQUIRE(n_users >= 1 && table_dev && sums_dev && work_dev, "pk_eval_reduce: bad arguments");
    const int64_t nb = pk_ceil_div(n_users, PK_EVAL_RED_ROWS);
    hipStream_t st = pk_stream(stream);
    hipLaunchKernelGGL(eval_reduce_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, n_users, table_dev,
                       static_cast<double *>(work_dev));
    hipLaunchKernelGGL(eval_reduce_final_kernel, dim3(1), dim3(64), 0, st, nb, static_cast<const double *>(work_dev), sums_dev);
    PK_CHECK_LAUNCH("eval_reduce kernels");
    return PK_OK;
}

// ---- coverage (evaluation.py:239-242): number of distinct recommended items -----------------------------------------
__global__ __launch_bounds__(256) void mark_ids_kernel(int64_t n, const int64_t *__restrict__ ids, int64_t n_bins,
                                                       int32_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    if (id >= 0 && id < n_bins) flags[id] = 1;      // every writer stores the same value
    else if (id < 0) flags[n_bins] = 1;             // the padding constant of short lists is one more "item" to np.unique
}

__global__ __launch_bounds__(256) void count_flags_kernel(int64_t n, const int32_t *__restrict__ flags,
                                                          unsigned long long *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n && flags[i] != 0;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));   // integer adds: order-free
}

/* count_dev[0] (int64) = number of distinct values of ids in [0, n_bins), plus one when any id is negative (the reference's
 * `len(np.unique(recommendations))` counts the padding constant of short lists too); flags_dev: int32[n_bins + 1] scratch */
extern "C" int pk_unique_count_i64(void *stream, int64_t n, const int64_t *ids_dev, int64_t n_bins, int32_t *flags_dev,
                                   int64_t *count_dev) {
    PK_REQUIRE(n >= 0 && n_bins >= 1 && flags_dev && count_dev && (n == 0 || ids_dev), "pk_unique_count_i64: bad arguments");
    hipStream_t st = pk_stream(stream);
    (void)hipMemsetAsync(flags_dev, 0, (n_bins + 1) * 4, st);
    (void)hipMemsetAsync(count_dev, 0, 8, st);
    if (n > 0)
        hipLaunchKernelGGL(mark_ids_kernel, dim3((unsigned)pk_ceil_div(n, 256)), dim3(256), 0, st, n, ids_dev, n_bins, flags_dev);
    hipLaunchKernelGGL(count_flags_kernel, dim3((unsigned)pk_ceil_div(n_bins + 1, 256)), dim3(256), 0, st, n_bins + 1, flags_dev,
                       reinterpret_cast<unsigned long long *>(count_dev));
    PK_CHECK_LAUNCH("unique_count kernels");
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_evalmetrics() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&eval_user_metrics_kernel));
}
