// K5: sparse tensor-times-matrix for HOOI (CoFFee), fp64.
//
//   res[i0, j, k] += val * u[i1, j] * v[i2, k]          for every nnz (i0, i1, i2)
//
// restates numba `dttm_seq` (lib/sparse.py:203-216) as called from `ttm3d_seq`
// (lib/tensor.py:7-19).  The host layer sorts the nnz by the output mode once per mode, so the
// output row of a task is fixed and the r_a x r_b accumulator block of that row lives in
// registers (lane owns entries lane + 64 g of the flattened [r_a x r_b] block, j = e / r_b,
// k = e % r_b — the reference's C-order reshape, tensor.py:70,74,78).  Long rows (e.g. the five
// feedback levels of mode 2, each holding ~nnz/5 entries) are split into tasks with partial
// blocks that a fix-up pass adds in slot order: deterministic, no atomics.  Latency-bound by
// nature (0.5 GFLOP on ML-1M): the factor rows u[i1,:], v[i2,:] are tiny and L1/L2 resident.
#include "pk_common.h"

template <int EPL>
__global__ __launch_bounds__(256) void ttm_kernel(
    int64_t n_tasks, const int32_t *__restrict__ task_row, const int64_t *__restrict__ task_begin,
    const int64_t *__restrict__ task_end, const int32_t *__restrict__ task_slot,
    const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2, const double *__restrict__ vals,
    const double *__restrict__ u, int64_t ldu, int ra, const double *__restrict__ v, int64_t ldv, int rb,
    double *__restrict__ res, int64_t ldr, double *__restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    const int64_t p0 = task_begin[task], p1 = task_end[task];
    const int nout = ra * rb;
    int jj[EPL], kk[EPL];
    double acc[EPL];
#pragma unroll
    for (int g = 0; g < EPL; ++g) {
        int e = lane + 64 * g;
        if (e >= nout) e = nout - 1;
        jj[g] = e / rb;
        kk[g] = e % rb;
        acc[g] = 0.0;
    }
    for (int64_t p = p0; p < p1; p += 64) {
        const int cnt = (int)((p1 - p) < 64 ? (p1 - p) : 64);
        int a1 = 0, a2 = 0;
        double av = 0.0;
        if (lane < cnt) {
            a1 = idx1[p + lane];
            a2 = idx2[p + lane];
            av = vals ? vals[p + lane] : 1.0;
        }
        for (int t0 = 0; t0 < cnt; t0 += 4)
#pragma unroll
        for (int t = t0; t < t0 + 4; ++t) {  // padded steps use (i1 = 0, i2 = 0, val = 0)
            const int i1 = __builtin_amdgcn_readlane(a1, t);
            const int i2 = __builtin_amdgcn_readlane(a2, t);
            const double vv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(av), t),
                                               __builtin_amdgcn_readlane(__double2loint(av), t));
            const double *ur = u + (int64_t)i1 * ldu;
            const double *vr = v + (int64_t)i2 * ldv;
#pragma unroll
            for (int g = 0; g < EPL; ++g) acc[g] = fma(vv * ur[jj[g]], vr[kk[g]], acc[g]);
        }
    }
    const int slot = task_slot[task];
    double *dst = slot < 0 ? res + (int64_t)task_row[task] * ldr : partial + (int64_t)slot * nout;
#pragma unroll
    for (int g = 0; g < EPL; ++g) {
        const int e = lane + 64 * g;
        if (e < nout) dst[e] = acc[g];
    }
}

// Sum of the partial rows of a long row, in a FIXED order: a workgroup owns (long row, 32 output columns); its eight
// 32-lane halves take every eighth slot (s0 + g, s0 + g + 8, ...) and the eight sums are added in g order.  (One thread
// per column walking all slots: the feedback mode of a rating tensor has 5 rows of 2e5 entries — 770 slots each — and
// the fix-up took 0.4 ms on 5 workgroups, as long as the TTM itself.)
__global__ __launch_bounds__(256) void ttm_fixup_kernel(int64_t n_long, const int32_t *__restrict__ long_row,
                                                        const int32_t *__restrict__ slot_begin,
                                                        const int32_t *__restrict__ slot_end,
                                                        const double *__restrict__ partial, int nout,
                                                        double *__restrict__ res, int64_t ldr) {
    __shared__ double s_part[8][32];
    const int64_t r = blockIdx.x;
    if (r >= n_long) return;
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int c = blockIdx.y * 32 + l;
    const int s0 = slot_begin[r], s1 = slot_end[r];
    double acc = 0.0;
    if (c < nout)
        for (int s = s0 + g; s < s1; s += 8) acc += partial[(int64_t)s * nout + c];
    s_part[g][l] = acc;
    __syncthreads();
    if (g == 0 && c < nout) {
        double tot = s_part[0][l];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += s_part[k][l];
        res[(int64_t)long_row[r] * ldr + c] = tot;
    }
}

extern "C" int pk_ttm_f64(void *stream, int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                          const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                          const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                          const int32_t *long_slot_end_dev, const int32_t *idx1_dev, const int32_t *idx2_dev,
                          const double *vals_dev, const double *u_dev, int64_t ldu, int32_t ra, const double *v_dev,
                          int64_t ldv_, int32_t rb, double *res_dev, int64_t ldr, double *partial_dev) {
    PK_REQUIRE(n_tasks >= 0 && ra >= 1 && rb >= 1 && (int64_t)ra * rb <= 1024,
               "pk_ttm_f64: need ra*rb <= 1024 (got %d x %d)", ra, rb);
    PK_REQUIRE(ldu >= ra && ldv_ >= rb && ldr >= (int64_t)ra * rb, "pk_ttm_f64: bad leading dimension");
    PK_REQUIRE(n_long == 0 || partial_dev != nullptr, "pk_ttm_f64: partial buffer required");
    if (n_tasks == 0) return PK_OK;
    hipStream_t st = pk_stream(stream);
    dim3 grid((unsigned)pk_ceil_div(n_tasks, 4)), block(256);
    const int nout = ra * rb;
    const int epl = (nout + 63) / 64;
#define PK_TTM_LAUNCH(E)                                                                                       \
    hipLaunchKernelGGL((ttm_kernel<E>), grid, block, 0, st, n_tasks, task_row_dev, task_begin_dev, task_end_dev, \
                       task_slot_dev, idx1_dev, idx2_dev, vals_dev, u_dev, ldu, ra, v_dev, ldv_, rb, res_dev,  \
                       ldr, partial_dev)
    if (epl <= 1) PK_TTM_LAUNCH(1);
    else if (epl <= 2) PK_TTM_LAUNCH(2);
    else if (epl <= 4) PK_TTM_LAUNCH(4);
    else if (epl <= 8) PK_TTM_LAUNCH(8);
    else PK_TTM_LAUNCH(16);
#undef PK_TTM_LAUNCH
    PK_CHECK_LAUNCH("ttm_kernel");
    if (n_long > 0) {
        hipLaunchKernelGGL(ttm_fixup_kernel, dim3((unsigned)n_long, (unsigned)pk_ceil_div(nout, 32)), dim3(256), 0, st, n_long, long_row_dev,
                           long_slot_begin_dev, long_slot_end_dev, partial_dev, nout, res_dev, ldr);
        PK_CHECK_LAUNCH("ttm_fixup_kernel");
    }
    return PK_OK;
}

// ------------------------------------------------------------------------------------------
// pk_tucker_predict_f64: CoffeeModel.predict_feedback (models.py:1068-1091) — for every holdout (user, item) pair the
// feedback level f maximising  sum_abc g[a, b, c] u[user, a] v[item, b] w[f, c]  (np.argmax: the first maximum wins).
// One thread per pair: the r0 x r1 x r2 core is read through uniform (scalar) loads, the pair's contraction against the
// feedback mode is r2 <= 16 accumulators in registers, then L <= 64 dot products of length r2.  A few million
// multiply-adds for an ML-1M-sized holdout: latency-bound, tens of microseconds.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tucker_predict_kernel(int64_t n, const int64_t *__restrict__ users,
                                                             const int64_t *__restrict__ items, const double *__restrict__ u,
                                                             int64_t ldu, const double *__restrict__ v, int64_t ldv,
                                                             const double *__restrict__ w, int64_t ldw,
                                                             const double *__restrict__ g, int r0, int r1, int r2, int L,
                                                             int64_t *__restrict__ pred, double *__restrict__ scores) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    const double *ur = u + users[h] * ldu, *vr = v + items[h] * ldv;
    double acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0;
    for (int a = 0; a < r0; ++a) {
        const double ua = ur[a];
        for (int b = 0; b < r1; ++b) {
            const double p = ua * vr[b];
            const double *gc = g + ((int64_t)a * r1 + b) * r2;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < r2) acc[c] = fma(gc[c], p, acc[c]);
        }
    }
    int best = 0;
    double best_s = 0.0;
    for (int f = 0; f < L; ++f) {
        double sc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < r2) sc = fma(w[(int64_t)f * ldw + c], acc[c], sc);
        if (scores) scores[h * L + f] = sc;
        if (f == 0 || sc > best_s) {
            best_s = sc;
            best = f;
        }
    }
    pred[h] = best;
}

extern "C" int pk_tucker_predict_f64(void *stream, int64_t n, const int64_t *users_dev, const int64_t *items_dev,
                                     const double *u_dev, int64_t ldu, const double *v_dev, int64_t ldv_, const double *w_dev,
                                     int64_t ldw, const double *core_dev, int32_t r0, int32_t r1, int32_t r2, int32_t L,
                                     int64_t *pred_dev, double *scores_dev) {
    PK_REQUIRE(n >= 0 && r0 >= 1 && r1 >= 1 && r2 >= 1 && r2 <= 16 && L >= 1 && L <= 4096,
               "pk_tucker_predict_f64: need 1 <= r2 <= 16 feedback-mode columns (got %d), L = %d", r2, L);
    PK_REQUIRE(ldu >= r0 && ldv_ >= r1 && ldw >= r2, "pk_tucker_predict_f64: bad leading dimension");
    if (n == 0) return PK_OK;
    PK_REQUIRE(users_dev && items_dev && u_dev && v_dev && w_dev && core_dev && pred_dev, "pk_tucker_predict_f64: null pointer");
    hipLaunchKernelGGL(tucker_predict_kernel, dim3((unsigned)pk_ceil_div(n, 256)), dim3(256), 0, pk_stream(stream), n, users_dev,
                       items_dev, u_dev, ldu, v_dev, ldv_, w_dev, ldw, core_dev, r0, r1, r2, L, pred_dev, scores_dev);
    PK_CHECK_LAUNCH("tucker_predict_kernel");
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_ttm() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&ttm_fixup_kernel));
}
