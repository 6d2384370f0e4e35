// K2: dense tall-skinny fp64 pieces of the block eigensolver and of HOOI (gfx950).
//   gram   : G = A^T B           (n x la, n x lb -> la x lb), split over rows, deterministic reduce
//   tsmm   : out = X C           (n x lin times lin x lout)
//   axpbypcz, resid_colnorm2, scale_cols, dgemm_small
// All HBM-bound (n is 1e5..5e7, l <= 256): the tiles are shaped so every global access is a
// coalesced run of >= 128 bytes and each element of the tall operand is read once per 64 output
// columns.  These replace ARPACK's Fortran re-orthogonalisation + LAPACK calls inside
// scipy.sparse.linalg.svds (reference call sites models.py:844, lib/tensor.py:71,75,79).
#include "pk_common.h"

// ------------------------------------------------------------------------------------------ gram
static int gram_splits(int64_t n, int la, int lb) {
    // enough row splits to fill the chip several times over (one 64 x 64 tile per workgroup and split), but at
    // least 64 rows per split; the partial tiles are added by gram_reduce_kernel in split order (deterministic).
    // (105 splits of 256 rows left 60 % of the CUs idle at n_items = 26 744, l = 64: 52 + 38 us per Gram matrix.)
    int64_t tiles = pk_ceil_div(la, 64) * pk_ceil_div(lb, 64);
    // wide left operands (block Lanczos: the whole Krylov basis against one block): every split writes la x lb partial
    // sums, so half as many workgroups there — 147 splits of a 896 x 64 product wrote and re-read 2 x 67 MB of partials
    // next to 110 MB of operands
    int64_t s = pk_ceil_div(tiles >= 4 ? 1024 : 2048, tiles);
    int64_t max_by_rows = pk_ceil_div(n, 64);
    if (s > max_by_rows) s = max_by_rows;
    if (s > 1024) s = 1024;
    // at most two steps of rows: ONE pass, the workgroup of a tile writes it — no partial sums, no reduce launch.  (Tried
    // for every product of <= 1024 rows — the projected problems of the block Lanczos build: a single workgroup walking
    // 896 rows is slower than 14 of them and a reduce launch: nested solve 34 -> 39.7 ms per build.)
    if (n <= 64) s = 1;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" int64_t pk_gram_work_bytes(int64_t n, int32_t la, int32_t lb) {
    return (int64_t)gram_splits(n, la, lb) * la * lb * (int64_t)sizeof(double);
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// block (256 threads = 4 waves) computes a 64x64 tile of A^T B over its row range on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64, D[16 x 16] += P[16 x 4] Q[4 x 16] with lane l holding P[l & 15][l >> 4], Q[l >> 4][l & 15] and
// D[(l >> 4) + 4 r][l & 15] in register r (MI355X guide: NOT the f32 C/D map).  For a Gram product the contraction runs
// over ROWS of the tall operands, so P = (A tile)^T: lane l reads sA[k0 + (l >> 4)][16 w + (l & 15)] — 16 consecutive
// doubles per quarter-wave, conflict-free — and wave w owns output rows [16 w, 16 w + 16) of the tile against all four
// 16-column blocks of B: 16 MFMAs per 16 staged rows instead of 256 scalar FMAs and 128 LDS reads per thread.
// (BASELINE.json configs[3] / north_star: the dense contractions of the HOOI path on MFMA; lib/tensor.py:70-80.)
__global__ __launch_bounds__(256) void gram_kernel(int64_t n, int la, int lb, const double *__restrict__ A,
                                                   int64_t lda, const double *__restrict__ B, int64_t ldb,
                                                   double *__restrict__ partial, int tiles_j,
                                                   int64_t rows_per_split) {
    // Round 4: 32 rows per step, the rows of step t + 1 requested (16-byte loads into registers) before the products of
    // step t (the first version loaded 16 rows with 8-byte loads, synchronised, multiplied, synchronised)
    constexpr int RT = 32;
    __shared__ __attribute__((aligned(16))) double sA[RT][64];
    __shared__ __attribute__((aligned(16))) double sB[RT][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
    const int split = blockIdx.y;
    const int64_t r_begin = (int64_t)split * rows_per_split;
    int64_t r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;
    const bool vec = ((lda | ldb) & 1) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && (la & 1) == 0 && (lb & 1) == 0;
    f64x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
    double2 ra[4], rb[4];          // slot = tid + 256 q: row = slot >> 5, column pair = slot & 31
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = tid + 256 * q;
            const int rr = slot >> 5, c = (slot & 31) * 2;
            const int64_t row = r0 + rr;
            const int ca = ti * 64 + c, cb = tj * 64 + c;
            double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
            if (row < r_end) {
                const double *pa = A + row * lda + ca, *pb = B + row * ldb + cb;
                if (vec && ca + 1 < la) va = *reinterpret_cast<const double2 *>(pa);
                else {
                    if (ca < la) va.x = pa[0];
                    if (ca + 1 < la) va.y = pa[1];
                }
                if (vec && cb + 1 < lb) vb = *reinterpret_cast<const double2 *>(pb);
                else {
                    if (cb < lb) vb.x = pb[0];
                    if (cb + 1 < lb) vb.y = pb[1];
                }
            }
            ra[q] = va;
            rb[q] = vb;
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += RT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = tid + 256 * q;
            *reinterpret_cast<double2 *>(&sA[slot >> 5][(slot & 31) * 2]) = ra[q];
            *reinterpret_cast<double2 *>(&sB[slot >> 5][(slot & 31) * 2]) = rb[q];
        }
        __syncthreads();
        if (r0 + RT < r_end) fetch(r0 + RT);
#pragma unroll
        for (int k0 = 0; k0 < RT; k0 += 4) {
            const double p = sA[k0 + (lane >> 4)][16 * wave + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, sB[k0 + (lane >> 4)][16 * b + (lane & 15)], acc[b], 0, 0, 0);
        }
        __syncthreads();
    }
    double *dst = partial + (int64_t)split * la * lb;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j = tj * 64 + 16 * b + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 64 + 16 * wave + (lane >> 4) + 4 * r;
            if (i < la && j < lb) dst[(int64_t)i * lb + j] = acc[b][r];
        }
    }
}

// 64 output elements per workgroup; its sixteen waves take every sixteenth split and their sums are added in wave order
// (with four waves a thread walked 104 partial tiles one after the other at n_items = 26 744: 32 us, more than the
// products themselves)
#define PK_GRAM_RED_WAVES 16
__global__ __launch_bounds__(64 * PK_GRAM_RED_WAVES) void gram_reduce_kernel(int la, int lb, int splits,
                                                                              const double *__restrict__ partial,
                                                                              double *__restrict__ G, int64_t ldg) {
    __shared__ double s_part[PK_GRAM_RED_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const int64_t total = (int64_t)la * lb;
    double acc = 0.0;
    if (e < total)
        for (int s = wave; s < splits; s += PK_GRAM_RED_WAVES) acc += partial[(int64_t)s * total + e];
    s_part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < total) {
        double tot = s_part[0][lane];
#pragma unroll
        for (int w = 1; w < PK_GRAM_RED_WAVES; ++w) tot += s_part[w][lane];
        G[(e / lb) * ldg + (e % lb)] = tot;
    }
}

extern "C" int pk_gram_f64(void *stream, int64_t n, int32_t la, int32_t lb, const double *A_dev, int64_t lda,
                           const double *B_dev, int64_t ldb, double *G_dev, int64_t ldg, void *work_dev) {
    PK_REQUIRE(n >= 1 && la >= 1 && lb >= 1 && la <= 4096 && lb <= 4096, "pk_gram_f64: bad sizes");
    PK_REQUIRE(lda >= la && ldb >= lb && ldg >= lb, "pk_gram_f64: bad leading dimension");
    PK_REQUIRE(work_dev != nullptr, "pk_gram_f64: work buffer required");
    hipStream_t st = pk_stream(stream);
    const int splits = gram_splits(n, la, lb);
    const int tiles_i = (int)pk_ceil_div(la, 64), tiles_j = (int)pk_ceil_div(lb, 64);
    int64_t rows_per_split = pk_ceil_div(n, splits);
    rows_per_split = pk_ceil_div(rows_per_split, 32) * 32;
    if (splits == 1 && ldg == lb) {       // single pass straight into G (the partial layout of split 0 IS G when ldg = lb)
        hipLaunchKernelGGL(gram_kernel, dim3(tiles_i * tiles_j, 1), dim3(256), 0, st, n, la, lb, A_dev, lda,
                           B_dev, ldb, G_dev, tiles_j, rows_per_split);
        PK_CHECK_LAUNCH("gram_kernel");
        return PK_OK;
    }
    hipLaunchKernelGGL(gram_kernel, dim3(tiles_i * tiles_j, splits), dim3(256), 0, st, n, la, lb, A_dev, lda,
                       B_dev, ldb, static_cast<double *>(work_dev), tiles_j, rows_per_split);
    PK_CHECK_LAUNCH("gram_kernel");
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)pk_ceil_div((int64_t)la * lb, 64)), dim3(64 * PK_GRAM_RED_WAVES), 0, st,
                       la, lb, splits, static_cast<const double *>(work_dev), G_dev, ldg);
    PK_CHECK_LAUNCH("gram_reduce_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------ tsmm
// block computes 64 rows x 64 output columns; k advanced 32 at a time through LDS; the products run on the fp64 matrix
// cores (v_mfma_f64_16x16x4_f64, layout in gram_kernel): wave w owns rows [16 w, 16 w + 16) of the block, P[i][k] =
// sX[16 w + i][k0 + k] (row stride 33: the 16 rows of a quarter-wave fall into different banks), Q[k][j] = sC[k0 + k][16 b + j].
// Round 4: the tile of step t + 1 is requested (16-byte loads into registers) BEFORE the products of step t, and stored to
// LDS after them.  The first version loaded, synchronised, multiplied, synchronised: with n / 16 waves in all (1.6 per SIMD
// at 26 744 rows) nothing covered the load latency — 117 us for a 26 744 x 448 operand (0.8 TB/s), three times per block of
// the block Lanczos build (re-orthogonalisation against the whole Krylov basis).
// SUB = 2: the general epilogue out = alpha X C + beta Zin + gamma Z2 (Zin / Z2 may be NULL) — one step of the Chebyshev
// recurrence on a small dense operator in ONE launch (driver.hip: DenseOp::filter_step; the projected problems of a block
// Lanczos build are solved by ~90 of those steps, each three launches before: Gram product, its reduction, the recurrence)
template <int SUB>
__global__ __launch_bounds__(256) void tsmm_kernel(int64_t n, int lin, int lout, const double *__restrict__ X,
                                                   int64_t ldx, const double *__restrict__ C, int64_t ldc,
                                                   double *__restrict__ out, int64_t ldo, const double *__restrict__ Zin,
                                                   int64_t ldz, double alpha = 0.0, double beta = 0.0, double gamma = 0.0,
                                                   const double *__restrict__ Z2 = nullptr, int64_t ldz2 = 0) {
    constexpr int KT = 32;
    __shared__ double sX[64][KT + 1];
    __shared__ __attribute__((aligned(16))) double sC[KT][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int col0 = blockIdx.y * 64;
    // 16-byte loads need even offsets: ldx / ldc even and 16-byte aligned bases (checked by the launcher: VEC)
    const bool vec = ((ldx | ldc) & 1) == 0 && ((((uintptr_t)X) | ((uintptr_t)C)) & 15) == 0 && (lin & 1) == 0 && (lout & 1) == 0;
    f64x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // per thread and k-tile: 4 double2 of X (slot = tid + 256 q: row = slot >> 4, k pair = slot & 15) and 4 double2 of C
    // (slot: k = slot >> 5, column pair = slot & 31)
    double2 rx[4], rc[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = tid + 256 * q;
            {
                const int r = slot >> 4, kk = (slot & 15) * 2;
                const int64_t row = row0 + r;
                double2 v = make_double2(0.0, 0.0);
                if (row < n) {
                    const double *src = X + row * ldx + k0 + kk;
                    if (vec && k0 + kk + 1 < lin) v = *reinterpret_cast<const double2 *>(src);
                    else {
                        if (k0 + kk < lin) v.x = src[0];
                        if (k0 + kk + 1 < lin) v.y = src[1];
                    }
                }
                rx[q] = v;
            }
            {
                const int kk = slot >> 5, c = (slot & 31) * 2;
                double2 v = make_double2(0.0, 0.0);
                if (k0 + kk < lin) {
                    const double *src = C + (int64_t)(k0 + kk) * ldc + col0 + c;
                    if (vec && col0 + c + 1 < lout) v = *reinterpret_cast<const double2 *>(src);
                    else {
                        if (col0 + c < lout) v.x = src[0];
                        if (col0 + c + 1 < lout) v.y = src[1];
                    }
                }
                rc[q] = v;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = tid + 256 * q;
            const int r = slot >> 4, kk = (slot & 15) * 2;
            sX[r][kk] = rx[q].x;
            sX[r][kk + 1] = rx[q].y;
            *reinterpret_cast<double2 *>(&sC[slot >> 5][(slot & 31) * 2]) = rc[q];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < lin; k0 += KT) {
        stage();
        __syncthreads();
        if (k0 + KT < lin) fetch(k0 + KT);          // in flight during the products below
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            const double p = sX[16 * wave + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, sC[kk + (lane >> 4)][16 * b + (lane & 15)], acc[b], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int c = col0 + 16 * b + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 16 * wave + (lane >> 4) + 4 * r;
            if (row < n && c < lout) {
                if constexpr (SUB == 1) out[row * ldo + c] = Zin[row * ldz + c] - acc[b][r];     // out may alias Zin: same element, same thread
                else if constexpr (SUB == 2) {
                    double v = alpha * acc[b][r];
                    if (Zin) v = fma(beta, Zin[row * ldz + c], v);
                    if (Z2) v = fma(gamma, Z2[row * ldz2 + c], v);
                    out[row * ldo + c] = v;
                } else out[row * ldo + c] = acc[b][r];
            }
        }
    }
}

extern "C" int pk_tsmm_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                           const double *C_dev, int64_t ldc, double *out_dev, int64_t ldo) {
    PK_REQUIRE(n >= 1 && lin >= 1 && lout >= 1, "pk_tsmm_f64: bad sizes");
    PK_REQUIRE(ldx >= lin && ldc >= lout && ldo >= lout, "pk_tsmm_f64: bad leading dimension");
    PK_REQUIRE(X_dev != out_dev, "pk_tsmm_f64: out must not alias X");
    hipLaunchKernelGGL(tsmm_kernel<0>, dim3((unsigned)pk_ceil_div(n, 64), (unsigned)pk_ceil_div(lout, 64)), dim3(256),
                       0, pk_stream(stream), n, lin, lout, X_dev, ldx, C_dev, ldc, out_dev, ldo, (const double *)nullptr, (int64_t)0);
    PK_CHECK_LAUNCH("tsmm_kernel");
    return PK_OK;
}

// out = Z - X C in one pass (the projection step X - V (V^T X) of the solvers without the product's round trip through
// memory: block Lanczos re-orthogonalises every new block against the whole basis, twice per step)
extern "C" int pk_tsmm_sub_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                               const double *C_dev, int64_t ldc, const double *Z_dev, int64_t ldz, double *out_dev, int64_t ldo) {
    PK_REQUIRE(n >= 1 && lin >= 1 && lout >= 1, "pk_tsmm_sub_f64: bad sizes");
    PK_REQUIRE(ldx >= lin && ldc >= lout && ldo >= lout && ldz >= lout, "pk_tsmm_sub_f64: bad leading dimension");
    PK_REQUIRE(X_dev != out_dev && Z_dev && C_dev && X_dev && out_dev, "pk_tsmm_sub_f64: bad pointers (out must not alias X)");
    hipLaunchKernelGGL(tsmm_kernel<1>, dim3((unsigned)pk_ceil_div(n, 64), (unsigned)pk_ceil_div(lout, 64)), dim3(256),
                       0, pk_stream(stream), n, lin, lout, X_dev, ldx, C_dev, ldc, out_dev, ldo, Z_dev, ldz);
    PK_CHECK_LAUNCH("tsmm_kernel<sub>");
    return PK_OK;
}

// out = alpha X C + beta Z1 + gamma Z2 in one pass (Z1 / Z2 may be NULL; out must alias none of the inputs)
extern "C" int pk_tsmm_axpby_f64(void *stream, int64_t n, int32_t lin, int32_t lout, const double *X_dev, int64_t ldx,
                                 const double *C_dev, int64_t ldc, double alpha, double beta, const double *Z1_dev, int64_t ldz1,
                                 double gamma, const double *Z2_dev, int64_t ldz2, double *out_dev, int64_t ldo) {
    PK_REQUIRE(n >= 1 && lin >= 1 && lout >= 1, "pk_tsmm_axpby_f64: bad sizes");
    PK_REQUIRE(ldx >= lin && ldc >= lout && ldo >= lout && (!Z1_dev || ldz1 >= lout) && (!Z2_dev || ldz2 >= lout),
               "pk_tsmm_axpby_f64: bad leading dimension");
    PK_REQUIRE(X_dev && C_dev && out_dev && X_dev != out_dev && C_dev != out_dev && Z1_dev != out_dev && Z2_dev != out_dev,
               "pk_tsmm_axpby_f64: bad pointers (out aliases an input)");
    hipLaunchKernelGGL(tsmm_kernel<2>, dim3((unsigned)pk_ceil_div(n, 64), (unsigned)pk_ceil_div(lout, 64)), dim3(256),
                       0, pk_stream(stream), n, lin, lout, X_dev, ldx, C_dev, ldc, out_dev, ldo, Z1_dev, ldz1, alpha, beta, gamma,
                       Z2_dev, ldz2);
    PK_CHECK_LAUNCH("tsmm_kernel<axpby>");
    return PK_OK;
}

// ---- the bookkeeping of an orthonormalisation pass without the host and without a dozen one-element launches:
// flags[0] += sum_i |info[i]|  (the Cholesky verdicts of the passes),  flags[1] = max(flags[1], max |G - I|) with NaN / inf
// counted as 1 (G: the l x l Gram matrix of the finished block), and info[] is ZEROED for the next block.  One workgroup.
__global__ __launch_bounds__(256) void orth_check_kernel(int l, const double *__restrict__ G, int64_t ldg, int32_t *__restrict__ info,
                                                         int n_info, double *__restrict__ flags) {
    __shared__ double s_max[256];
    double m = 0.0;
    for (int e = threadIdx.x; e < l * l; e += 256) {
        const int i = e / l, j = e - i * l;
        const double d = fabs(G[(int64_t)i * ldg + j] - (i == j ? 1.0 : 0.0));
        m = (d <= 1e300) ? fmax(m, d) : fmax(m, 1.0);          // NaN and inf count as 1 (a lost block)
    }
    s_max[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + w]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int bad = 0;
        for (int i = 0; i < n_info; ++i) {
            bad += info[i] < 0 ? -info[i] : info[i];
            info[i] = 0;
        }
        flags[0] += (double)bad;
        flags[1] = fmax(flags[1], s_max[0]);
    }
}

extern "C" int pk_orth_check_f64(void *stream, int32_t l, const double *G_dev, int64_t ldg, int32_t *info_dev, int32_t n_info,
                                 double *flags_dev) {
    PK_REQUIRE(l >= 1 && l <= 4096 && ldg >= l && G_dev && info_dev && n_info >= 0 && n_info <= 64 && flags_dev, "pk_orth_check_f64: bad arguments");
    hipLaunchKernelGGL(orth_check_kernel, dim3(1), dim3(256), 0, pk_stream(stream), l, G_dev, ldg, info_dev, n_info, flags_dev);
    PK_CHECK_LAUNCH("orth_check_kernel");
    return PK_OK;
}

// ------------------------------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void axpbypcz_kernel(int64_t n, double alpha, const double *__restrict__ Z,
                                                       double beta, const double *__restrict__ Y, double gamma,
                                                       const double *__restrict__ X, double *__restrict__ out) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (; i + 1 < n; i += stride) {
        double2 z = *reinterpret_cast<const double2 *>(Z + i);
        double2 r = make_double2(alpha * z.x, alpha * z.y);
        if (Y) {
            double2 y = *reinterpret_cast<const double2 *>(Y + i);
            r.x = fma(beta, y.x, r.x);
            r.y = fma(beta, y.y, r.y);
        }
        if (X) {
            double2 x = *reinterpret_cast<const double2 *>(X + i);
            r.x = fma(gamma, x.x, r.x);
            r.y = fma(gamma, x.y, r.y);
        }
        *reinterpret_cast<double2 *>(out + i) = r;
    }
    if (i < n) {  // odd tail element
        double r = alpha * Z[i];
        if (Y) r = fma(beta, Y[i], r);
        if (X) r = fma(gamma, X[i], r);
        out[i] = r;
    }
}

extern "C" int pk_axpbypcz_f64(void *stream, int64_t n_elems, double alpha, const double *Z_dev, double beta,
                               const double *Y_dev, double gamma, const double *X_dev, double *out_dev) {
    PK_REQUIRE(n_elems >= 0 && Z_dev && out_dev, "pk_axpbypcz_f64: bad arguments");
    PK_REQUIRE(((uintptr_t)Z_dev % 16 == 0) && ((uintptr_t)out_dev % 16 == 0) && ((uintptr_t)Y_dev % 16 == 0) &&
                   ((uintptr_t)X_dev % 16 == 0),
               "pk_axpbypcz_f64: buffers must be 16-byte aligned");
    if (n_elems == 0) return PK_OK;
    int64_t blocks = pk_ceil_div(pk_ceil_div(n_elems, 2), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpbypcz_kernel, dim3((unsigned)blocks), dim3(256), 0, pk_stream(stream), n_elems, alpha,
                       Z_dev, beta, Y_dev, gamma, X_dev, out_dev);
    PK_CHECK_LAUNCH("axpbypcz_kernel");
    return PK_OK;
}

// partial[b, j] = sum_{i in block b} (Z[i,j] - theta[j] X[i,j])^2 ; host (or caller) sums over b
#define PK_RESID_ROWS 1024
extern "C" int32_t pk_resid_blocks(int64_t n) { return (int32_t)pk_ceil_div(n, PK_RESID_ROWS); }

__global__ __launch_bounds__(256) void resid_colnorm2_kernel(int64_t n, int l, const double *__restrict__ Z,
                                                             int64_t ldz, const double *__restrict__ X,
                                                             int64_t ldx, const double *__restrict__ theta,
                                                             double *__restrict__ partial) {
    // 256 threads: column = tid % cw, row lane = tid / cw, with cw = min(l rounded to pow2, 256)
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * PK_RESID_ROWS;
    int64_t r1 = r0 + PK_RESID_ROWS;
    if (r1 > n) r1 = n;
    int cw = 1;
    while (cw < l && cw < 256) cw <<= 1;
    const int rl = tid / cw, nrl = 256 / cw;
    for (int j0 = 0; j0 < l; j0 += cw) {
        const int j = j0 + (tid % cw);
        double acc = 0.0;
        if (j < l) {
            const double th = theta[j];
            for (int64_t i = r0 + rl; i < r1; i += nrl) {
                double d = Z[i * ldz + j] - th * X[i * ldx + j];
                acc = fma(d, d, acc);
            }
        }
        red[tid] = acc;
        __syncthreads();
        for (int s = nrl / 2; s > 0; s >>= 1) {
            if (rl < s) red[tid] += red[tid + s * cw];
            __syncthreads();
        }
        if (rl == 0 && j < l) partial[(int64_t)blockIdx.x * l + j] = red[tid];
        __syncthreads();
    }
}

extern "C" int pk_resid_colnorm2_f64(void *stream, int64_t n, int32_t l, const double *Z_dev, int64_t ldz,
                                     const double *X_dev, int64_t ldx, const double *theta_dev,
                                     double *partial_dev) {
    PK_REQUIRE(n >= 1 && l >= 1 && ldz >= l && ldx >= l, "pk_resid_colnorm2_f64: bad sizes");
    hipLaunchKernelGGL(resid_colnorm2_kernel, dim3((unsigned)pk_resid_blocks(n)), dim3(256), 0, pk_stream(stream),
                       n, l, Z_dev, ldz, X_dev, ldx, theta_dev, partial_dev);
    PK_CHECK_LAUNCH("resid_colnorm2_kernel");
    return PK_OK;
}

__global__ __launch_bounds__(256) void scale_cols_kernel(int64_t n, int l, double *__restrict__ X, int64_t ldx,
                                                         const double *__restrict__ s) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * l, stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        int64_t i = e / l;
        int j = (int)(e % l);
        X[i * ldx + j] *= s[j];
    }
}

extern "C" int pk_scale_cols_f64(void *stream, int64_t n, int32_t l, double *X_dev, int64_t ldx,
                                 const double *s_dev) {
    PK_REQUIRE(n >= 1 && l >= 1 && ldx >= l, "pk_scale_cols_f64: bad sizes");
    int64_t blocks = pk_ceil_div(n * l, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, pk_stream(stream), n, l, X_dev, ldx,
                       s_dev);
    PK_CHECK_LAUNCH("scale_cols_kernel");
    return PK_OK;
}

// C[M x N] = op(A) op(B), one thread per output element (l x l glue only; perf irrelevant)
__global__ __launch_bounds__(256) void dgemm_small_kernel(int tA, int tB, int M, int N, int K,
                                                          const double *__restrict__ A, int64_t lda,
                                                          const double *__restrict__ B, int64_t ldb,
                                                          double *__restrict__ C, int64_t ldc) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)M * N) return;
    const int i = (int)(e / N), j = (int)(e % N);
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        double a = tA ? A[(int64_t)k * lda + i] : A[(int64_t)i * lda + k];
        double b = tB ? B[(int64_t)j * ldb + k] : B[(int64_t)k * ldb + j];
        acc = fma(a, b, acc);
    }
    C[(int64_t)i * ldc + j] = acc;
}

extern "C" int pk_dgemm_small_f64(void *stream, int transA, int transB, int32_t M, int32_t N, int32_t K,
                                  const double *A_dev, int64_t lda, const double *B_dev, int64_t ldb,
                                  double *C_dev, int64_t ldc) {
    PK_REQUIRE(M >= 1 && N >= 1 && K >= 1 && ldc >= N, "pk_dgemm_small_f64: bad sizes");
    hipLaunchKernelGGL(dgemm_small_kernel, dim3((unsigned)pk_ceil_div((int64_t)M * N, 256)), dim3(256), 0,
                       pk_stream(stream), transA, transB, M, N, K, A_dev, lda, B_dev, ldb, C_dev, ldc);
    PK_CHECK_LAUNCH("dgemm_small_kernel");
    return PK_OK;
}

// ---- Cholesky of a small Gram matrix + inverse of the factor (CholeskyQR orthonormalisation) ----------
// G + shift_rel * trace(G) * I  (n x n, symmetric positive definite) = R^T R with R upper triangular; writes
// Rinv = R^-1 (upper triangular, zeros below), so that X <- X Rinv has X^T X = I up to cond(G) * eps.
// One workgroup, ONE barrier per column, no substitution phase: the elimination G = L D L^T is run on the augmented
// matrix [G | I] — the row operations that clear column j below the diagonal turn I into L^-1 — and both halves
// share one n x n array: the upper triangle (with the diagonal) holds D L^T, the strictly lower triangle L^-1
// (unit diagonal implied; the symmetric lower half of G is never needed).  Step j, for every row i > j with
// f = A[j][i] / d_j:   A[i][c] -= f * A[j][c]  for c < j (L^-1 part),  A[i][j] = -f,  A[i][k] -= f * A[j][k]  for
// k >= i (trailing update); row j itself is not touched, so the only hazard is between steps.  At the end
// R = D^1/2 L^T, hence Rinv[k][c] = Linv[c][k] / sqrt(d_c).  Threads form a (rows x columns) grid, a thread keeps its
// column of the pivot row in a register.  The array lives in LDS for n <= PK_CHOL_LDS_MAX (leading dimension n + 1
// against bank conflicts of the transposed read at the end), in global `work` otherwise.
// Round 1 factorised with three barriers per column and back-substituted one thread per column through global
// memory: 161 us at n = 64, 727 us at n = 128; the eigensolver orthonormalises 3 times per outer iteration.
// info[0] = 0, or j + 1 if the pivot of column j is not positive (rank-deficient block: the caller falls
// back to the eigen-whitening, which clamps).
#define PK_CHOL_LDS_MAX 136
#define PK_CHOL_THREADS 1024
// SCALED: the factorisation runs on the column-scaled matrix D G D, D = diag(G)^-1/2 (unit diagonal; a zero or negative diagonal
// entry is a failure like a non-positive pivot), and Rinv comes back as D R'^-1: X Rinv is orthonormal all the same, but the
// Cholesky sees the conditioning of the SCALED block — for a block of filtered Ritz vectors, whose columns differ by the
// filter's amplification (up to 1e7) and are otherwise nearly orthogonal, ~1e1 instead of ~1e14 (round 6: one pass where two
// were needed between the segments of a nested solve).
template <bool IN_LDS, bool SCALED = false>
__global__ __launch_bounds__(PK_CHOL_THREADS) void chol_rinv_kernel(int n, const double *__restrict__ G, int64_t ldg,
                                                                    double shift_rel, double *__restrict__ Rinv,
                                                                    int64_t ldr, double *__restrict__ work,
                                                                    int32_t *__restrict__ info, int tx_log2) {
    extern __shared__ double chol_lds[];
    __shared__ double s_shift;
    double *A = IN_LDS ? chol_lds : work;
    const int ld = IN_LDS ? n + 1 : n;
    const int tid = threadIdx.x;
    const int TX = 1 << tx_log2, TY = (int)blockDim.x >> tx_log2;      // (the launcher sizes the workgroup for n: a barrier of 4 waves costs a third of one of 16)
    const int tx = tid & (TX - 1), ty = tid >> tx_log2;
    if (tid < 64) {
        double tr = 0.0;                             // shift = shift_rel * trace(G) >= shift_rel * ||X||_2^2
        for (int i = tid; i < n; i += 64) tr += SCALED ? 1.0 : G[(int64_t)i * ldg + i];      // (the scaled matrix has a unit diagonal)
        for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);      // fixed order: the same sum on every rank
        if (tid == 0) s_shift = shift_rel * tr;
    }
    __syncthreads();
    const double shift = s_shift;
    for (int i = ty; i < n; i += TY)
        for (int c = tx; c < n; c += TX) {
            double g = (c >= i) ? G[(int64_t)i * ldg + c] : 0.0;
            if constexpr (SCALED) {
                const double gi = G[(int64_t)i * ldg + i], gc = G[(int64_t)c * ldg + c];
                // a non-positive diagonal entry makes the first pivot of its column fail below (NaN or <= 0 compares false to > 0)
                g = (c > i) ? g / (sqrt(gi) * sqrt(gc)) : (c == i ? (gi > 0.0 ? 1.0 : -1.0) : 0.0);
            }
            A[i * ld + c] = (c > i) ? g : (c == i ? g + shift : 0.0);
        }
    int fail = 0;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double d = A[j * ld + j];
        if (!(d > 0.0)) {
            fail = j + 1;
            break;                                   // uniform: every thread read the same d
        }
        const double invd = 1.0 / d;
        for (int c = tx; c < n; c += TX) {
            const double pj = (c == j) ? 1.0 : A[j * ld + c];
            // (issuing the loads of four rows before the first store — row j is not written in step j — was measured and is
            // SLOWER: 33.7 us against 25 us on the build's mix of 16 x 16 and 64 x 64 factorisations; so was a 256-thread workgroup)
            for (int i = j + 1 + ty; i < n; i += TY)
                if (c <= j || c >= i) {
                    const double f = A[j * ld + i] * invd;
                    A[i * ld + c] = fma(-f, pj, A[i * ld + c]);
                }
        }
    }
    __syncthreads();
    if (tid == 0) info[0] = fail;
    if (fail) return;
    for (int k = ty; k < n; k += TY)
        for (int c = tx; c < n; c += TX) {
            double v = 0.0;
            if (c >= k) {
                const double rs = 1.0 / sqrt(A[c * ld + c]);
                v = (c == k) ? rs : A[c * ld + k] * rs;
                if constexpr (SCALED) v /= sqrt(G[(int64_t)k * ldg + k]);       // Rinv = D R'^-1: row k carries d_k
            }
            Rinv[(int64_t)k * ldr + c] = v;
        }
}

extern "C" int64_t pk_chol_work_bytes(int32_t n) { return (n > PK_CHOL_LDS_MAX) ? (int64_t)n * n * 8 : 0; }

static int chol_rinv_launch(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel, double *Rinv_dev, int64_t ldr,
                            void *work_dev, int32_t *info_dev, bool scaled);

extern "C" int pk_chol_rinv_f64(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel,
                                double *Rinv_dev, int64_t ldr, void *work_dev, int32_t *info_dev) {
    return chol_rinv_launch(stream, n, G_dev, ldg, shift_rel, Rinv_dev, ldr, work_dev, info_dev, false);
}

extern "C" int pk_chol_rinv_scaled_f64(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel,
                                       double *Rinv_dev, int64_t ldr, void *work_dev, int32_t *info_dev) {
    return chol_rinv_launch(stream, n, G_dev, ldg, shift_rel, Rinv_dev, ldr, work_dev, info_dev, true);
}

static int chol_rinv_launch(void *stream, int32_t n, const double *G_dev, int64_t ldg, double shift_rel, double *Rinv_dev, int64_t ldr,
                            void *work_dev, int32_t *info_dev, bool scaled) {
    PK_REQUIRE(n >= 1 && n <= 1024 && ldg >= n && ldr >= n, "pk_chol_rinv_f64: bad sizes");
    PK_REQUIRE(G_dev && Rinv_dev && info_dev && (n <= PK_CHOL_LDS_MAX || work_dev), "pk_chol_rinv_f64: bad pointers");
    static PkDeviceOnce attr_set;   
    if (attr_set.pending()) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_rinv_kernel<true, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize,
                                            PK_CHOL_LDS_MAX * (PK_CHOL_LDS_MAX + 1) * 8);
        if (e1 == hipSuccess)
            e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_rinv_kernel<true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, PK_CHOL_LDS_MAX * (PK_CHOL_LDS_MAX + 1) * 8);
        if (e1 != hipSuccess) {
            pk_set_error("pk_chol_rinv_f64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e1));
            return PK_E_LAUNCH;
        }
        attr_set.done();   
    }
    int tx_log2 = 4;                                 // columns of the thread grid: the power of two >= n, 16 ... 256
    while ((1 << tx_log2) < n && tx_log2 < 8) ++tx_log2;
    // one barrier per column is what the factorisation costs: small matrices take a small workgroup (round 6: the nested solves
    // of a narrow-block Lanczos build run 150 of these per build; 64 x 64 on 1024 threads: 42 us)
    // (round 6 measured three cheaper-looking forms of this kernel for n <= 64 and kept none: 256 threads instead of 1024 — 75 us
    // against 37 at 64 x 64; the loads of four rows issued before the first store — 33.7 us against 25 us on a build's mix of
    // sizes; a dedicated kernel, lane = column, one wave for n <= 16 and four above, loads batched — 19 us against 12 at 16 x 16,
    // 66 us against 37 at 64 x 64 (tools/probes/chol_probe.py))
    const int threads = PK_CHOL_THREADS;
#define PK_CHOL_LAUNCH(SC)                                                                                                   \
    do {                                                                                                                     \
        if (n <= PK_CHOL_LDS_MAX)                                                                                            \
            hipLaunchKernelGGL((chol_rinv_kernel<true, SC>), dim3(1), dim3(threads), (size_t)n * (n + 1) * sizeof(double),   \
                               pk_stream(stream), n, G_dev, ldg, shift_rel, Rinv_dev, ldr, static_cast<double *>(work_dev),  \
                               info_dev, tx_log2);                                                                           \
        else                                                                                                                 \
            hipLaunchKernelGGL((chol_rinv_kernel<false, SC>), dim3(1), dim3(PK_CHOL_THREADS), 0, pk_stream(stream), n, G_dev, \
                               ldg, shift_rel, Rinv_dev, ldr, static_cast<double *>(work_dev), info_dev, tx_log2);           \
    } while (0)
    if (scaled) PK_CHOL_LAUNCH(true); else PK_CHOL_LAUNCH(false);
#undef PK_CHOL_LAUNCH
    PK_CHECK_LAUNCH("chol_rinv_kernel");
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_dense() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&gram_reduce_kernel));
}
