// K2b: the r LEADING eigenpairs of a symmetric PSD matrix (n <= 176, r <= 32) in ONE launch of one workgroup:
// Householder tridiagonalisation in LDS -> Sturm-count multisection for the r largest eigenvalues -> inverse iteration
// on the tridiagonal matrix -> back-transformation -> a check of what came out (residuals, orthonormality).
//
// Why: the HOOI unfoldings (lib/tensor.py:70-80: `svds(unfolding, k=r)`) need the r = 30 leading left singular vectors of
// an [n_mode x 120..150] matrix, i.e. 30 eigenpairs of its 120..150-column Gram matrix.  The Jacobi kernels (eigh.hip)
// deliver ALL of them, and pay for it: ~9 sweeps x n tournament steps x ~2 us = 2.0 ms at 120 columns, 3.6 ms at 150 (block
// method) — 52 of the 60 ms of a (30,30,5) build (tools/probes/hooi_eigh_sweeps.py); warm starts hardly help because the
// TRAILING eigenvectors — which nobody needs — keep rotating between HOOI iterations.  A direct method does n reflector
// steps ONCE and then only works on the r wanted pairs.
//
// Accuracy: eigenvalues to eps * ||S|| (absolute), vectors to eps * ||S|| / gap — the class of LAPACK's dsyevx, which is
// what the Gram route allows anyway (forming S already costs eps * ||S||).  Vectors of close eigenvalues (gap < 1e-3 ||T||)
// are re-orthogonalised inside their cluster (modified Gram-Schmidt, as dstein does).  The kernel CHECKS its result
// against S itself — max |S x - lambda x| and max |X^T X - I| — and reports failure (info[0] = 0: NaN-safe) instead of
// returning anything doubtful: the caller then runs the Jacobi kernel, which has no such limits.  Degenerate inputs
// (identity, zero matrix) go that way by design.
#include "pk_common.h"
#include <float.h>
#include <math.h>

#define ETOP_THREADS 1024
#define ETOP_NMAX 176
#define ETOP_NMIN 8
#define ETOP_RMAX 32
#define ETOP_ZS 33          // row stride of Z [n][33]: conflict-free for "lane = vector" and for "lane = row" access

__host__ __device__ constexpr size_t etop_lds_bytes(int n) {
    const size_t p1 = ((size_t)n * (n + 1) / 2 + 2 * (size_t)n) * sizeof(double);           // packed lower triangle, v, p
    const size_t p3 = (size_t)n * (ETOP_ZS + 2 * ETOP_RMAX) * sizeof(double);               // Z, U diagonal (reciprocal), U super-diagonal
    return p1 > p3 ? p1 : p3;
}

__device__ __forceinline__ int etop_pk(int r, int c) { return (r * (r + 1)) / 2 + c; }      // r >= c
__device__ __forceinline__ double etop_wave_sum(double v) {
    v += pk_lane_xor<1>(v);
    v += pk_lane_xor<2>(v);
    v += pk_lane_xor<4>(v);
    v += pk_lane_xor<8>(v);
    v += pk_lane_xor<16>(v);
    v += pk_lane_xor<32>(v);
    return v;
}
__device__ __forceinline__ double etop_half_sum(double v) {     // over the 32 lanes of a half wave
    v += pk_lane_xor<1>(v);
    v += pk_lane_xor<2>(v);
    v += pk_lane_xor<4>(v);
    v += pk_lane_xor<8>(v);
    v += pk_lane_xor<16>(v);
    return v;
}
// 1 / q for finite |q| >= DBL_MIN: hardware seed + two Newton steps (the Sturm and substitution recurrences are chains of
// dependent divisions; the IEEE expansion is ~4x as long)
__device__ __forceinline__ double etop_rcp(double q) {
    double r = __builtin_amdgcn_rcp(q);
    r = r * fma(-q, r, 2.0);
    r = r * fma(-q, r, 2.0);
    return r;
}
// number of eigenvalues of the tridiagonal (d, e2 = e^2) below x (LAPACK dlaebz's recurrence, pivmin guard).  q stays
// finite: |q| >= pivmin = DBL_MIN max(1, max e2) bounds e2 / q by 1 / DBL_MIN, so the Newton steps never see Inf * 0
__device__ __forceinline__ int etop_count(const double *d, const double *e2, int n, double x, double pivmin) {
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int c = q < 0.0;
    // the recurrence is one chain of dependent divisions; its operands (LDS broadcasts) are fetched a block ahead
    int i = 1;
    for (; i + 4 <= n; i += 4) {
        const double d0 = d[i], d1 = d[i + 1], d2 = d[i + 2], d3 = d[i + 3];
        const double f0 = e2[i - 1], f1 = e2[i], f2 = e2[i + 1], f3 = e2[i + 2];
#define ETOP_STURM(dd_, ff_)                                    \
        q = (dd_ - x) - ff_ * etop_rcp(q);                      \
        if (fabs(q) < pivmin) q = -pivmin;                      \
        c += q < 0.0;
        ETOP_STURM(d0, f0)
        ETOP_STURM(d1, f1)
        ETOP_STURM(d2, f2)
        ETOP_STURM(d3, f3)
    }
    for (; i < n; ++i) {
        ETOP_STURM(d[i], e2[i - 1])
    }
#undef ETOP_STURM
    return c;
}

// The same count with IEEE division (one thread, once per solve): the completeness check below must not share the
// approximate reciprocal whose rounding it guards against
__device__ __forceinline__ int etop_count_exact(const double *d, const double *e2, int n, double x, double pivmin) {
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int c = q < 0.0;
    for (int i = 1; i < n; ++i) {
        q = (d[i] - x) - e2[i - 1] / q;
        if (fabs(q) < pivmin) q = -pivmin;
        c += q < 0.0;
    }
    return c;
}

__global__ __launch_bounds__(ETOP_THREADS) void eigh_top_kernel(int n, const double *__restrict__ S, int64_t lds_, int r,
                                                                double *__restrict__ evecs, int64_t ldv,
                                                                double *__restrict__ evals, double *refl,
                                                                int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double etop_smem[];
    __shared__ double s_d[ETOP_NMAX], s_e[ETOP_NMAX], s_e2[ETOP_NMAX], s_tau[ETOP_NMAX];
    __shared__ double s_lam[ETOP_RMAX], s_shift[ETOP_RMAX], s_lo[ETOP_RMAX], s_hi[ETOP_RMAX];
    __shared__ int s_cluster[ETOP_RMAX], s_again[ETOP_RMAX];
    __shared__ int s_cnt[ETOP_THREADS];
    __shared__ double s_red[ETOP_THREADS / 64];
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ETOP_PROFILE
    unsigned long long prof_t[8];
    int prof_n = 0;
#define ETOP_STAMP() prof_t[prof_n++] = __builtin_readcyclecounter()
#else
#define ETOP_STAMP()
#endif
    ETOP_STAMP();

    // ---- P0: S -> packed lower triangle in LDS, scaled by a power of two so that max |entry| <= 1 -------------------
    double *A = etop_smem;
    double *v = A + (n * (n + 1)) / 2;
    double *p = v + n;
    double amax = 0.0;
    for (int e = tid; e < n * n; e += ETOP_THREADS) {
        const int i = e / n, j = e - i * n;
        if (j <= i) amax = fmax(amax, fabs(S[(int64_t)i * lds_ + j]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) s_red[wave] = amax;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    amax = 0.0;
    for (int w = 0; w < ETOP_THREADS / 64; ++w) amax = fmax(amax, s_red[w]);
    if (!(amax > 0.0) || !(amax < INFINITY)) {      // zero matrix, NaN, Inf: not ours (uniform branch)
        if (tid == 0) info[0] = 0;
        return;
    }
    const int ex = __builtin_amdgcn_frexp_exp(amax);
    for (int e = tid; e < n * n; e += ETOP_THREADS) {
        const int i = e / n, j = e - i * n;
        if (j <= i) A[etop_pk(i, j)] = ldexp(S[(int64_t)i * lds_ + j], -ex);
    }
    __syncthreads();

    ETOP_STAMP();
    // ---- P1: Householder tridiagonalisation (dsytd2, lower): T = Q^T A Q, reflector k kept in refl[k * n + .] ----------
    // A step has two short serial parts — the reflector's scalars, and K / w between the product and the update — and
    // they belong to ONE wave: done redundantly by all sixteen they cost every SIMD four times their instructions (the
    // first version: 7.5 K clocks per step, most of it these parts and their wave reductions).
    constexpr int XPL = (ETOP_NMAX + 63) / 64;            // entries of a column per lane of wave 0
#ifdef ETOP_PROFILE
    unsigned long long sub_t[4] = {0ull, 0ull, 0ull, 0ull}, sub_0;
#define ETOP_SUB(i) { const unsigned long long now_ = __builtin_readcyclecounter(); sub_t[i] += now_ - sub_0; sub_0 = now_; }
#else
#define ETOP_SUB(i)
#endif
    for (int k = 0; k < n - 2; ++k) {
        const int m = n - k - 1, g0 = k + 1;
#ifdef ETOP_PROFILE
        sub_0 = __builtin_readcyclecounter();
#endif
        if (wave == 0) {
            double x[XPL];
            double part = 0.0;
#pragma unroll
            for (int u = 0; u < XPL; ++u) {
                const int i = lane + 64 * u;
                x[u] = (i < m) ? A[etop_pk(g0 + i, k)] : 0.0;
                if (i >= 1) part = fma(x[u], x[u], part);
            }
            const double xn2 = etop_wave_sum(part);
            const double alpha = __shfl(x[0], 0, 64);
            double tau = 0.0, beta = alpha, inv = 0.0;
            if (xn2 > 0.0) {
                // seeds + Newton steps instead of the IEEE sqrt / divide expansions (~100 instructions on the critical
                // path of every step); entries are scaled to <= 1, nothing over- or underflows
                const double ss = fma(alpha, alpha, xn2);
                double y = __builtin_amdgcn_rsq(ss);
                y = y * fma(-0.5 * ss, y * y, 1.5);
                y = y * fma(-0.5 * ss, y * y, 1.5);
                double nrm = ss * y;
                nrm = fma(fma(-nrm, nrm, ss), 0.5 * y, nrm);
                beta = -copysign(nrm, alpha);
                double rb = etop_rcp(beta);
                rb = rb * fma(-beta, rb, 2.0);
                tau = (beta - alpha) * rb;
                const double dd = alpha - beta;
                inv = etop_rcp(dd);
                inv = inv * fma(-dd, inv, 2.0);
            }
#pragma unroll
            for (int u = 0; u < XPL; ++u) {
                const int i = lane + 64 * u;
                if (i < m) {
                    const double vi = (i == 0) ? 1.0 : x[u] * inv;
                    v[i] = vi;
                    refl[(int64_t)k * n + i] = vi;
                }
            }
            if (lane == 0) {
                s_d[k] = A[etop_pk(k, k)];
                s_e[k] = beta;
                s_tau[k] = tau;
            }
        }
        __syncthreads();
        ETOP_SUB(0)
        const double tau = s_tau[k];
        if (tau != 0.0) {                       // uniform
            // p = A22 v: 4, 8 or 16 lanes per row (more as the trailing block shrinks), packed indices advanced by
            // additions (row r of the packed triangle starts r + 1 entries behind row r - 1)
            const int tpr_log2 = m > 128 ? 2 : m > 64 ? 3 : 4;
            const int tpr = 1 << tpr_log2;
            const int row = tid >> tpr_log2, q = tid & (tpr - 1);
            double acc = 0.0, acc2 = 0.0;
            if (row < m) {
                const int gi = g0 + row;
                const double *ar = A + etop_pk(gi, g0);
                int j = q;
                for (; j + tpr <= row; j += 2 * tpr) {
                    acc = fma(ar[j], v[j], acc);
                    acc2 = fma(ar[j + tpr], v[j + tpr], acc2);
                }
                for (; j <= row; j += tpr) acc = fma(ar[j], v[j], acc);
                int idx = etop_pk(g0 + j, gi);                                    // column part: A(g0 + j, gi), j > row
                for (; j < m; j += tpr) {
                    acc2 = fma(A[idx], v[j], acc2);
                    idx += tpr * (g0 + j) + (tpr * (tpr + 1)) / 2;                  // sum of the next tpr row lengths
                }
            }
            acc += acc2;
            acc += pk_lane_xor<1>(acc);
            acc += pk_lane_xor<2>(acc);
            if (tpr_log2 >= 3) acc += pk_lane_xor<4>(acc);
            if (tpr_log2 >= 4) acc += pk_lane_xor<8>(acc);
            if (row < m && q == 0) p[row] = acc;
            __syncthreads();
            ETOP_SUB(1)
            // w = tau p - K v with K = tau^2 (p . v) / 2, written over p (wave 0);  A22 -= v w^T + w v^T (everybody)
            if (wave == 0) {
                double pl[XPL], vl[XPL];
                double pv = 0.0;
#pragma unroll
                for (int u = 0; u < XPL; ++u) {
                    const int i = lane + 64 * u;
                    pl[u] = (i < m) ? p[i] : 0.0;
                    vl[u] = (i < m) ? v[i] : 0.0;
                    pv = fma(pl[u], vl[u], pv);
                }
                pv = etop_wave_sum(pv);
                const double K = 0.5 * tau * tau * pv;
#pragma unroll
                for (int u = 0; u < XPL; ++u) {
                    const int i = lane + 64 * u;
                    if (i < m) p[i] = fma(tau, pl[u], -K * vl[u]);
                }
            }
            __syncthreads();
            ETOP_SUB(2)
            if (row < m) {
                const int gi = g0 + row;
                double *ar = A + etop_pk(gi, g0);
                const double vi = v[row], wi = p[row];
                for (int j = q; j <= row; j += tpr) ar[j] = fma(-vi, p[j], fma(-wi, v[j], ar[j]));     // (unrolling by four: no gain, the LDS pipe is the bound)
                // Round 4: FOUR ROWS PER THREAD (16 lanes per group of four rows; v[j] / (w[j], v[j]) read once per four matrix
                // elements: 8 -> 5 LDS instructions per four products, 16 -> 10 per four updated elements) was built, gave the same
                // eigenpairs, and was SLOWER: sub-phase clocks at 150 columns, product 3 663 -> 5 174, update 3 175 -> 5 846 (x 100),
                // a solve 0.83 -> 1.01 ms.  One row per thread has the 16 rows of a wave read CONSECUTIVE addresses in the column
                // part (a(j, gi) for adjacent gi) and short 32-byte runs in the row part; four rows per thread turn the column part
                // into 64 lanes on 16 different packed rows with a stride of ~150 doubles and the row part into four 128-byte runs
                // whose banks overlap pairwise: the LDS pipe is the bound, but by bank cycles, not by instruction count.
            }
            __syncthreads();
            ETOP_SUB(3)
        }
    }
#ifdef ETOP_PROFILE
    if (tid == 0)
        for (int i = 0; i < 4; ++i) info[8 + i] = (int)(sub_t[i] / 100);      // P1: scalars, product, K / w, update
#endif
    if (tid == 0) {
        s_d[n - 2] = A[etop_pk(n - 2, n - 2)];
        s_e[n - 2] = A[etop_pk(n - 1, n - 2)];
        s_d[n - 1] = A[etop_pk(n - 1, n - 1)];
        s_e[n - 1] = 0.0;
    }
    __syncthreads();
    if (tid < n) s_e2[tid] = s_e[tid] * s_e[tid];
    // norms and the Gershgorin interval (every thread for itself: n broadcast reads)
    double tn = 0.0, glo = INFINITY, ghi = -INFINITY, e2max = 0.0;
    for (int i = 0; i < n; ++i) {
        const double el = i ? fabs(s_e[i - 1]) : 0.0, er = (i < n - 1) ? fabs(s_e[i]) : 0.0;
        tn = fmax(tn, fmax(fabs(s_d[i]), er));
        e2max = fmax(e2max, er * er);
        glo = fmin(glo, s_d[i] - el - er);
        ghi = fmax(ghi, s_d[i] + el + er);
    }
    __syncthreads();
    const double pivmin = DBL_MIN * fmax(1.0, e2max);
    {
        const double span = fmax(ghi - glo, DBL_MIN);
        const double pad = 2.0 * DBL_EPSILON * n * span + 2.0 * pivmin;
        glo -= pad;
        ghi += pad;
    }

    ETOP_STAMP();
    // ---- P2: the r largest eigenvalues by multisection on Sturm counts ---------------------------------------------
    // eigenvalue j (descending) is the smallest x with count(x) >= n - j.  Round 0: 1024 points over the Gershgorin
    // interval, one per thread; then 32 lanes refine each eigenvalue's interval by 33 per round.
    {
        const double step = (ghi - glo) / (ETOP_THREADS + 1);
        s_cnt[tid] = etop_count(s_d, s_e2, n, glo + step * (tid + 1), pivmin);
        __syncthreads();
        if (tid < r) {
            const int want = n - tid;
            int lo_i = -1, hi_i = ETOP_THREADS;          // count(point lo_i) < want <= count(point hi_i); -1 / 1024 = the ends
            while (hi_i - lo_i > 1) {
                const int mid = (lo_i + hi_i) >> 1;
                if (s_cnt[mid] >= want) hi_i = mid; else lo_i = mid;
            }
            s_lo[tid] = (lo_i < 0) ? glo : glo + step * (lo_i + 1);
            s_hi[tid] = (hi_i >= ETOP_THREADS) ? ghi : glo + step * (hi_i + 1);
        }
        __syncthreads();
        // 16 lanes per eigenvalue, x 17 per round, eleven rounds: the recurrence is issue-bound (a wave instruction costs
        // its slots whatever it computes), and 8 waves x 11 rounds are fewer wave-rounds than 16 x 9 with 32 lanes
        const int j = tid >> 4, t = tid & 15;
        if (j < r) {                                      // whole 16-lane groups: the exchanges below stay inside one
            const int want = n - j;
            double lo = s_lo[j], hi = s_hi[j];
            for (int round = 0; round < 11; ++round) {
                const double x = lo + (hi - lo) * ((t + 1) * (1.0 / 17.0));
                const int c = etop_count(s_d, s_e2, n, x, pivmin);
                const unsigned long long b = __ballot(c >= want);
                const int base = lane & 48;
                const unsigned mine = (unsigned)(b >> base) & 0xFFFFu;
                const int first = mine ? __builtin_ctz(mine) : 16;
                const double x_first = __shfl(x, base + (first < 16 ? first : 0), 64);
                const double x_prev = __shfl(x, base + (first > 0 ? first - 1 : 0), 64);
                const double nlo = (first > 0) ? x_prev : lo, nhi = (first < 16) ? x_first : hi;
                lo = fmax(lo, nlo);
                hi = fmin(hi, nhi);
            }
            if (t == 0) s_lam[j] = 0.5 * (lo + hi);
        }
        __syncthreads();
        // COMPLETENESS: residuals and orthonormality (P5) say the r pairs are eigenpairs, not that they are the
        // LEADING ones — a count that the approximate reciprocal made non-monotone near an eigenvalue could bracket the
        // wrong slot.  With exact division: at most r eigenvalues may lie above a point just below the r-th value found
        // (more: a leading pair was skipped — or the (r+1)-th is equal to rounding, where "leading" is not defined: both go
        // to the Jacobi route, which computes every pair).
        if (tid == 0) {
            const double x = s_lam[r - 1] - 1e-11 * tn - 4.0 * pivmin;
            if (n - etop_count_exact(s_d, s_e2, n, x, pivmin) > r) s_fail = 1;
        }
        __syncthreads();
    }

    ETOP_STAMP();
    // ---- P3: eigenvectors of T by inverse iteration (dstein): vector j on lane j of wave 0 ---------------------------
    double *Z = etop_smem;                                  // [n][ETOP_ZS]
    double *Ua = Z + (size_t)n * ETOP_ZS;                   // [n][32]: reciprocal pivots of U
    double *Ub = Ua + (size_t)n * ETOP_RMAX;                // [n][32]: first super-diagonal of U
    const double ortol = 1e-3 * tn;
    if (tid == 0) {
        // shifts: equal eigenvalues are separated by 10 ulp so that their factorizations differ; clusters: runs of
        // eigenvalues closer than 1e-3 ||T|| are orthogonalised against each other
        for (int j = 0; j < r; ++j) {
            double sh = s_lam[j];
            if (j > 0) {
                const double pert = 10.0 * DBL_EPSILON * fmax(fabs(sh), tn * DBL_EPSILON);
                if (s_shift[j - 1] - sh < pert) sh = s_shift[j - 1] - pert;
            }
            s_shift[j] = sh;
            s_cluster[j] = (j > 0 && s_lam[j - 1] - s_lam[j] < ortol) ? s_cluster[j - 1] : j;
        }
        for (int j = 0; j < r; ++j) s_again[j] = (s_cluster[j] != j) || (j + 1 < r && s_cluster[j + 1] == j);
    }
    __syncthreads();
    // all r vectors on the lanes of ONE wave: a wave instruction costs its issue slots whatever the number of active
    // lanes, so two lanes on each of sixteen waves (the first version) made every SIMD issue the whole chain four times
    // (0.68 M clocks for 30 vectors where one vector alone took 0.28 M)
    const int vj = tid;
    const bool owner = tid < r;
    if (owner) {
        unsigned h = 0x9E3779B9u * (unsigned)(vj + 1);
        for (int i = 0; i < n; ++i) {                       // start vector: fixed pseudo-random numbers in (-1, 1)
            h = h * 1664525u + 1013904223u;
            Z[i * ETOP_ZS + vj] = ((h >> 8) * (1.0 / 8388608.0)) - 1.0;
        }
    }
    for (int round = 0; round < 2; ++round) {
        if (owner && (round == 0 || s_again[vj])) {
            const double sh = s_shift[vj];
            const double tl = fmax(tn * DBL_EPSILON, DBL_MIN);
            const int iters = round == 0 ? 2 : 1;
            double zs = 1.0;          // Z is read as Z * zs: 1 / max |z| of the previous sweep (start vectors are in (-1, 1))
            double nn = 1.0;          // sum of squares of the last sweep's solution
            for (int it = 0; it < iters; ++it) {
                // LU of T - sh I with partial pivoting (dlagtf), applied to the right-hand side on the fly; U keeps
                // (1 / pivot, first super-diagonal) per row in LDS and the interchange bits in three registers (second
                // super-diagonal = e[k+1] then).  Branch-free: the two lanes of a wave take different pivots.  The
                // operands of the NEXT step are requested before this step's division chain.
                unsigned long long w0 = 0ull, w1 = 0ull, w2 = 0ull;
                double a = s_d[0] - sh, b = s_e[0];
                double yk = Z[vj] * zs;
                double c = s_e[0], a1 = s_d[1] - sh, e1 = (1 < n - 1) ? s_e[1] : 0.0, y1 = Z[ETOP_ZS + vj] * zs;
                for (int k = 0; k < n - 1; ++k) {
                    const int kn = (k + 1 < n - 1) ? k + 1 : k;
                    const double c_n = s_e[kn], a1_n = s_d[kn + 1] - sh, e1_n = (kn + 1 < n - 1) ? s_e[kn + 1] : 0.0;
                    const double y1_n = Z[(kn + 1) * ETOP_ZS + vj] * zs;
                    const bool sw = fabs(c) > fabs(a);
                    double piv = sw ? c : a;
                    if (fabs(piv) < tl) piv = (piv < 0.0) ? -tl : tl;
                    const double rp = etop_rcp(piv);
                    const double mlt = (sw ? a : c) * rp;
                    const double ub = sw ? a1 : b;
                    const double ykeep = sw ? y1 : yk, yother = sw ? yk : y1;
                    Ua[k * ETOP_RMAX + vj] = rp;
                    Ub[k * ETOP_RMAX + vj] = ub;
                    Z[k * ETOP_ZS + vj] = ykeep;
                    const unsigned long long bit = (unsigned long long)sw << (k & 63);
                    w0 |= (k < 64) ? bit : 0ull;
                    w1 |= (k >= 64 && k < 128) ? bit : 0ull;
                    w2 |= (k >= 128) ? bit : 0ull;
                    a = fma(-mlt, ub, sw ? b : a1);
                    b = sw ? -mlt * e1 : e1;
                    yk = fma(-mlt, ykeep, yother);
                    c = c_n;
                    a1 = a1_n;
                    e1 = e1_n;
                    y1 = y1_n;
                }
                if (fabs(a) < tl) a = (a < 0.0) ? -tl : tl;
                // back substitution, operands one step ahead; the solution's largest entry and sum of squares on the way
                // (entries are bounded by ~1 / (eps ||T||) per sweep on a right-hand side of size 1: squares cannot overflow)
                double z1 = yk * etop_rcp(a), z2 = 0.0;     // z[k + 1], z[k + 2]
                Z[(n - 1) * ETOP_ZS + vj] = z1;
                double zmax = fabs(z1);
                nn = z1 * z1;
                int kp = n - 2;
                double ub_n = Ub[kp * ETOP_RMAX + vj], ua_n = Ua[kp * ETOP_RMAX + vj], y_n = Z[kp * ETOP_ZS + vj], e_n = 0.0;
                for (int k = n - 2; k >= 0; --k) {
                    const double ub = ub_n, ua = ua_n, e_k1 = e_n;
                    double y = y_n;
                    kp = (k > 0) ? k - 1 : 0;
                    ub_n = Ub[kp * ETOP_RMAX + vj];
                    ua_n = Ua[kp * ETOP_RMAX + vj];
                    y_n = Z[kp * ETOP_ZS + vj];
                    e_n = s_e[kp + 1];                       // kp + 1 <= n - 2: a stored super-diagonal entry
                    const unsigned long long w = (k >= 128) ? w2 : (k >= 64) ? w1 : w0;
                    const double dd = ((w >> (k & 63)) & 1ull) ? e_k1 : 0.0;
                    y = fma(-ub, z1, y);
                    y = fma(-dd, z2, y);
                    y *= ua;
                    Z[k * ETOP_ZS + vj] = y;
                    zmax = fmax(zmax, fabs(y));
                    nn = fma(y, y, nn);
                    z2 = z1;
                    z1 = y;
                }
                zs = (zmax > 0.0 && zmax < INFINITY) ? etop_rcp(zmax) : 1.0;
            }
            const double sc = 1.0 / sqrt(nn);               // a solution that is not finite gives NaN here: caught by the check
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                const double t0 = Z[i * ETOP_ZS + vj], t1 = Z[(i + 1) * ETOP_ZS + vj], t2 = Z[(i + 2) * ETOP_ZS + vj], t3 = Z[(i + 3) * ETOP_ZS + vj];
                Z[i * ETOP_ZS + vj] = t0 * sc;
                Z[(i + 1) * ETOP_ZS + vj] = t1 * sc;
                Z[(i + 2) * ETOP_ZS + vj] = t2 * sc;
                Z[(i + 3) * ETOP_ZS + vj] = t3 * sc;
            }
            for (; i < n; ++i) Z[i * ETOP_ZS + vj] *= sc;
        }
        __syncthreads();
        // modified Gram-Schmidt inside the clusters: one wave per cluster, members in order
        for (int c = wave; c < r; c += ETOP_THREADS / 64) {
            if (s_cluster[c] != c) continue;
            for (int j = c + 1; j < r && s_cluster[j] == c; ++j) {
                for (int i = c; i < j; ++i) {
                    double dot = 0.0;
                    for (int t = lane; t < n; t += 64) dot = fma(Z[t * ETOP_ZS + i], Z[t * ETOP_ZS + j], dot);
                    dot = etop_wave_sum(dot);
                    for (int t = lane; t < n; t += 64) Z[t * ETOP_ZS + j] = fma(-dot, Z[t * ETOP_ZS + i], Z[t * ETOP_ZS + j]);
                }
                double nn = 0.0;
                for (int t = lane; t < n; t += 64) nn = fma(Z[t * ETOP_ZS + j], Z[t * ETOP_ZS + j], nn);
                nn = etop_wave_sum(nn);
                const double sc = 1.0 / sqrt(nn);           // a vector that collapsed gives Inf / NaN: caught by the check
                for (int t = lane; t < n; t += 64) Z[t * ETOP_ZS + j] *= sc;
            }
        }
        __syncthreads();
    }

    ETOP_STAMP();
    // ---- P4: back-transformation X = Q Z, vector j on half wave j (no workgroup barrier: a vector has one owner) --------
    // ~110 instructions per reflector and half wave, sixteen waves: issue-bound (tried: 16 lanes per vector — the longer
    // per-lane chains cost more than the saved slots; reflectors three steps ahead — the ring's register moves cost more
    // than the L2 latency they hide)
    {
        const int qj = tid >> 5, qt = tid & 31;
        constexpr int VPL = (ETOP_NMAX + 31) / 32;          // reflector entries per lane
        if (qj < r) {
            double vn[VPL];
            {
                const int k = n - 3, m = n - k - 1;
#pragma unroll
                for (int u = 0; u < VPL; ++u) vn[u] = (k >= 0 && qt + 32 * u < m) ? refl[(int64_t)k * n + qt + 32 * u] : 0.0;
            }
            for (int k = n - 3; k >= 0; --k) {
                const int m = n - k - 1;
                double vc[VPL];
#pragma unroll
                for (int u = 0; u < VPL; ++u) vc[u] = vn[u];
                if (k > 0) {
#pragma unroll
                    for (int u = 0; u < VPL; ++u) vn[u] = (qt + 32 * u < m + 1) ? refl[(int64_t)(k - 1) * n + qt + 32 * u] : 0.0;
                }
                const double tau = s_tau[k];
                if (tau == 0.0) continue;
                double sacc = 0.0;
#pragma unroll
                for (int u = 0; u < VPL; ++u) {
                    const int i = qt + 32 * u;
                    if (i < m) sacc = fma(vc[u], Z[(k + 1 + i) * ETOP_ZS + qj], sacc);
                }
                sacc = etop_half_sum(sacc) * tau;
#pragma unroll
                for (int u = 0; u < VPL; ++u) {
                    const int i = qt + 32 * u;
                    if (i < m) Z[(k + 1 + i) * ETOP_ZS + qj] = fma(-sacc, vc[u], Z[(k + 1 + i) * ETOP_ZS + qj]);
                }
            }
        }
    }
    __syncthreads();

    ETOP_STAMP();
    // ---- P5: check against S itself, then write (rows of evecs = eigenvectors, largest |component| positive) ----------
    const int hj = tid >> 5, ht = tid & 31;
    double worst_res = 0.0, worst_orth = 0.0;
    bool bad = false;
    // (S x)_i = sum_c S[c][i] x_c (S is symmetric): half wave hj owns vector hj, its lane ht the rows ht, ht + 32, ...
    // S comes through LDS in chunks of 16 rows fetched ONCE by the whole workgroup (the space of P3's factors is free): every
    // half wave reading the rows from global memory itself was an L1 / L2 round trip per row and vector (0.22 M clocks)
    constexpr int RPL = (ETOP_NMAX + 31) / 32;
    constexpr int CH = 16;
    double acc[RPL];
#pragma unroll
    for (int u = 0; u < RPL; ++u) acc[u] = 0.0;
    {
        double *Sbuf = Z + (size_t)n * ETOP_ZS;            // CH * n doubles <= the two factor arrays (2 * 32 * n)
        for (int c0 = 0; c0 < n; c0 += CH) {
            const int rows = (n - c0 < CH) ? (n - c0) : CH;
            __syncthreads();                                // the previous chunk has been consumed
            for (int e = tid; e < rows * n; e += ETOP_THREADS) {
                const int rr = e / n, cc = e - rr * n;
                Sbuf[e] = S[(int64_t)(c0 + rr) * lds_ + cc];
            }
            __syncthreads();
            if (hj < r) {
                for (int rr = 0; rr < rows; ++rr) {
                    const double xc = Z[(c0 + rr) * ETOP_ZS + hj];
                    const double *sc = Sbuf + rr * n;
#pragma unroll
                    for (int u = 0; u < RPL; ++u) {
                        const int i = ht + 32 * u;
                        if (i < n) acc[u] = fma(sc[i], xc, acc[u]);
                    }
                }
            }
        }
    }
    if (hj < r) {
        const double lam = s_lam[hj];
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            const int i = ht + 32 * u;
            if (i < n) {
                const double rr = fabs(fma(-ldexp(lam, ex), Z[i * ETOP_ZS + hj], acc[u]));
                if (!(rr <= worst_res)) worst_res = rr;          // NaN propagates into worst_res
            }
        }
        for (int l = 0; l <= hj; ++l) {
            double dot = 0.0;
            for (int i = ht; i < n; i += 32) dot = fma(Z[i * ETOP_ZS + hj], Z[i * ETOP_ZS + l], dot);
            dot = etop_half_sum(dot);
            const double dev = fabs(dot - (l == hj ? 1.0 : 0.0));
            if (!(dev <= worst_orth)) worst_orth = dev;
        }
        const double scale = ldexp(fmax(fabs(s_lam[0]), tn), ex);           // ~ ||S||
        bad = !(worst_res <= 1e-12 * scale) || !(worst_orth <= 1e-12);
    }
    if (bad) atomicOr(&s_fail, 1);
    __syncthreads();
    const int fail = s_fail;
    if (hj < r && !fail) {
        double best = 0.0;
        int bestc = 0x7fffffff;
        for (int c = ht; c < n; c += 32) {
            const double a = fabs(Z[c * ETOP_ZS + hj]);
            if (a > best) {
                best = a;
                bestc = c;
            }
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int oc = __shfl_xor(bestc, off, 64);
            if (ob > best || (ob == best && oc < bestc)) {
                best = ob;
                bestc = oc;
            }
        }
        const double sgn = (bestc != 0x7fffffff && Z[bestc * ETOP_ZS + hj] < 0.0) ? -1.0 : 1.0;
        for (int c = ht; c < n; c += 32) evecs[(int64_t)hj * ldv + c] = sgn * Z[c * ETOP_ZS + hj];
        if (ht == 0) evals[hj] = fmax(ldexp(s_lam[hj], ex), 0.0);
    }
    if (tid == 0) info[0] = fail ? 0 : 1;
#ifdef ETOP_PROFILE
    ETOP_STAMP();
    if (tid == 0)
        for (int i = 1; i < prof_n; ++i) info[1 + i] = (int)((prof_t[i] - prof_t[i - 1]) / 100);     // P0 .. P5 in units of 100 clocks
#endif
}

extern "C" int pk_eigh_top_supported(int32_t n, int32_t r) {
    return n >= ETOP_NMIN && n <= ETOP_NMAX && r >= 1 && r <= ETOP_RMAX && r <= n;
}

extern "C" int64_t pk_eigh_top_work_bytes(int32_t n) { return (int64_t)n * n * (int64_t)sizeof(double); }

// The r leading eigenpairs of the symmetric PSD n x n matrix S (read only): evals[0..r) descending, row j of evecs = the
// j-th eigenvector.  info[0] = 1: the result passed the kernel's own check; 0: nothing usable was written — the caller
// falls back to pk_eigh_psd_f64.  work: pk_eigh_top_work_bytes(n) of device scratch (the reflectors).
extern "C" int pk_eigh_top_f64(void *stream, int32_t n, const double *S_dev, int64_t lds_, int32_t r, double *evecs_dev,
                               int64_t ldv, double *evals_dev, void *work_dev, int32_t *info_dev) {
    PK_REQUIRE(pk_eigh_top_supported(n, r), "pk_eigh_top_f64: n=%d, r=%d outside [%d, %d] x [1, %d]", n, r, ETOP_NMIN, ETOP_NMAX, ETOP_RMAX);
    PK_REQUIRE(lds_ >= n && ldv >= n, "pk_eigh_top_f64: bad leading dimension");
    PK_REQUIRE(S_dev && evecs_dev && evals_dev && work_dev && info_dev, "pk_eigh_top_f64: bad pointers");
    {   // per call: the limit is a per-DEVICE attribute (several contexts on different GPUs in one process)
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(eigh_top_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)etop_lds_bytes(ETOP_NMAX));
        if (e1 != hipSuccess) {
            pk_set_error("pk_eigh_top_f64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e1));
            return PK_E_LAUNCH;
        }
    }
    hipLaunchKernelGGL(eigh_top_kernel, dim3(1), dim3(ETOP_THREADS), etop_lds_bytes(n), pk_stream(stream), n, S_dev, lds_, r,
                       evecs_dev, ldv, evals_dev, reinterpret_cast<double *>(work_dev), info_dev);
    PK_CHECK_LAUNCH("eigh_top_kernel");
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_eigh_top() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&eigh_top_kernel));
}
