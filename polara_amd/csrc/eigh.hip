// K2: symmetric PSD eigen-decomposition by one-sided (Hestenes) Jacobi, one workgroup of 16 waves.
//
// Used for the l x l Rayleigh-Ritz / whitening problems of the block eigensolver and for the Gram
// matrices of the HOOI unfoldings — the role LAPACK (dsyevd/dgesdd inside scipy's svds,
// models.py:844, lib/tensor.py:71-79) plays in the reference.  l <= 1024, typically 64..256, so the
// whole problem lives in one CU's L1/L2 and the only thing that matters is the number of barriers:
// a round-robin tournament gives n/2 independent row pairs per step, one __syncthreads per step.
//
// Row formulation: W := S (symmetric PSD).  Plane rotations are applied to ROW pairs of W until all
// rows are mutually orthogonal.  At convergence W = diag(lambda) U^T: lambda_i = ||W_i|| and
// W_i / lambda_i is the eigenvector — no accumulated rotation matrix is needed.
// One-sided Jacobi is insensitive to row scaling, which is what the whitening step needs
// (Gram matrices of Chebyshev-filtered blocks are badly scaled by construction).
#include "pk_common.h"
#include <math.h>

#define EIGH_THREADS 1024
#define EIGH_WAVES (EIGH_THREADS / 64)

__device__ __forceinline__ void rr_pair(int m, int step, int k, int &p, int &q) {
    // circle method on m (even) players: player m-1 fixed, the others rotate.  0 <= step < m - 1, 0 <= k < m / 2, so
    // both sums stay below 2 (m - 1): one conditional subtraction instead of a division by a run-time value
    const int r = m - 1;
    if (k == 0) {
        p = r;
        q = step;
    } else {
        p = step + k;
        p -= (p >= r) ? r : 0;
        q = step - k + r;
        q -= (q >= r) ? r : 0;
    }
    if (p > q) {
        int t = p;
        p = q;
        q = t;
    }
}

// Plane rotation that makes two rows with squared norms alpha, beta and inner product gamma orthogonal:
//   t = sgn(d) 2 gamma / (|d| + sqrt(d^2 + 4 gamma^2)),  d = beta - alpha;   cs = 1 / sqrt(1 + t^2),  sn = cs t
// (the textbook zeta = d / (2 gamma), t = sgn(zeta) / (|zeta| + sqrt(1 + zeta^2)) multiplied through by |2 gamma|: one square
// root and one reciprocal less).  The whole chain sits on the critical path of a tournament step — 16 waves wait at the
// barrier for it — so it runs on v_rsq_f64 / v_rcp_f64 seeds with Newton steps instead of the IEEE divide and sqrt
// expansions (~5 x 15 dependent instructions).  The ANGLE only needs to be good enough for the pair to come out
// orthogonal to working accuracy next time round (two Newton steps); cs is refined three times because cs^2 + sn^2 =
// cs^2 (1 + t^2) must be 1 to rounding — a rotation that is not orthogonal would rescale the rows, i.e. the eigenvalues.
// The inputs are scaled by a power of two so that nothing overflows or underflows whatever the scale of S.
__device__ __forceinline__ double eigh_rsqrt(double x, int newton) {
    double y = __builtin_amdgcn_rsq(x);
    for (int i = 0; i < newton; ++i) y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
__device__ __forceinline__ bool eigh_rotation(double alpha, double beta, double gamma, double tol2, double &cs, double &sn) {
    const int e = __builtin_amdgcn_frexp_exp(fmax(alpha, beta));       // 0 for a zero argument
    alpha = ldexp(alpha, -e);
    beta = ldexp(beta, -e);
    gamma = ldexp(gamma, -e);
    if (!(gamma * gamma > tol2 * alpha * beta) || gamma == 0.0) return false;
    const double d = beta - alpha, g2 = 2.0 * gamma;
    const double x = fma(d, d, g2 * g2);
    const double h = x * eigh_rsqrt(x, 2);
    const double s = fabs(d) + h;
    double r = __builtin_amdgcn_rcp(s);
    r = r * fma(-s, r, 2.0);
    r = r * fma(-s, r, 2.0);
    const double t = (d >= 0.0 ? g2 : -g2) * r;
    cs = eigh_rsqrt(fma(t, t, 1.0), 3);
    sn = cs * t;
    return true;
}

// One-sided Jacobi needs no accumulated rotation matrix for a PSD input: at convergence the rows of
// W = J S are mutually orthogonal TO RELATIVE ACCURACY (that is the stopping rule) and W = Lambda U^T,
// so eigenvector i is simply W_i / ||W_i|| and lambda_i = ||W_i||.  Rows that vanish exactly (an
// exactly singular S, e.g. the all-zero matrix) are completed by Gram-Schmidt on unit vectors.
// IN_LDS: the n x n work matrix lives in the CU's LDS (n <= 136: 148 KiB of the 160 KiB), so a
// rotation costs LDS latency instead of an L2 round trip; larger n (<= 1024) stay L2 resident.
// Each wave works on PW independent row pairs at a time so the three wave reductions of a pair
// overlap with those of the other pairs (the butterfly is latency-, not throughput-bound).
#define EIGH_PW 4   // row pairs per wave: one per 16-lane DPP row

// All-reduce (sum) of a double within each 16-lane DPP row using only VALU data-parallel
// primitives: quad_perm xor-1, xor-2, row_half_mirror, row_mirror.  No LDS traffic — the
// ds_bpermute-based __shfl_xor version made the single CU's LDS pipe the bottleneck of the sweep.
template <int CTRL>
__device__ __forceinline__ double eigh_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double eigh_row16_sum(double v) {
    v = eigh_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = eigh_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = eigh_dpp_add<0x141>(v);  // row_half_mirror
    v = eigh_dpp_add<0x140>(v);  // row_mirror
    return v;
}

// One pair of a tournament step, done by a 16-lane DPP row: lane t owns columns t + 16 j.  NC > 0: the (at most NC)
// columns of both rows a lane owns stay in registers between the three inner products and the rotation, so a step moves
// every row through the LDS pipe twice (read, write) instead of three times — with one workgroup on one CU that pipe
// (128 B/clk) is what bounds a step once n reaches ~100: 128 x 128 x 8 B x 3 = 3 072 clocks of a ~4 500-clock step.
// NC == 0: any n, rows re-read for the rotation.  Same summation order either way.
template <int NC>
__device__ __forceinline__ bool eigh_pair_step(double *wp, double *wq, int n, int t, bool act, double tol2) {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
    double cs = 1.0, sn = 0.0;
    if constexpr (NC > 0) {
        double a[NC], b[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = t + 16 * j;
            const bool in = act && c < n;
            a[j] = in ? wp[c] : 0.0;
            b[j] = in ? wq[c] : 0.0;
            alpha = fma(a[j], a[j], alpha);
            beta = fma(b[j], b[j], beta);
            gamma = fma(a[j], b[j], gamma);
        }
        alpha = eigh_row16_sum(alpha);
        beta = eigh_row16_sum(beta);
        gamma = eigh_row16_sum(gamma);
        const bool rot = act && eigh_rotation(alpha, beta, gamma, tol2, cs, sn);
        if (rot) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = t + 16 * j;
                if (c < n) {
                    wp[c] = cs * a[j] - sn * b[j];
                    wq[c] = sn * a[j] + cs * b[j];
                }
            }
        }
        return rot;
    } else {
        if (act) {
            for (int c = t; c < n; c += 16) {
                const double a = wp[c], b = wq[c];
                alpha = fma(a, a, alpha);
                beta = fma(b, b, beta);
                gamma = fma(a, b, gamma);
            }
        }
        alpha = eigh_row16_sum(alpha);
        beta = eigh_row16_sum(beta);
        gamma = eigh_row16_sum(gamma);
        const bool rot = act && eigh_rotation(alpha, beta, gamma, tol2, cs, sn);
        if (rot) {
            for (int c = t; c < n; c += 16) {
                const double a = wp[c], b = wq[c];
                wp[c] = cs * a - sn * b;
                wq[c] = sn * a + cs * b;
            }
        }
        return rot;
    }
}
#define EIGH_LDS_MAX 136
template <bool IN_LDS, int NC>
__global__ __launch_bounds__(EIGH_THREADS) void eigh_psd_kernel(int n, double *__restrict__ Wg, int64_t ldwg,
                                                                double *__restrict__ Rg, int64_t ldrg,
                                                                double *__restrict__ evals, int max_sweeps,
                                                                double tol, int *__restrict__ info, int precondition,
                                                                const int *__restrict__ chol_flag) {
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    constexpr int NMAX = IN_LDS ? EIGH_LDS_MAX : 1024;
    __shared__ int s_rot;
    __shared__ int s_flag;
    __shared__ double s_lam[NMAX];
    __shared__ int s_rank[NMAX];
    __shared__ double s_vec[NMAX];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *W = IN_LDS ? eigh_smem : Wg;
    const int64_t ldw = IN_LDS ? n : ldwg;
    if (IN_LDS) {
        for (int e = tid; e < n * n; e += EIGH_THREADS) W[e] = Wg[(int64_t)(e / n) * ldwg + (e % n)];
    }
    __syncthreads();
    // Cholesky preconditioning (Veselic-Hari): S = R^T R, then the Jacobi sweeps run on the rows of R instead of the
    // rows of S.  The rows of the converged W are then sigma_i u_i^T with sigma_i^2 = lambda_i — the same eigenvectors —
    // but R has the square root of the condition number of S, and the matrices that arrive here are graded (Ritz values
    // over two to four orders of magnitude, HOOI Gram matrices over more): 23 sweeps become 10 on a 120 x 120 matrix
    // with eigenvalues e^(-j/8) (CPU study with this tournament order), 11 become 5 on a graded Gram matrix; a flat
    // spectrum gains nothing.  One barrier per column like chol_rinv_kernel (dense.hip).  A pivot that is not positive
    // — S singular to rounding — restores S and the sweeps run on it as before (rows that vanish are rebuilt below).
    // (n > EIGH_LDS_MAX: eigh_chol_global_kernel did this before the block rounds and left its verdict in chol_flag)
    bool chol_ok = (!IN_LDS && chol_flag) ? (*chol_flag != 0) : false;
    if (IN_LDS && n >= 4 && max_sweeps > 0 && precondition) {
        int tx_log2 = 4;
        while ((1 << tx_log2) < n) ++tx_log2;
        const int TX = 1 << tx_log2, TY = EIGH_THREADS >> tx_log2;
        const int tx = tid & (TX - 1), ty = tid >> tx_log2;
        int fail = 0;
        for (int j = 0; j < n; ++j) {
            __syncthreads();
            const double d = W[j * n + j];
            if (!(d > 0.0)) {
                fail = 1;
                break;                               // uniform: every thread read the same d
            }
            const double invd = 1.0 / d;
            for (int c = j + 1 + tx; c < n; c += TX) {
                const double pj = W[j * n + c];
                for (int i = j + 1 + ty; i <= c; i += TY) W[i * n + c] = fma(-W[j * n + i] * invd, pj, W[i * n + c]);
            }
        }
        __syncthreads();
        if (!fail) {
            for (int i = tid; i < n; i += EIGH_THREADS) s_lam[i] = 1.0 / sqrt(W[i * n + i]);
            __syncthreads();
            for (int e = tid; e < n * n; e += EIGH_THREADS) {
                const int i = e / n, c = e - i * n;
                W[e] = (c >= i) ? W[e] * s_lam[i] : 0.0;
            }
            chol_ok = true;
        } else {
            for (int e = tid; e < n * n; e += EIGH_THREADS) W[e] = Wg[(int64_t)(e / n) * ldwg + (e % n)];
        }
        __syncthreads();
    }

    const int m = (n + 1) & ~1;  // even number of players; index n (if any) is a bye
    const double tol2 = tol * tol;
    int sweep = 0, converged = (n <= 1);
    for (; sweep < max_sweeps && !converged; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int step = 0; step < m - 1; ++step) {
            for (int k0 = wave * EIGH_PW; k0 < m / 2; k0 += EIGH_WAVES * EIGH_PW) {
                // lanes [16 s, 16 s + 16) of the wave own pair k0 + s; lane t of the row owns columns t + 16 e
                const int slot = lane >> 4, t = lane & 15;
                const int k = k0 + slot;
                bool act = k < m / 2;
                int p = 0, q = 0;
                if (act) {
                    rr_pair(m, step, k, p, q);
                    act = q < n;  // bye
                }
                double *wp = W + (int64_t)p * ldw, *wq = W + (int64_t)q * ldw;
                // rotate only if the pair is not yet orthogonal to relative accuracy tol
                const bool rot = eigh_pair_step<NC>(wp, wq, n, t, act, tol2);
                const unsigned long long rb = __ballot(rot && t == 0);
                if (lane == 0 && rb) atomicAdd(&s_rot, __popcll(rb));
            }
            __syncthreads();  // workgroup-scope: rows written by one wave are read by another next step
        }
        converged = (s_rot == 0);
        __syncthreads();
    }

    // eigenvalues = row norms of W (squared when the sweeps ran on the Cholesky factor).  The rotation test gamma^2 > tol^2 alpha beta enforces mutual
    // orthogonality to RELATIVE accuracy for every pair of rows whose squared norms do not underflow
    // in that product; rows below 1e-100 * max (exactly singular S) escaped it, are not orthogonal
    // and are rebuilt below.  Everything above that — including rows at 1e-15 * max, which still carry
    // the trailing directions of a Chebyshev-filtered block — is kept and just normalised.
    for (int i = wave; i < n; i += EIGH_WAVES) {
        const double *wi = W + (int64_t)i * ldw;
        double a = 0.0;
        for (int c = lane; c < n; c += 64) a = fma(wi[c], wi[c], a);
        a = sqrt(pk_wave_sum(a));
        if (lane == 0) s_lam[i] = a;
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (int i = 0; i < n; ++i) mx = fmax(mx, s_lam[i]);
        s_vec[0] = mx * 1e-100;
    }
    __syncthreads();
    const double noise = s_vec[0];
    __syncthreads();
    for (int i = wave; i < n; i += EIGH_WAVES) {
        double *wi = W + (int64_t)i * ldw;
        const double a = s_lam[i];
        const bool ok = (a > noise) && (a > 1e-290);
        const double inv = ok ? 1.0 / a : 0.0;
        for (int c = lane; c < n; c += 64) wi[c] *= inv;
        if (lane == 0) s_rank[i] = ok ? 1 : 0;   // s_rank doubles as the "row is valid" flag here
    }
    __syncthreads();
    if (chol_ok) {                               // the rows were those of R: lambda = sigma^2
        for (int i = tid; i < n; i += EIGH_THREADS) s_lam[i] *= s_lam[i];
        __syncthreads();
    }
    // rebuild the invalid rows: Gram-Schmidt of unit vectors against all valid rows
    for (int i = 0; i < n; ++i) {
        if (s_rank[i]) continue;                       // uniform: shared memory
        for (int j = 0; j < n; ++j) {
            for (int c = tid; c < n; c += EIGH_THREADS) {
                double v = (c == j) ? 1.0 : 0.0;
                for (int r = 0; r < n; ++r) {
                    if (!s_rank[r]) continue;
                    const double *wr = W + (int64_t)r * ldw;
                    v -= wr[j] * wr[c];
                }
                s_vec[c] = v;
            }
            __syncthreads();
            if (tid == 0) {
                double nn = 0.0;
                for (int c = 0; c < n; ++c) nn += s_vec[c] * s_vec[c];
                s_flag = (nn > 0.25);
                if (s_flag) {
                    const double inv = 1.0 / sqrt(nn);
                    double *wi = W + (int64_t)i * ldw;
                    for (int c = 0; c < n; ++c) wi[c] = s_vec[c] * inv;
                    s_rank[i] = 1;
                }
            }
            __syncthreads();
            const int done = s_flag;
            __syncthreads();
            if (done) break;
        }
    }
    __syncthreads();
    // rank-sort descending (ties by index)
    for (int i = tid; i < n; i += EIGH_THREADS) {
        const double li = s_lam[i];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = s_lam[j];
            r += (lj > li) || (lj == li && j < i);
        }
        s_rank[i] = r;
        evals[r] = li;
    }
    __syncthreads();
    // emit the rows in rank order, fixing the sign so the largest |component| is positive
    for (int i = wave; i < n; i += EIGH_WAVES) {
        const double *ri = W + (int64_t)i * ldw;
        double best = 0.0;
        int bestc = 0x7fffffff;
        for (int c = lane; c < n; c += 64) {
            double v = fabs(ri[c]);
            if (v > best) {
                best = v;
                bestc = c;
            }
        }
        // wave arg-max with lowest-column tie-break
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double ob = __shfl_xor(best, off, 64);
            int oc = __shfl_xor(bestc, off, 64);
            if (ob > best || (ob == best && oc < bestc)) {
                best = ob;
                bestc = oc;
            }
        }
        const double sgn = (bestc != 0x7fffffff && ri[bestc] < 0.0) ? -1.0 : 1.0;
        double *dst = Rg + (int64_t)s_rank[i] * ldrg;
        for (int c = lane; c < n; c += 64) dst[c] = sgn * ri[c];
    }
    if (tid == 0 && info) {
        info[0] = sweep;
        info[1] = converged;
    }
}

// ---- n > EIGH_LDS_MAX: block Jacobi ------------------------------------------------------------------------
// The work matrix no longer fits one CU's LDS (150 x 150 fp64 = 180 KB: the unfoldings of a (30, 30, 5) HOOI; 256 x
// 256: the solver block of a rank-200 build).  Rotating it in global memory costs an L2 round trip per step of the
// tournament (20.7 ms at n = 150, 59 ms at n = 256 — 92 % of the (30,30,5) build, profiles/r02_hooi_*).  Instead the
// rows are cut into nb (even) blocks of w rows with 2 w n 8 B <= EIGH_BLOCK_LDS; a ROUND pairs the blocks by the circle
// method (nb / 2 disjoint pairs = nb / 2 workgroups), each workgroup loads its 2 w rows into LDS, runs one full
// round-robin sweep over ALL pairs of those rows there and writes them back; nb - 1 rounds make an outer sweep in
// which every pair of rows has met at least once.  Rounds are separate launches (stream order is the grid barrier);
// the rotation counts of an outer sweep go to a device counter, and every launch of a later sweep returns at once
// when the previous sweep rotated nothing — the host never reads the counter.
#define EIGH_BLOCK_LDS (144 * 1024)
// one round of the block method for the pair of row blocks (bi, bj): load 2 w rows, sweep all their pairs in LDS, store
template <int NC>
__device__ __forceinline__ void eigh_block_round_body(int n, double *__restrict__ Wg, int64_t ldwg, int w, int nb, int round,
                                                      int pair, int sweep, int *__restrict__ counters, double tol,
                                                      double *W, int *s_rot) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bi, bj;
    rr_pair(nb, round, pair, bi, bj);
    const int i0 = bi * w, j0 = bj * w;
    const int ci = max(0, min(n, i0 + w) - i0), cj = max(0, min(n, j0 + w) - j0);
    const int nr = ci + cj;                                    // rows of this pair of blocks
    if (nr < 2) return;                                        // workgroup-uniform
    for (int e = tid; e < nr * n; e += EIGH_THREADS) {
        const int r = e / n, c = e - r * n;
        const int gr = r < ci ? i0 + r : j0 + (r - ci);
        W[e] = Wg[(int64_t)gr * ldwg + c];
    }
    if (tid == 0) *s_rot = 0;
    __syncthreads();
    const int m = (nr + 1) & ~1;
    const double tol2 = tol * tol;
    for (int step = 0; step < m - 1; ++step) {
        for (int k0 = wave * EIGH_PW; k0 < m / 2; k0 += EIGH_WAVES * EIGH_PW) {
            const int slot = lane >> 4, t = lane & 15;
            const int k = k0 + slot;
            bool act = k < m / 2;
            int p = 0, q = 0;
            if (act) {
                rr_pair(m, step, k, p, q);
                act = q < nr;  // bye
            }
            double *wp = W + (int64_t)p * n, *wq = W + (int64_t)q * n;
            const bool rot = eigh_pair_step<NC>(wp, wq, n, t, act, tol2);
            const unsigned long long rb = __ballot(rot && t == 0);
            if (lane == 0 && rb) atomicAdd(s_rot, __popcll(rb));
        }
        __syncthreads();
    }
    for (int e = tid; e < nr * n; e += EIGH_THREADS) {
        const int r = e / n, c = e - r * n;
        const int gr = r < ci ? i0 + r : j0 + (r - ci);
        Wg[(int64_t)gr * ldwg + c] = W[e];
    }
    if (tid == 0 && *s_rot) atomicAdd(&counters[sweep], *s_rot);
}

template <int NC>
__global__ __launch_bounds__(EIGH_THREADS) void eigh_block_round_kernel(int n, double *__restrict__ Wg, int64_t ldwg, int w,
                                                                        int nb, int round, int sweep,
                                                                        int *__restrict__ counters, double tol) {
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    __shared__ int s_rot;
    if (sweep > 0 && counters[sweep - 1] == 0) return;        // the previous outer sweep found every pair orthogonal
    eigh_block_round_body<NC>(n, Wg, ldwg, w, nb, round, blockIdx.x, sweep, counters, tol, eigh_smem, &s_rot);
}

// All sweeps and rounds of the block method in ONE launch: nb / 2 workgroups (2 ... 6: one per CU, trivially co-resident),
// a grid barrier between rounds.  The launch-per-round form above issued max_sweeps * (nb - 1) launches per solve whatever
// the sweep count (the early-exit ones return at once but are still launched): 90 at 150 columns, 1 260 per (30,30,5)
// HOOI build — a third of its wall time on the host.  Barrier: every thread releases its stores (the L2s of different XCDs
// are not coherent within a kernel: agent-scope fences write back / invalidate), one thread counts the workgroup in on
// a monotone counter and spins (bounded: a barrier that does not complete sets an error flag instead of hanging the GPU).
__device__ __forceinline__ void eigh_grid_barrier(int *bar, int n_wg, int &epoch, int *err) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int target = epoch * n_wg;
        long spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 24)) {       // ~ seconds: something is badly wrong (a workgroup never arrived)
                atomicExch(err, 1);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence();
}

template <int NC>
__global__ __launch_bounds__(EIGH_THREADS) void eigh_block_persistent_kernel(int n, double *__restrict__ Wg, int64_t ldwg, int w,
                                                                             int nb, int max_sweeps, int *__restrict__ counters,
                                                                             double tol, int *__restrict__ bar) {
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    __shared__ int s_rot;
    __shared__ int s_epoch;
    if (threadIdx.x == 0) s_epoch = 0;
    __syncthreads();
    const int n_wg = gridDim.x;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // counters[sweep - 1] is complete: every workgroup added to it before the barrier that ended that sweep
        if (sweep > 0 && __hip_atomic_load(&counters[sweep - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
        for (int round = 0; round < nb - 1; ++round) {
            eigh_block_round_body<NC>(n, Wg, ldwg, w, nb, round, blockIdx.x, sweep, counters, tol, eigh_smem, &s_rot);
            int epoch = s_epoch;
            eigh_grid_barrier(bar, n_wg, epoch, bar + 1);
            if (threadIdx.x == 0) s_epoch = epoch;
            __syncthreads();
            if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // a barrier timed out: give up
        }
    }
}

__global__ void eigh_counters_init_kernel(int *counters, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) counters[i] = 0;
}

// Cholesky preconditioning of the block path (see eigh_psd_kernel): W := R with S = R^T R, in place in global memory,
// one workgroup, one barrier per column; S is kept in `backup` (the eigenvector output buffer, free until the last
// kernel) and restored if a pivot is not positive.  flag[0] = 1: the rounds run on R and the eigenvalues are the
// squared row norms.
__global__ __launch_bounds__(EIGH_THREADS) void eigh_chol_global_kernel(int n, double *__restrict__ W, int64_t ldw,
                                                                        double *__restrict__ backup, int64_t ldb,
                                                                        int *__restrict__ flag) {
    __shared__ double s_inv[1024];
    const int tid = threadIdx.x;
    int tx_log2 = 4;
    while ((1 << tx_log2) < n && tx_log2 < 8) ++tx_log2;
    const int TX = 1 << tx_log2, TY = EIGH_THREADS >> tx_log2;
    const int tx = tid & (TX - 1), ty = tid >> tx_log2;
    for (int i = ty; i < n; i += TY)
        for (int c = tx; c < n; c += TX) backup[(int64_t)i * ldb + c] = W[(int64_t)i * ldw + c];
    int fail = 0;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double d = W[(int64_t)j * ldw + j];
        if (!(d > 0.0)) {
            fail = 1;
            break;
        }
        const double invd = 1.0 / d;
        for (int c = j + 1 + tx; c < n; c += TX) {
            const double pj = W[(int64_t)j * ldw + c];
            for (int i = j + 1 + ty; i <= c; i += TY)
                W[(int64_t)i * ldw + c] = fma(-W[(int64_t)j * ldw + i] * invd, pj, W[(int64_t)i * ldw + c]);
        }
    }
    __syncthreads();
    if (!fail) {
        for (int i = tid; i < n; i += EIGH_THREADS) s_inv[i] = 1.0 / sqrt(W[(int64_t)i * ldw + i]);
        __syncthreads();
        for (int i = ty; i < n; i += TY)
            for (int c = tx; c < n; c += TX) W[(int64_t)i * ldw + c] = (c >= i) ? W[(int64_t)i * ldw + c] * s_inv[i] : 0.0;
    } else {
        for (int i = ty; i < n; i += TY)
            for (int c = tx; c < n; c += TX) W[(int64_t)i * ldw + c] = backup[(int64_t)i * ldb + c];
    }
    if (tid == 0) flag[0] = fail ? 0 : 1;
}

// info[0] = outer sweeps that rotated something (+ the clean one that certified convergence), info[1] = converged
__global__ void eigh_block_info_kernel(const int *counters, int max_sweeps, const int *bar, int *info) {
    int s = 0;
    while (s < max_sweeps && counters[s] != 0) ++s;
    info[0] = s < max_sweeps ? s + 1 : max_sweeps;
    info[1] = (s < max_sweeps) && bar[1] == 0;      // bar[1]: a grid barrier of the persistent kernel timed out
}

static int eigh_psd_impl(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev,
                         int64_t ldv, double *evals_dev, int32_t max_sweeps, double tol,
                         int32_t *info_dev, int persistent) {
    PK_REQUIRE(n >= 1 && n <= 1024, "pk_eigh_psd_f64: n=%d out of range [1,1024]", n);
    PK_REQUIRE(lds_ >= n && ldv >= n, "pk_eigh_psd_f64: bad leading dimension");
    PK_REQUIRE(S_dev && evecs_dev && evals_dev && S_dev != evecs_dev, "pk_eigh_psd_f64: bad pointers");
    if (max_sweeps <= 0) max_sweeps = 40;
    if (tol <= 0.0) tol = 2.0 * 2.220446049250313e-16 * sqrt((double)n);  // ~ LAPACK dgesvj's sqrt(m)*eps
    const int precondition = 1;        // sweeps on the Cholesky factor (0, the round-1 form — sweeps on S itself — is not selectable at run time)
    hipStream_t st = pk_stream(stream);
    if (n <= EIGH_LDS_MAX) {
        // instances by columns per lane of a 16-lane row (rows held in registers across a rotation)
        using kern_t = void (*)(int, double *, int64_t, double *, int64_t, double *, int, double, int *, int, const int *);
        kern_t kern = n <= 32 ? eigh_psd_kernel<true, 2> : n <= 64 ? eigh_psd_kernel<true, 4>
                    : n <= 96 ? eigh_psd_kernel<true, 6> : n <= 128 ? eigh_psd_kernel<true, 8> : eigh_psd_kernel<true, 9>;
        const int slot = n <= 32 ? 0 : n <= 64 ? 1 : n <= 96 ? 2 : n <= 128 ? 3 : 4;
        (void)slot;
        {   // per call: the attribute is per DEVICE and the coarse ABI may hold contexts on several; it is cheap
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                EIGH_LDS_MAX * EIGH_LDS_MAX * 8);
            if (e1 != hipSuccess) {
                pk_set_error("pk_eigh_psd_f64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e1));
                return PK_E_LAUNCH;
            }
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(EIGH_THREADS), (size_t)n * n * sizeof(double),
                           st, n, S_dev, lds_, evecs_dev, ldv, evals_dev, max_sweeps, tol, info_dev, precondition,
                           (const int *)nullptr);
        PK_CHECK_LAUNCH("eigh_psd_kernel");
        return PK_OK;
    }
    // block Jacobi: nb (even) blocks of w rows, two blocks at a time in LDS
    int nb = 4;
    while ((int64_t)2 * ((n + nb - 1) / nb) * n * 8 > EIGH_BLOCK_LDS) nb += 2;
    const int w = (n + nb - 1) / nb;
    const size_t lds_bytes = (size_t)2 * w * n * sizeof(double);
    using round_t = void (*)(int, double *, int64_t, int, int, int, int, int *, double);
    round_t round_kern = n <= 160 ? eigh_block_round_kernel<10> : n <= 256 ? eigh_block_round_kernel<16> : eigh_block_round_kernel<0>;
    const int slot_b = n <= 160 ? 0 : n <= 256 ? 1 : 2;
    (void)slot_b;
    {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(round_kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, EIGH_BLOCK_LDS);
        if (e1 != hipSuccess) {
            pk_set_error("pk_eigh_psd_f64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e1));
            return PK_E_LAUNCH;
        }
    }
    if (max_sweeps > 30) max_sweeps = 30;
    // the rotation counters of the outer sweeps live in evals_dev until the final pass writes the eigenvalues there
    // (n > 136 doubles: room for every counter, the Cholesky flag and the two words of the grid barrier)
    int *counters = reinterpret_cast<int *>(evals_dev);
    hipLaunchKernelGGL(eigh_counters_init_kernel, dim3(1), dim3(64), 0, st, counters, max_sweeps + 4);
    int *chol_flag = counters + max_sweeps + 1;          // zeroed above: "sweeps ran on S"
    int *bar = counters + max_sweeps + 2;                // [0] arrivals, [1] barrier time-out flag
    if (precondition)
        hipLaunchKernelGGL(eigh_chol_global_kernel, dim3(1), dim3(EIGH_THREADS), 0, st, n, S_dev, lds_, evecs_dev, ldv, chol_flag);
    bool launched = false;
    if (persistent) {
        using pers_t = void (*)(int, double *, int64_t, int, int, int, int *, double, int *);
        pers_t pers_kern = n <= 160 ? eigh_block_persistent_kernel<10> : n <= 256 ? eigh_block_persistent_kernel<16> : eigh_block_persistent_kernel<0>;
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(pers_kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, EIGH_BLOCK_LDS);
        if (e1 != hipSuccess) {
            pk_set_error("pk_eigh_psd_f64: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e1));
            return PK_E_LAUNCH;
        }
        // The kernel synchronises its nb / 2 workgroups with a grid barrier, so they must be CO-RESIDENT: a cooperative
        // launch is the runtime's guarantee of that (a plain launch next to other streams' kernels has none,
        // and a workgroup that is not resident turns the bounded spin into half-rotated vectors).  Where the runtime
        // refuses the cooperative launch the solve takes the launch-per-round form below — stream order is its barrier.
        int n_arg = n, w_arg = w, nb_arg = nb, ms_arg = max_sweeps;
        int64_t ld_arg = lds_;
        double tol_arg = tol;
        void *args[] = {&n_arg, &S_dev, &ld_arg, &w_arg, &nb_arg, &ms_arg, &counters, &tol_arg, &bar};
        hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(pers_kern), dim3(nb / 2), dim3(EIGH_THREADS),
                                                   args, (unsigned int)lds_bytes, st);
        if (ce == hipSuccess) launched = true;
        else (void)hipGetLastError();
    }
    if (!launched) {
        for (int sweep = 0; sweep < max_sweeps; ++sweep)
            for (int round = 0; round < nb - 1; ++round)
                hipLaunchKernelGGL(round_kern, dim3(nb / 2), dim3(EIGH_THREADS), lds_bytes, st, n, S_dev, lds_, w, nb,
                                   round, sweep, counters, tol);
    }
    if (info_dev) hipLaunchKernelGGL(eigh_block_info_kernel, dim3(1), dim3(1), 0, st, counters, max_sweeps, bar, info_dev);
    // norms, ordering, signs: the tail of the one-workgroup kernel (no sweeps of its own)
    hipLaunchKernelGGL((eigh_psd_kernel<false, 0>), dim3(1), dim3(EIGH_THREADS), 0, st, n, S_dev, lds_, evecs_dev, ldv, evals_dev,
                       0, tol, (int *)nullptr, 0, (const int *)chol_flag);
    PK_CHECK_LAUNCH("eigh block kernels");
    return PK_OK;
}

extern "C" int pk_eigh_psd_f64(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev,
                               int64_t ldv, double *evals_dev, int32_t max_sweeps, double tol,
                               int32_t *info_dev) {
    return eigh_psd_impl(stream, n, S_dev, lds_, evecs_dev, ldv, evals_dev, max_sweeps, tol, info_dev, 1);
}

// The same solve with one launch per (sweep, round) beyond 136 columns — no grid barrier anywhere: what a caller re-runs
// when the persistent form reports info[1] = 0 (identical below 137 columns, where one workgroup does everything).
extern "C" int pk_eigh_psd_rounds_f64(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev,
                                      int64_t ldv, double *evals_dev, int32_t max_sweeps, double tol,
                                      int32_t *info_dev) {
    return eigh_psd_impl(stream, n, S_dev, lds_, evecs_dev, ldv, evals_dev, max_sweeps, tol, info_dev, 0);
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_eigh() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&eigh_counters_init_kernel));
}
