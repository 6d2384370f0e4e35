// K2: symmetric PSD eigen-decomposition by one-sided (Hestenes) Jacobi, one workgroup of 16 waves.
//
// Used for the l x l Rayleigh-Ritz / whitening problems of the block eigensolver and for the Gram
// matrices of the HOOI unfoldings — the role LAPACK (dsyevd/dgesdd inside scipy's svds,
// models.py:844, lib/tensor.py:71-79) plays in the reference.  l <= 1024, typically 64..256, so the
// whole problem lives in one CU's L1/L2 and the only thing that matters is the number of barriers:
// a round-robin tournament gives n/2 independent row pairs per step, one __syncthreads per step.
//
// Row formulation: W := S (symmetric PSD), R := I.  Plane rotations are applied to ROW pairs of W
// until all rows are mutually orthogonal; the same rotations accumulate in R.  At convergence
// W = diag(lambda) U^T and R = U^T, i.e. row i of R is the eigenvector of lambda_i = ||W_i||.
// One-sided Jacobi is insensitive to row scaling, which is what the whitening step needs
// (Gram matrices of Chebyshev-filtered blocks are badly scaled by construction).
#include "pk_common.h"
#include <math.h>

#define EIGH_THREADS 1024
#define EIGH_WAVES (EIGH_THREADS / 64)

__device__ __forceinline__ void rr_pair(int m, int step, int k, int &p, int &q) {
    // circle method on m (even) players: player m-1 fixed, the others rotate
    if (k == 0) {
        p = m - 1;
        q = step % (m - 1);
    } else {
        p = (step + k) % (m - 1);
        q = (step - k + (m - 1)) % (m - 1);
    }
    if (p > q) {
        int t = p;
        p = q;
        q = t;
    }
}

__global__ __launch_bounds__(EIGH_THREADS) void eigh_psd_kernel(int n, double *__restrict__ W, int64_t ldw,
                                                                double *__restrict__ R, int64_t ldr,
                                                                double *__restrict__ evals, int max_sweeps,
                                                                double tol, int *__restrict__ info) {
    __shared__ int s_rot;
    __shared__ double s_lam[1024];
    __shared__ int s_rank[1024];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int e = tid; e < n * n; e += EIGH_THREADS) {
        int i = e / n, j = e % n;
        R[(int64_t)i * ldr + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();

    const int m = (n + 1) & ~1;  // even number of players; index n (if any) is a bye
    const double tol2 = tol * tol;
    int sweep = 0, converged = (n <= 1);
    for (; sweep < max_sweeps && !converged; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int step = 0; step < m - 1; ++step) {
            for (int k = wave; k < m / 2; k += EIGH_WAVES) {
                int p, q;
                rr_pair(m, step, k, p, q);
                if (q >= n) continue;  // bye
                double *wp = W + (int64_t)p * ldw, *wq = W + (int64_t)q * ldw;
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int c = lane; c < n; c += 64) {
                    double a = wp[c], b = wq[c];
                    alpha = fma(a, a, alpha);
                    beta = fma(b, b, beta);
                    gamma = fma(a, b, gamma);
                }
                alpha = pk_wave_sum(alpha);
                beta = pk_wave_sum(beta);
                gamma = pk_wave_sum(gamma);
                // rotate only if the pair is not yet orthogonal to relative accuracy tol
                if (gamma * gamma > tol2 * alpha * beta && gamma != 0.0) {
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + t * t);
                    const double sn = cs * t;
                    for (int c = lane; c < n; c += 64) {
                        double a = wp[c], b = wq[c];
                        wp[c] = cs * a - sn * b;
                        wq[c] = sn * a + cs * b;
                    }
                    double *rp = R + (int64_t)p * ldr, *rq = R + (int64_t)q * ldr;
                    for (int c = lane; c < n; c += 64) {
                        double a = rp[c], b = rq[c];
                        rp[c] = cs * a - sn * b;
                        rq[c] = sn * a + cs * b;
                    }
                    if (lane == 0) atomicAdd(&s_rot, 1);
                }
            }
            __syncthreads();  // workgroup-scope: rows written by one wave are read by another next step
        }
        converged = (s_rot == 0);
        __syncthreads();
    }

    // eigenvalues = row norms of W; rank-sort descending (ties by index)
    for (int i = wave; i < n; i += EIGH_WAVES) {
        const double *wi = W + (int64_t)i * ldw;
        double a = 0.0;
        for (int c = lane; c < n; c += 64) a = fma(wi[c], wi[c], a);
        a = pk_wave_sum(a);
        if (lane == 0) s_lam[i] = sqrt(a);
    }
    __syncthreads();
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
    // permute rows of R into W (scratch), fixing the sign so the largest |component| is positive
    for (int i = wave; i < n; i += EIGH_WAVES) {
        const double *ri = R + (int64_t)i * ldr;
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
        double *dst = W + (int64_t)s_rank[i] * ldw;
        for (int c = lane; c < n; c += 64) dst[c] = sgn * ri[c];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += EIGH_THREADS) {
        int i = e / n, j = e % n;
        R[(int64_t)i * ldr + j] = W[(int64_t)i * ldw + j];
    }
    if (tid == 0 && info) {
        info[0] = sweep;
        info[1] = converged;
    }
}

extern "C" int pk_eigh_psd_f64(void *stream, int32_t n, double *S_dev, int64_t lds_, double *evecs_dev,
                               int64_t ldv, double *evals_dev, int32_t max_sweeps, double tol,
                               int32_t *info_dev) {
    PK_REQUIRE(n >= 1 && n <= 1024, "pk_eigh_psd_f64: n=%d out of range [1,1024]", n);
    PK_REQUIRE(lds_ >= n && ldv >= n, "pk_eigh_psd_f64: bad leading dimension");
    PK_REQUIRE(S_dev && evecs_dev && evals_dev && S_dev != evecs_dev, "pk_eigh_psd_f64: bad pointers");
    if (max_sweeps <= 0) max_sweeps = 40;
    if (tol <= 0.0) tol = 2.0 * 2.220446049250313e-16 * sqrt((double)n);  // ~ LAPACK dgesvj's sqrt(m)*eps
    hipLaunchKernelGGL(eigh_psd_kernel, dim3(1), dim3(EIGH_THREADS), 0, pk_stream(stream), n, S_dev, lds_,
                       evecs_dev, ldv, evals_dev, max_sweeps, tol, info_dev);
    PK_CHECK_LAUNCH("eigh_psd_kernel");
    return PK_OK;
}
