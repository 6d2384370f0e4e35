// K1/K4: CSR x dense (fp64 accumulate) for gfx950.
//
// out[r, :] = sum_p vals[p] * X[indices[p], :]      (scipy csr_matvecs restated for a block of
// nc right-hand sides; reference call sites: models.py:844 via svds' A^T(A x) operator and
// models.py:860 `test_matrix.dot(v)`).
//
// Mapping: one 64-lane wave per task (a contiguous nnz range of ONE row).  Lanes own columns of
// the dense block (lane + 64*g), so every gather of a row of X is a fully coalesced 512-byte
// read; the (index, value) pairs of the task are fetched 64 at a time with one coalesced load and
// broadcast to the wave through v_readlane (SGPR), which makes the X row base address scalar.
// HBM/L2-bound by construction: 1 FMA per 8 bytes gathered.  No atomics: long rows are split into
// tasks that write partial sums, added in slot order by spmm_fixup_kernel (deterministic).
#include "pk_common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

template <typename VT>
__device__ __forceinline__ double pk_bcast_val(VT a, int t);
template <>
__device__ __forceinline__ double pk_bcast_val<float>(float a, int t) {
    return (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), t));
}
template <>
__device__ __forceinline__ double pk_bcast_val<double>(double a, int t) {
    int lo = __builtin_amdgcn_readlane(__double2loint(a), t);
    int hi = __builtin_amdgcn_readlane(__double2hiint(a), t);
    return __hiloint2double(hi, lo);
}

// ACC = true: the launch is one row block of a larger plan (output row = task_row - row_base) and ADDS to out
template <typename VT, int CPL, bool ACC>
__global__ __launch_bounds__(256) void spmm_csr_kernel(
    int64_t n_tasks, const int32_t *__restrict__ task_row, const int64_t *__restrict__ task_begin,
    const int64_t *__restrict__ task_end, const int32_t *__restrict__ task_slot,
    const int32_t *__restrict__ indices, const VT *__restrict__ vals, const double *__restrict__ X,
    int64_t ldx, int nc, double *__restrict__ out, int64_t ldo, double *__restrict__ partial, int64_t row_base) {
    const int lane = threadIdx.x & 63;
    // wave id made provably uniform so the task descriptors live in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    const int64_t p0 = task_begin[task];
    const int64_t p1 = task_end[task];
    if constexpr (ACC) {
        if (p0 == p1 && task_slot[task] < 0) return;   // nothing to add
    }

    int col[CPL];
    double acc[CPL];
#pragma unroll
    for (int g = 0; g < CPL; ++g) {
        int c = lane + 64 * g;
        col[g] = c < nc ? c : nc - 1;  // clamp: out-of-range lanes read a valid column, discard later
        acc[g] = 0.0;
    }

    // Software pipeline: the 64 (index, value) pairs of the NEXT chunk are requested while the
    // current chunk is processed, and within a chunk the row gathers run two groups of 8 deep
    // (group g+1 is issued before group g is consumed), so 8..16 x 512 B gathers per wave are in
    // flight continuously instead of draining to zero after every group.  Lanes >= cnt hold
    // (j = 0, a = 0): padded steps add 0 * X[0, :] and need no branch.
    int j = 0;
    VT a = (VT)0;
    if (p0 + lane < p1) {
        j = indices[p0 + lane];
        a = vals[p0 + lane];
    }
    for (int64_t p = p0; p < p1; p += 64) {
        const int cnt = (int)((p1 - p) < 64 ? (p1 - p) : 64);
        int jn = 0;
        VT an = (VT)0;
        if (p + 64 + lane < p1) {  // prefetch the next chunk's pairs
            jn = indices[p + 64 + lane];
            an = vals[p + 64 + lane];
        }
        double xa[8][CPL], xb[8][CPL];
        auto issue = [&](int t0, double(&x)[8][CPL]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jt = __builtin_amdgcn_readlane(j, t0 + u);
                const double *xr = X + (int64_t)jt * ldx;
#pragma unroll
                for (int g = 0; g < CPL; ++g) x[u][g] = xr[col[g]];
            }
        };
        auto consume = [&](int t0, const double(&x)[8][CPL]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double at = pk_bcast_val<VT>(a, t0 + u);
#pragma unroll
                for (int g = 0; g < CPL; ++g) acc[g] = fma(at, x[u][g], acc[g]);
            }
        };
        issue(0, xa);
#pragma unroll
        for (int g8 = 0; g8 < 64; g8 += 16) {
            if (g8 < cnt) {
                if (g8 + 8 < cnt) issue(g8 + 8, xb);
                consume(g8, xa);
                if (g8 + 8 < cnt) {
                    if (g8 + 16 < cnt) issue(g8 + 16, xa);
                    consume(g8 + 8, xb);
                }
            }
        }
        j = jn;
        a = an;
    }

    const int slot = task_slot[task];
    double *dst = slot < 0 ? out + ((int64_t)task_row[task] - (ACC ? row_base : 0)) * ldo : partial + (int64_t)slot * nc;
    const bool add = ACC && slot < 0;
#pragma unroll
    for (int g = 0; g < CPL; ++g) {
        int c = lane + 64 * g;
        if (c < nc) dst[c] = add ? dst[c] + acc[g] : acc[g];
    }
}

// ---- second mapping: GROUPS nnz per wave step --------------------------------------------------
// The wave is cut into GROUPS groups of LG = 64/GROUPS lanes; group g takes nnz g, g+GROUPS, ... of the
// task and every lane owns FOUR columns (2l, 2l+1, 2LG+2l, 2LG+2l+1), fetched with two 16-byte loads
// that are contiguous across the group.  Per wave step 2 row loads and 4 FMAs serve GROUPS nnz (~3-4
// instructions per nnz for nc <= 64 instead of ~15 in the column-per-lane kernel); the 64 pairs of a
// chunk still arrive through one coalesced load and reach the groups via ds_bpermute.  The group
// partial sums are added in a fixed order at the end (deterministic).  Needs even nc / ldx and a
// 16-byte aligned X, else the column-per-lane kernel runs.
// Measured (S-1M, 1e8 nnz, MI355X): 5x fewer instructions and half the VMEM instructions buy only
// 3-5 % (A.X 3.47 -> 3.28 ms, build SpMM total 351 -> 341 ms); fetching the pairs with per-group
// broadcast loads instead of ds_bpermute was 35 % SLOWER.  The gathers themselves bound this kernel:
// nnz * nc * 8 bytes of 400-512 B row pieces through L2 -> L1 at ~16 TB/s (A.X, X on chip) or from HBM
// at ~7.3 TB/s (A^T.Y) — neither instruction issue nor loads in flight.
// XT = float: the dense block is read in fp32 (the fp32 image of the item factors for the approximate fold-in
// of the scoring pass, scoring.py); a lane's four columns are then consecutive (4l .. 4l+3) and arrive with
// ONE 16-byte load.  Accumulation and output stay fp64.
// OFF32: every byte offset into X fits 32 bits, every row index and the row stride 24 (the launcher checks x_rows): the
// offset of a row piece is one full-rate v_mad_u32_u24 on top of the UNIFORM base pointer, so the loads take the
// scalar-base + 32-bit-offset form.  The 64-bit form costs two quarter-rate 32-bit multiplies, a 64-bit multiply-add
// and three more VALU instructions per gathered row — 12 of the ~33 SIMD cycles a row costs in the fold-in.
// PRED: only the tasks of rows whose flag word intersects `flag_mask` run (row_flags[row]; every other wave leaves at
// once): the exact re-fold of the users a scoring pass could not certify — the flags are where the re-scoring kernel
// left them, on the device, and the plan, the mapping and the summation order are those of the full product.
#ifndef PK_SPMM_U_F64
#define PK_SPMM_U_F64 4        // wave steps per register set of the fp64 dense block (GROUPS <= 8)
#endif
#ifndef PK_SPMM_WPE
#define PK_SPMM_WPE 1          // waves per SIMD the register allocation is held to (1: whatever the kernel needs)
#endif
// one row task on one wave: the body of spmm_csr_groups_kernel (a function of its own so that the list-driven kernel below
// can walk the tasks of LISTED rows with exactly the same mapping, loads and summation order)
template <typename VT, int GROUPS, typename XT, bool ACC, bool OFF32>
__device__ __forceinline__ void spmm_groups_task(
    const int64_t task, const int lane, const int32_t *__restrict__ task_row, const int64_t *__restrict__ task_begin,
    const int64_t *__restrict__ task_end, const int32_t *__restrict__ task_slot,
    const int32_t *__restrict__ indices, const VT *__restrict__ vals, const XT *__restrict__ X,
    int64_t ldx, int nc, double *__restrict__ out, int64_t ldo, double *__restrict__ partial, int64_t row_base) {
    constexpr bool XF = sizeof(XT) == 4;
    constexpr int LG = 64 / GROUPS;
    // wave steps per register set (two sets in flight); an fp32 row piece is one float4 per lane and step
    // (half the registers of the fp64 pair), so twice as many steps fit in flight: fold-in 2.20 -> 1.98 ms.
    // (16 steps: slower, occupancy; 8 steps for the fp64 block: slower too, 3.30 -> 4.07 ms.)
    // narrow instances (round 4): GROUPS = 8 / 16 — 8 / 4 lanes per gathered row for <= 32 / <= 16 fp64 columns, so
    // that a narrow panel does not idle three quarters of every gather instruction's lanes the way GROUPS = 4 does at nc = 16
    // (round 6: the fp32 block also runs on the narrow mappings — 16 / 32 fp32 columns are ONE 16-byte load per lane of a 4- / 8-lane
    // group: the rounded late products of a block Lanczos build, driver.hip::lanczos_step; a chunk holds 64 / GROUPS wave steps
    // and the two register files alternate an even number of sets, so 2 / 4 steps per set there)
    constexpr int U = XF ? (GROUPS <= 4 ? 8 : (GROUPS == 8 ? 4 : 2)) : (GROUPS == 16 ? 2 : PK_SPMM_U_F64);
    using XA = typename std::conditional<XF, float4, double2>::type;
    const int64_t p0 = task_begin[task];
    const int n = (int)(task_end[task] - p0);
    if constexpr (ACC) {
        if (n == 0 && task_slot[task] < 0) return;   // nothing to add
    }
    const int g = lane / LG, l = lane % LG;
    // columns (c0, c0+1) and (c1, c1+1) of this lane; nc is even (a multiple of 4 for fp32 X): a pair is in or
    // out as a whole
    const int c0 = XF ? 4 * l : 2 * l, c1 = XF ? 4 * l + 2 : 2 * LG + 2 * l;
    const bool ok0 = c0 < nc, ok1 = c1 < nc;
    const XT *x0 = X + (ok0 ? c0 : 0);
    const XT *x1 = X + (ok1 ? c1 : 0);
    const int32_t *ip = indices + p0;
    const VT *vp = vals + p0;

    double2 acc0 = make_double2(0.0, 0.0), acc1 = make_double2(0.0, 0.0);
    constexpr int SPC = 64 / GROUPS;      // wave steps per 64-pair chunk
    constexpr int SETS = SPC / U;         // register sets per chunk (U steps each)
    static_assert(SETS >= 2 && SETS % 2 == 0, "the two register files alternate: an even number of sets per chunk");
    // the 64 (index, value) pairs of a chunk are fetched with one coalesced load (next chunk prefetched)
    // and handed to the groups through ds_bpermute: no extra VMEM instruction per step
    int jc = 0, jn = 0;
    VT ac = (VT)0, an = (VT)0;
    if (lane < n) {
        jc = ip[lane];
        ac = vp[lane];
    }
    if (64 + lane < n) {
        jn = ip[64 + lane];
        an = vp[64 + lane];
    }
    const char *Xb = reinterpret_cast<const char *>(X);
    const unsigned stride_b = (unsigned)(ldx * (int64_t)sizeof(XT));           // row stride in bytes, < 2^24 (OFF32)
    const unsigned lo0 = (unsigned)((ok0 ? c0 : 0) * (int)sizeof(XT)), lo1 = (unsigned)((ok1 ? c1 : 0) * (int)sizeof(XT));
    auto issue = [&](int jch, int st0, XA(&xa)[U], auto &xb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = __shfl(jch, (st0 + u) * GROUPS + g, 64);
            if constexpr (OFF32) {
                // byte offset = row * stride + column offset in ONE full-rate instruction (24-bit operands, 32-bit sum);
                // written out because the compiler turns the C expression into a quarter-rate v_mad_u64_u32
                unsigned o0, o1;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(o0) : "v"(jj), "v"(stride_b), "v"(lo0));
                xa[u] = *reinterpret_cast<const XA *>(Xb + o0);
                if constexpr (!XF) {
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(o1) : "v"(jj), "v"(stride_b), "v"(lo1));
                    xb[u] = *reinterpret_cast<const double2 *>(Xb + o1);
                }
            } else {
                const int64_t off = (int64_t)jj * ldx;
                xa[u] = *reinterpret_cast<const XA *>(x0 + off);
                if constexpr (!XF) xb[u] = *reinterpret_cast<const double2 *>(x1 + off);
            }
        }
    };
    auto consume = [&](VT ach, int st0, const XA(&xa)[U], const auto &xb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double aa = (double)__shfl(ach, (st0 + u) * GROUPS + g, 64);
            if constexpr (XF) {
                acc0.x = fma(aa, (double)xa[u].x, acc0.x);
                acc0.y = fma(aa, (double)xa[u].y, acc0.y);
                acc1.x = fma(aa, (double)xa[u].z, acc1.x);
                acc1.y = fma(aa, (double)xa[u].w, acc1.y);
            } else {
                acc0.x = fma(aa, xa[u].x, acc0.x);
                acc0.y = fma(aa, xa[u].y, acc0.y);
                acc1.x = fma(aa, xb[u].x, acc1.x);
                acc1.y = fma(aa, xb[u].y, acc1.y);
            }
        }
    };
    XA xa0[U], xa1[U];
    double2 xb0[XF ? 1 : U], xb1[XF ? 1 : U];
    issue(jc, 0, xa0, xb0);
    for (int p = 0; p < n; p += 64) {
        const int cnt = (n - p) < 64 ? (n - p) : 64;   // pairs of this chunk; padded lanes hold (0, 0.0)
        const bool more = p + 64 < n;
        int jf = 0;
        VT af = (VT)0;
        if (p + 128 + lane < n) {   // pairs two chunks ahead
            jf = ip[p + 128 + lane];
            af = vp[p + 128 + lane];
        }
        // sets alternate between the two register files; a set is skipped when the chunk ends before it
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            const bool have = k * U * GROUPS < cnt;
            const bool have_next = (k + 1 < SETS) ? ((k + 1) * U * GROUPS < cnt) : more;
            if ((k & 1) == 0) {
                if (have_next) {
                    if (k + 1 < SETS) issue(jc, (k + 1) * U, xa1, xb1);
                    else issue(jn, 0, xa1, xb1);
                }
                if (have) consume(ac, k * U, xa0, xb0);
            } else {
                if (have_next) {
                    if (k + 1 < SETS) issue(jc, (k + 1) * U, xa0, xb0);
                    else issue(jn, 0, xa0, xb0);
                }
                if (have) consume(ac, k * U, xa1, xb1);
            }
        }
        jc = jn; ac = an;
        jn = jf; an = af;
    }

    // add the GROUPS partial sums in group order (fixed): after the exchange every lane holds the total
    if constexpr (GROUPS == 16) {
        acc0.x += pk_lane_xor<4>(acc0.x); acc0.y += pk_lane_xor<4>(acc0.y);
        acc1.x += pk_lane_xor<4>(acc1.x); acc1.y += pk_lane_xor<4>(acc1.y);
    }
    if constexpr (GROUPS >= 8) {
        acc0.x += pk_lane_xor<8>(acc0.x); acc0.y += pk_lane_xor<8>(acc0.y);
        acc1.x += pk_lane_xor<8>(acc1.x); acc1.y += pk_lane_xor<8>(acc1.y);
    }
    if constexpr (GROUPS >= 4) {
        // (g0 + g1) and (g2 + g3) first, then the two pairs: the same association in every lane
        acc0.x += pk_lane_xor<16>(acc0.x); acc0.y += pk_lane_xor<16>(acc0.y);
        acc1.x += pk_lane_xor<16>(acc1.x); acc1.y += pk_lane_xor<16>(acc1.y);
    }
    if constexpr (GROUPS >= 2) {
        acc0.x += pk_lane_xor<32>(acc0.x); acc0.y += pk_lane_xor<32>(acc0.y);
        acc1.x += pk_lane_xor<32>(acc1.x); acc1.y += pk_lane_xor<32>(acc1.y);
    }
    const int slot = task_slot[task];
    double *dst = slot < 0 ? out + ((int64_t)task_row[task] - (ACC ? row_base : 0)) * ldo : partial + (int64_t)slot * nc;
    if (g == 0) {
        if (ACC && slot < 0) {
            if (ok0) {
                acc0.x += dst[c0];
                acc0.y += dst[c0 + 1];
            }
            if (ok1) {
                acc1.x += dst[c1];
                acc1.y += dst[c1 + 1];
            }
        }
        if (ok0) {
            dst[c0] = acc0.x;
            dst[c0 + 1] = acc0.y;
        }
        if (ok1) {
            dst[c1] = acc1.x;
            dst[c1 + 1] = acc1.y;
        }
    }
}

template <typename VT, int GROUPS, typename XT, bool ACC, bool OFF32, bool PRED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PK_SPMM_WPE, 8))) void spmm_csr_groups_kernel(
    int64_t n_tasks, const int32_t *__restrict__ task_row, const int64_t *__restrict__ task_begin,
    const int64_t *__restrict__ task_end, const int32_t *__restrict__ task_slot,
    const int32_t *__restrict__ indices, const VT *__restrict__ vals, const XT *__restrict__ X,
    int64_t ldx, int nc, double *__restrict__ out, int64_t ldo, double *__restrict__ partial, int64_t row_base,
    const int32_t *__restrict__ row_flags = nullptr, int flag_mask = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    if constexpr (PRED) {
        if (!(row_flags[task_row[task]] & flag_mask)) return;
    }
    spmm_groups_task<VT, GROUPS, XT, ACC, OFF32>(task, lane, task_row, task_begin, task_end, task_slot, indices, vals, X, ldx, nc, out,
                                                 ldo, partial, row_base);
}

// The product on LISTED rows: list[0 .. *count) (device-side, e.g. the users a scoring pass could not certify), row =
// row_offset + list[i].  A workgroup of PK_LIST_WAVES waves takes listed rows in turn; its waves share the row's tasks
// [row_first_task[row], row_first_task[row + 1]) (wave w: task t0 + w, t0 + w + PK_LIST_WAVES, ...): the users that need
// re-folding are the heavy ones — a 9 000-entry row is nine tasks, and one wave walking them one after the other made the
// whole launch last as long as that row (95 us for 1 267 listed users, first version).  What the flag-predicated launch
// of the full plan costs — a wave per task of EVERY row, 250 K workgroups that leave at once on S-1M: 0.23 ms for 7 650
// listed users — this does not: its grid is the list.
#define PK_LIST_WAVES 8
template <typename VT, int GROUPS, bool OFF32>
__global__ __launch_bounds__(64 * PK_LIST_WAVES) void spmm_csr_rows_list_kernel(
    int64_t cap, const int32_t *__restrict__ list, const int32_t *__restrict__ count, int64_t row_offset,
    const int64_t *__restrict__ row_first_task, const int32_t *__restrict__ task_row, const int64_t *__restrict__ task_begin,
    const int64_t *__restrict__ task_end, const int32_t *__restrict__ task_slot,
    const int32_t *__restrict__ indices, const VT *__restrict__ vals, const double *__restrict__ X,
    int64_t ldx, int nc, double *__restrict__ out, int64_t ldo, double *__restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n = ((int64_t)*count < cap) ? (int64_t)*count : cap;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        // (wave-uniform by construction; said so, so that the task descriptors stay in scalar registers as in the plan-driven kernel)
        const int64_t row = row_offset + __builtin_amdgcn_readfirstlane(list[i]);
        const int64_t t0 = row_first_task[row], t1 = row_first_task[row + 1];
        const int tb = __builtin_amdgcn_readfirstlane((int)t0), te = __builtin_amdgcn_readfirstlane((int)t1);
        for (int64_t t = tb + wave; t < te; t += PK_LIST_WAVES)
            spmm_groups_task<VT, GROUPS, double, false, OFF32>(t, lane, task_row, task_begin, task_end, task_slot, indices, vals, X, ldx,
                                                               nc, out, ldo, partial, 0);
    }
}


// (The persistent fold-in instance with the head of the factor image in LDS — round 4, measured slower: 0.305 against 0.273 ms
// — lives in csrc/experiments/spmm_variants.hip; record: profiles/r04_fold_head_probe_ml20m.txt)
// out[row, :] = sum_{s in [slot_begin, slot_end)} partial[s, :]   (fixed order)
__global__ __launch_bounds__(256) void spmm_fixup_kernel(
    int64_t n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ slot_begin,
    const int32_t *__restrict__ slot_end, const double *__restrict__ partial, int nc,
    double *__restrict__ out, int64_t ldo, int64_t row_base, int accumulate,
    const int32_t *__restrict__ row_flags = nullptr, int flag_mask = 0) {
    const int64_t r = blockIdx.x;
    if (r >= n_long) return;
    if (row_flags && !(row_flags[long_row[r]] & flag_mask)) return;     // (flagged product: this row's tasks did not run)
    const int s0 = slot_begin[r], s1 = slot_end[r];
    double *dst = out + ((int64_t)long_row[r] - row_base) * ldo;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        double acc = 0.0;
        for (int s = s0; s < s1; ++s) acc += partial[(int64_t)s * nc + c];
        dst[c] = accumulate ? dst[c] + acc : acc;
    }
}

template <typename VT>
static int launch_spmm(hipStream_t st, int64_t n_tasks, const int32_t *task_row, const int64_t *task_begin,
                       const int64_t *task_end, const int32_t *task_slot, const int32_t *indices,
                       const void *vals, const void *Xv, int x_kind, int64_t ldx, int nc, double *out, int64_t ldo,
                       double *partial, int64_t row_base, int accumulate, int64_t x_rows) {
    dim3 grid((unsigned)pk_ceil_div(n_tasks, 4)), block(256);
    const VT *v = static_cast<const VT *>(vals);
    if (x_kind == PK_VAL_F32) {
        // fp32 dense block: groups mapping only (one 16-byte load = 4 columns)
        const float *X = static_cast<const float *>(Xv);
        if (!((nc % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X) % 16 == 0))) {
            pk_set_error("pk_spmm_csr_x: an fp32 dense block needs nc %% 4 == 0, ldx %% 4 == 0 and 16-byte alignment");
            return PK_E_UNSUPPORTED;
        }
        // 32-bit offsets: the caller told us how many rows X has, they number fewer than 2^24, a row is a whole number of
        // 16-byte units and the last byte of X lies below 4 GiB
        const bool off32 = x_rows > 0 && x_rows < (1 << 24) && ldx * 4 < (1 << 24) &&
                           x_rows * ldx * 4 < ((int64_t)1 << 32);
#define PK_SPMM_LAUNCH_F(G, A, O)                                                                                   \
    hipLaunchKernelGGL((spmm_csr_groups_kernel<VT, G, float, A, O>), grid, block, 0, st, n_tasks, task_row,         \
                       task_begin, task_end, task_slot, indices, v, X, ldx, nc, out, ldo, partial, row_base)
#define PK_SPMM_GROUPS_F(G)                                                                                          \
    do {                                                                                                             \
        if (accumulate) { if (off32) PK_SPMM_LAUNCH_F(G, true, true); else PK_SPMM_LAUNCH_F(G, true, false); }       \
        else { if (off32) PK_SPMM_LAUNCH_F(G, false, true); else PK_SPMM_LAUNCH_F(G, false, false); }                \
    } while (0)
        if (nc <= 16) PK_SPMM_GROUPS_F(16);
        else if (nc <= 32) PK_SPMM_GROUPS_F(8);
        else if (nc <= 64) PK_SPMM_GROUPS_F(4);
        else if (nc <= 128) PK_SPMM_GROUPS_F(2);
        else PK_SPMM_GROUPS_F(1);
#undef PK_SPMM_GROUPS_F
#undef PK_SPMM_LAUNCH_F
        return PK_OK;
    }
    const double *X = static_cast<const double *>(Xv);
    const bool paired = (nc % 2 == 0) && (ldx % 2 == 0) && (((uintptr_t)X) % 16 == 0);
    if (paired) {
        const bool off32 = x_rows > 0 && x_rows < (1 << 24) && ldx * 8 < (1 << 24) &&
                           x_rows * ldx * 8 < ((int64_t)1 << 32);
#define PK_SPMM_LAUNCH_D(G, A, O)                                                                                   \
    hipLaunchKernelGGL((spmm_csr_groups_kernel<VT, G, double, A, O>), grid, block, 0, st, n_tasks, task_row,        \
                       task_begin, task_end, task_slot, indices, v, X, ldx, nc, out, ldo, partial, row_base)
#define PK_SPMM_GROUPS(G)                                                                                             \
    do {                                                                                                              \
        if (accumulate) { if (off32) PK_SPMM_LAUNCH_D(G, true, true); else PK_SPMM_LAUNCH_D(G, true, false); }        \
        else { if (off32) PK_SPMM_LAUNCH_D(G, false, true); else PK_SPMM_LAUNCH_D(G, false, false); }                 \
    } while (0)
        if (nc <= 16) PK_SPMM_GROUPS(16);
        else if (nc <= 32) PK_SPMM_GROUPS(8);
        else if (nc <= 64) PK_SPMM_GROUPS(4);
        else if (nc <= 128) PK_SPMM_GROUPS(2);
        else PK_SPMM_GROUPS(1);
#undef PK_SPMM_GROUPS
#undef PK_SPMM_LAUNCH_D
        return PK_OK;
    }
    const int cpl = (nc + 63) / 64;
#define PK_SPMM_CASE(C)                                                                                            \
    case C:                                                                                                        \
        if (accumulate)                                                                                            \
            hipLaunchKernelGGL((spmm_csr_kernel<VT, C, true>), grid, block, 0, st, n_tasks, task_row, task_begin,  \
                               task_end, task_slot, indices, v, X, ldx, nc, out, ldo, partial, row_base);          \
        else                                                                                                       \
            hipLaunchKernelGGL((spmm_csr_kernel<VT, C, false>), grid, block, 0, st, n_tasks, task_row, task_begin, \
                               task_end, task_slot, indices, v, X, ldx, nc, out, ldo, partial, row_base);          \
        break;
    switch (cpl) {
        PK_SPMM_CASE(1)
        PK_SPMM_CASE(2)
        PK_SPMM_CASE(3)
        PK_SPMM_CASE(4)
        default:
            pk_set_error("pk_spmm_csr_f64: nc=%d unsupported (max 256)", nc);
            return PK_E_UNSUPPORTED;
    }
#undef PK_SPMM_CASE
    return PK_OK;
}

extern "C" int pk_spmm_csr_ex(void *stream, int64_t n_tasks, const int32_t *task_row_dev,
                              const int64_t *task_begin_dev, const int64_t *task_end_dev,
                              const int32_t *task_slot_dev, int64_t n_long, const int32_t *long_row_dev,
                              const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                              const int32_t *indices_dev, const void *vals_dev, int val_kind,
                              const void *X_dev, int x_kind, int64_t ldx, int32_t nc, double *out_dev, int64_t ldo,
                              double *partial_dev, int64_t row_base, int32_t accumulate, int64_t x_rows) {
    PK_REQUIRE(n_tasks >= 0 && nc >= 1 && nc <= 256, "pk_spmm_csr: bad sizes n_tasks=%lld nc=%d",
               (long long)n_tasks, nc);
    PK_REQUIRE(ldo >= nc && ldx >= nc, "pk_spmm_csr: ldo/ldx < nc");
    PK_REQUIRE(x_kind == PK_VAL_F32 || x_kind == PK_VAL_F64, "pk_spmm_csr_x: bad x_kind %d", x_kind);
    PK_REQUIRE(n_long == 0 || partial_dev != nullptr, "pk_spmm_csr: partial buffer required");
    PK_REQUIRE(row_base >= 0 && (accumulate || row_base == 0), "pk_spmm_csr_ex: a row base needs accumulate (block 0 has base 0)");
    if (n_tasks == 0) return PK_OK;
    hipStream_t st = pk_stream(stream);
    int rc;
    if (val_kind == PK_VAL_F32)
        rc = launch_spmm<float>(st, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev,
                                indices_dev, vals_dev, X_dev, x_kind, ldx, nc, out_dev, ldo, partial_dev, row_base, accumulate, x_rows);
    else if (val_kind == PK_VAL_F64)
        rc = launch_spmm<double>(st, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev,
                                 indices_dev, vals_dev, X_dev, x_kind, ldx, nc, out_dev, ldo, partial_dev, row_base, accumulate, x_rows);
    else {
        pk_set_error("pk_spmm_csr: bad val_kind %d", val_kind);
        return PK_E_INVALID;
    }
    if (rc != PK_OK) return rc;
    PK_CHECK_LAUNCH("spmm_csr_kernel");
    if (n_long > 0) {
        hipLaunchKernelGGL(spmm_fixup_kernel, dim3((unsigned)n_long), dim3(256), 0, st, n_long, long_row_dev,
                           long_slot_begin_dev, long_slot_end_dev, partial_dev, nc, out_dev, ldo, row_base, accumulate);
        PK_CHECK_LAUNCH("spmm_fixup_kernel");
    }
    return PK_OK;
}

extern "C" int pk_spmm_csr_x(void *stream, int64_t n_tasks, const int32_t *task_row_dev,
                             const int64_t *task_begin_dev, const int64_t *task_end_dev,
                             const int32_t *task_slot_dev, int64_t n_long, const int32_t *long_row_dev,
                             const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                             const int32_t *indices_dev, const void *vals_dev, int val_kind,
                             const void *X_dev, int x_kind, int64_t ldx, int32_t nc, double *out_dev, int64_t ldo,
                             double *partial_dev) {
    return pk_spmm_csr_ex(stream, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev, n_long,
                          long_row_dev, long_slot_begin_dev, long_slot_end_dev, indices_dev, vals_dev, val_kind, X_dev,
                          x_kind, ldx, nc, out_dev, ldo, partial_dev, 0, 0, 0);
}

extern "C" int pk_spmm_csr_f64(void *stream, int64_t n_tasks, const int32_t *task_row_dev,
                               const int64_t *task_begin_dev, const int64_t *task_end_dev,
                               const int32_t *task_slot_dev, int64_t n_long, const int32_t *long_row_dev,
                               const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                               const int32_t *indices_dev, const void *vals_dev, int val_kind,
                               const double *X_dev, int64_t ldx, int32_t nc, double *out_dev, int64_t ldo,
                               double *partial_dev) {
    return pk_spmm_csr_ex(stream, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev, n_long,
                          long_row_dev, long_slot_begin_dev, long_slot_end_dev, indices_dev, vals_dev, val_kind, X_dev,
                          PK_VAL_F64, ldx, nc, out_dev, ldo, partial_dev, 0, 0, 0);
}

// ---- the product restricted to flagged rows (fp64 dense block) ---------------------------------------------------------
template <typename VT>
static int launch_spmm_flagged(hipStream_t st, int64_t n_tasks, const int32_t *task_row, const int64_t *task_begin,
                               const int64_t *task_end, const int32_t *task_slot, const int32_t *indices, const void *vals,
                               const double *X, int64_t ldx, int nc, double *out, int64_t ldo, double *partial, int64_t x_rows,
                               const int32_t *row_flags, int flag_mask) {
    dim3 grid((unsigned)pk_ceil_div(n_tasks, 4)), block(256);
    const VT *v = static_cast<const VT *>(vals);
    const bool off32 = x_rows > 0 && x_rows < (1 << 24) && ldx * 8 < (1 << 24) && x_rows * ldx * 8 < ((int64_t)1 << 32);
#define PK_SPMM_LAUNCH_P(G, O)                                                                                          \
    hipLaunchKernelGGL((spmm_csr_groups_kernel<VT, G, double, false, O, true>), grid, block, 0, st, n_tasks, task_row,  \
                       task_begin, task_end, task_slot, indices, v, X, ldx, nc, out, ldo, partial, (int64_t)0, row_flags, flag_mask)
#define PK_SPMM_GROUPS_P(G) do { if (off32) PK_SPMM_LAUNCH_P(G, true); else PK_SPMM_LAUNCH_P(G, false); } while (0)
    if (nc <= 16) PK_SPMM_GROUPS_P(16);
    else if (nc <= 32) PK_SPMM_GROUPS_P(8);
    else if (nc <= 64) PK_SPMM_GROUPS_P(4);
    else if (nc <= 128) PK_SPMM_GROUPS_P(2);
    else PK_SPMM_GROUPS_P(1);
#undef PK_SPMM_GROUPS_P
#undef PK_SPMM_LAUNCH_P
    return PK_OK;
}

extern "C" int pk_spmm_csr_flagged_f64(void *stream, int64_t n_tasks, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                                       const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                                       const int32_t *long_row_dev, const int32_t *long_slot_begin_dev,
                                       const int32_t *long_slot_end_dev, const int32_t *indices_dev, const void *vals_dev,
                                       int val_kind, const double *X_dev, int64_t ldx, int32_t nc, double *out_dev, int64_t ldo,
                                       double *partial_dev, int64_t x_rows, const int32_t *row_flags_dev, int32_t flag_mask) {
    PK_REQUIRE(n_tasks >= 0 && nc >= 2 && nc <= 256 && nc % 2 == 0, "pk_spmm_csr_flagged_f64: nc=%d (even, 2..256)", nc);
    PK_REQUIRE(ldo >= nc && ldx >= nc && ldx % 2 == 0 && (((uintptr_t)X_dev) % 16) == 0, "pk_spmm_csr_flagged_f64: ldx even, X 16-byte aligned");
    PK_REQUIRE(row_flags_dev != nullptr, "pk_spmm_csr_flagged_f64: row flags required");
    PK_REQUIRE(n_long == 0 || partial_dev != nullptr, "pk_spmm_csr_flagged_f64: partial buffer required");
    PK_REQUIRE(val_kind == PK_VAL_F32 || val_kind == PK_VAL_F64, "pk_spmm_csr_flagged_f64: bad val_kind %d", val_kind);
    if (n_tasks == 0) return PK_OK;
    hipStream_t st = pk_stream(stream);
    if (val_kind == PK_VAL_F32)
        launch_spmm_flagged<float>(st, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev, indices_dev, vals_dev, X_dev,
                                   ldx, nc, out_dev, ldo, partial_dev, x_rows, row_flags_dev, flag_mask);
    else
        launch_spmm_flagged<double>(st, n_tasks, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev, indices_dev, vals_dev, X_dev,
                                    ldx, nc, out_dev, ldo, partial_dev, x_rows, row_flags_dev, flag_mask);
    PK_CHECK_LAUNCH("spmm_csr_groups_kernel (flagged rows)");
    if (n_long > 0) {
        hipLaunchKernelGGL(spmm_fixup_kernel, dim3((unsigned)n_long), dim3(256), 0, st, n_long, long_row_dev, long_slot_begin_dev,
                           long_slot_end_dev, partial_dev, nc, out_dev, ldo, (int64_t)0, 0, row_flags_dev, flag_mask);
        PK_CHECK_LAUNCH("spmm_fixup_kernel (flagged rows)");
    }
    return PK_OK;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_spmm() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&spmm_fixup_kernel));
}

// ---- the product on listed rows ------------------------------------------------------------------------------------------
extern "C" int pk_spmm_csr_rows_list_f64(void *stream, int64_t cap, const int32_t *list_dev, const int32_t *count_dev, int64_t row_offset,
                                         const int64_t *row_first_task_dev, const int32_t *task_row_dev, const int64_t *task_begin_dev,
                                         const int64_t *task_end_dev, const int32_t *task_slot_dev, int64_t n_long,
                                         const int32_t *long_row_dev, const int32_t *long_slot_begin_dev, const int32_t *long_slot_end_dev,
                                         const int32_t *indices_dev, const void *vals_dev, int val_kind, const double *X_dev, int64_t ldx,
                                         int32_t nc, double *out_dev, int64_t ldo, double *partial_dev, int64_t x_rows,
                                         const int32_t *row_flags_dev, int32_t flag_mask) {
    PK_REQUIRE(cap >= 1 && nc >= 2 && nc <= 256 && nc % 2 == 0, "pk_spmm_csr_rows_list_f64: nc=%d (even, 2..256)", nc);
    PK_REQUIRE(ldo >= nc && ldx >= nc && ldx % 2 == 0 && (((uintptr_t)X_dev) % 16) == 0, "pk_spmm_csr_rows_list_f64: ldx even, X 16-byte aligned");
    PK_REQUIRE(list_dev && count_dev && row_first_task_dev && task_row_dev, "pk_spmm_csr_rows_list_f64: bad pointers");
    PK_REQUIRE(n_long == 0 || (partial_dev != nullptr && row_flags_dev != nullptr), "pk_spmm_csr_rows_list_f64: split rows need the partial buffer and the row flags");
    PK_REQUIRE(val_kind == PK_VAL_F32 || val_kind == PK_VAL_F64, "pk_spmm_csr_rows_list_f64: bad val_kind %d", val_kind);
    hipStream_t st = pk_stream(stream);
    int64_t wgs = cap;                 // one workgroup per listed row, at most 4 096 of them in flight (the rest in turn)
    if (wgs > 4096) wgs = 4096;
    const dim3 grid((unsigned)wgs), block(64 * PK_LIST_WAVES);
    const bool off32 = x_rows > 0 && x_rows < (1 << 24) && ldx * 8 < (1 << 24) && x_rows * ldx * 8 < ((int64_t)1 << 32);
#define PK_LIST_LAUNCH(VT, G, O)                                                                                             \
    hipLaunchKernelGGL((spmm_csr_rows_list_kernel<VT, G, O>), grid, block, 0, st, cap, list_dev, count_dev, row_offset,      \
                       row_first_task_dev, task_row_dev, task_begin_dev, task_end_dev, task_slot_dev, indices_dev,           \
                       static_cast<const VT *>(vals_dev), X_dev, ldx, nc, out_dev, ldo, partial_dev)
#define PK_LIST_GROUPS(VT, G) do { if (off32) PK_LIST_LAUNCH(VT, G, true); else PK_LIST_LAUNCH(VT, G, false); } while (0)
#define PK_LIST_NC(VT)                                                                                                       \
    do {                                                                                                                     \
        if (nc <= 16) PK_LIST_GROUPS(VT, 16);                                                                                \
        else if (nc <= 32) PK_LIST_GROUPS(VT, 8);                                                                            \
        else if (nc <= 64) PK_LIST_GROUPS(VT, 4);                                                                            \
        else if (nc <= 128) PK_LIST_GROUPS(VT, 2);                                                                           \
        else PK_LIST_GROUPS(VT, 1);                                                                                          \
    } while (0)
    if (val_kind == PK_VAL_F32) PK_LIST_NC(float); else PK_LIST_NC(double);
#undef PK_LIST_NC
#undef PK_LIST_GROUPS
#undef PK_LIST_LAUNCH
    PK_CHECK_LAUNCH("spmm_csr_rows_list_kernel");
    if (n_long > 0) {
        hipLaunchKernelGGL(spmm_fixup_kernel, dim3((unsigned)n_long), dim3(256), 0, st, n_long, long_row_dev, long_slot_begin_dev,
                           long_slot_end_dev, partial_dev, nc, out_dev, ldo, (int64_t)0, 0, row_flags_dev, flag_mask);
        PK_CHECK_LAUNCH("spmm_fixup_kernel (listed rows)");
    }
    return PK_OK;
}
