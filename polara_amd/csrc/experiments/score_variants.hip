// EXPERIMENT TREE — not part of libpolarahip.so.  score.hip as it stood at the end of round 4 (commit c50a3c5 + the per-TU loader of round 5): every sweep / fold-in variant that was built,
// verified against the default kernel and measured slower or mixed (profiles/r03_* r04_*) stays buildable here,
// behind its -D switches and run-time knobs, through `python tools/build_probe_lib.py out.so [-D...]` + POLARA_HIP_LIB=out.so.
// The product sources carry only the kernels that ship.
// K3: fused  scores = E V^T  (fp32 MFMA)  +  seen-item masking  +  per-user top-k candidates.
//
// Replaces the body of RecommenderModel._slice_recommender (models.py:359-371):
//   scores = (test_matrix.dot(v)).dot(v.T)          models.py:860   -> MFMA tiles, never stored
//   downvote_seen_items(scores, slice_data)         models.py:494-519 -> mask bits per 32-item tile
//   get_topk_elements -> apply_along_axis(topsort)  models.py:488-491,563 -> threshold + wave bitonic
//
// Orientation (the "swapped operand" trick): the MFMA computes  C[item][user] = V_tile * E_group^T
// with v_mfma_f32_32x32x2_f32, so in the C layout (col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5))
// a LANE owns ONE user (lane&31) and 16 of the tile's 32 items.  All per-user state — running
// threshold tau, position in the user's seen-tile stream, candidate count — is therefore
// lane-local in registers; no cross-lane traffic on the hot path and no workgroup barrier at all:
// each wave owns a group of 32 users and streams every item tile independently.
//
// Operands: both factor matrices are pre-packed (pk_pack_frag_f32 below) into MFMA fragment order
//   P[tile][q][lane][4]  with element e of lane (i = lane&31, h = lane>>5) = M[32*tile + i][8q + 2e + h]
// so a wave's fragment load is one fully coalesced 1 KiB global_load_dwordx4.  E fragments stay in
// registers for the whole kernel; V fragments stream from L2/MALL (V is n_items x K x 4 B, a few
// tens of MB, shared by every wave of the chip).
//
// Selection: tau = score of the KC-th best candidate seen so far for that user (lazy).  A score
// above tau is appended to a lane-private LDS ring (8 or 16 entries); when a ring is full the wave
// sorts {both rings of the user, current top-KC list} with a key-only bitonic network (two users at a
// time, one per half of the wave, for KC = 16) and refreshes the list and tau.
// After warm-up almost every tile takes the fast path: 8 v_max3 + 1 compare + 1 ballot.
//
// Pruning (exact): |E_u . V_i| <= ||E_u|| ||V_i||.  The caller passes per-user upper bounds of
// ||E_u|| and, per 32-item tile, an upper bound of max ||V_i|| over ALL items from that tile to the end
// of the catalogue (a suffix maximum, so it is non-increasing whatever the item order; it falls
// fastest when the items are ordered by descending norm or popularity).  tau never decreases, so
// once  ||E_u|| * bound[tile] <= tau_u  holds for the 32 users of a wave no later item can enter any
// of their lists: the wave merges its rings, writes its lists and leaves the sweep for good.
// tau is the exact score of the KC-th list entry, but the lists are ordered by keys that drop the low
// 5-7 bits of the score: a tau refreshed later can be smaller by up to 2^-16 relative than an earlier one,
// and a dropped item can exceed it by as much — the re-scoring kernel's certification allows for 2^-15.
#include "pk_common.h"
#include <math.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define RING 16          // lane-private candidate ring entries (8 for KC == 16, see PAIRED below)
#if !defined(PK_SCORE_TWO_BUFFERS) && !defined(PK_SCORE_ROLL)
#define PK_SCORE_ROLL 1  // the rolling fragment buffer of the tile loop (round 4; see score_tile_roll)
#endif

#ifdef PK_SCORE_PROFILE
// tuning builds only: wave-cycles spent in [0] whole kernel, [1] flushes, [2] seen-list walk, [3] push path,
// [4] prologue (state restore), [5] threshold bootstrap; [6] flush count, [7] tiles
__device__ unsigned long long pk_prof[8];
extern "C" int pk_debug_profile(unsigned long long *out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_prof), sizeof(pk_prof));
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(pk_prof), z, sizeof(z));
    }
    return 0;
}
#define PROF_T() __builtin_readcyclecounter()
#define PROF_DECL unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_ADD(i, t0) do { prof_acc[i] += (unsigned long long)(__builtin_readcyclecounter() - (t0)); } while (0)
#define PROF_INC(i, n) do { prof_acc[i] += (unsigned long long)(n); } while (0)
#define PROF_FLUSH() do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&pk_prof[i_], prof_acc[i_]); } while (0)
#else
#define PROF_T() 0ull
#define PROF_DECL
#define PROF_ADD(i, t0) do { (void)(t0); } while (0)
#define PROF_INC(i, n) do { } while (0)
#define PROF_FLUSH() do { } while (0)
#endif
#define PK_IDX_NONE 0x7fffffff
#define PK_IDX_FLOOR (-2)       // last slot of a list: "not full although the sweep started from a threshold" (see the kernel's end)
#define PK_TILE_NONE 0xffffffff00000000ull   // end of a seen-tile stream

// Wave-wide key-only bitonic sort (descending) of 64*SLOTS 32-bit keys, element index i = lane + 64*slot.
// A key is the order-preserving image of a score with its low bits replaced by the element's source index
// (see pk_float_order / the flush code): a stage is a lane exchange + v_max_u32 + v_min_u32 + one v_cndmask
// on a compile-time lane mask, or a plain min/max when the partner is the other slot of the same lane.
template <int K, int J, int S>
constexpr unsigned long long pk_wave_stage_mask() {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int i = l + 64 * S;
        if (((i & J) == 0) == ((i & K) == 0)) m |= 1ull << l;   // this position keeps the larger element
    }
    return m;
}
template <int SLOTS, int K, int J>
__device__ __forceinline__ void pk_sort32_stage(unsigned (&key)[SLOTS]) {
    if constexpr (J >= 64) {
        // partner lives in the other slot of the same lane (SLOTS == 2, J == 64, K == 128: descending)
        const unsigned hi = key[0] > key[1] ? key[0] : key[1], lo = key[0] > key[1] ? key[1] : key[0];
        key[0] = hi;
        key[1] = lo;
    } else {
        unsigned other[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) other[s] = (unsigned)pk_lane_xor<J>((int)key[s]);
        {
            const unsigned hi = key[0] > other[0] ? key[0] : other[0], lo = key[0] > other[0] ? other[0] : key[0];
            key[0] = __builtin_amdgcn_inverse_ballot_w64(pk_wave_stage_mask<K, J, 0>()) ? hi : lo;
        }
        if constexpr (SLOTS == 2) {
            const unsigned hi = key[1] > other[1] ? key[1] : other[1], lo = key[1] > other[1] ? other[1] : key[1];
            key[1] = __builtin_amdgcn_inverse_ballot_w64(pk_wave_stage_mask<K, J, 1>()) ? hi : lo;
        }
    }
}
template <int SLOTS, int K, int J>
__device__ __forceinline__ void pk_sort32_merge(unsigned (&key)[SLOTS]) {
    pk_sort32_stage<SLOTS, K, J>(key);
    if constexpr (J > 1) pk_sort32_merge<SLOTS, K, (J >> 1)>(key);
}
template <int SLOTS, int K>
__device__ __forceinline__ void pk_sort32_levels(unsigned (&key)[SLOTS]) {
    if constexpr (K > 2) pk_sort32_levels<SLOTS, (K >> 1)>(key);
    pk_sort32_merge<SLOTS, K, (K >> 1)>(key);
}
template <int SLOTS>
__device__ __forceinline__ void pk_sort32_wave_desc(unsigned (&key)[SLOTS]) {
    static_assert(SLOTS == 1 || SLOTS == 2, "pk_sort32_wave_desc: 64 or 128 elements");
    pk_sort32_levels<SLOTS, 64 * SLOTS>(key);
}

// Key-only network inside each 32-lane half (both halves end up descending): the element is ONE 32-bit
// word — the score mapped to an order-preserving unsigned with its low 5 bits replaced by the source slot —
// so a stage is a lane exchange + v_max_u32 + v_min_u32 + one v_cndmask on a compile-time lane mask
// (~5 instructions instead of ~12 for a (score, item) pair with a tie-break).  The price: scores that
// agree in all but their low 5 mantissa bits (2^-18 relative; 6 or 7 bits, 2^-16, in the wave-wide sort
// above) may be ordered either way; the exact re-scoring pass widens its certification bound by
// 2^-15 |tau| to cover that (rescore.hip).
template <int K, int J>
constexpr unsigned long long pk_half_stage_mask() {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int t = l & 31;
        if (((t & J) == 0) == ((t & K) == 0)) m |= 1ull << l;   // this lane keeps the larger element
    }
    return m;
}
template <int K, int J>
__device__ __forceinline__ void pk_sort32_half_merge(unsigned &key) {
    const unsigned other = (unsigned)pk_lane_xor<J>((int)key);
    const unsigned hi = key > other ? key : other, lo = key > other ? other : key;
    key = __builtin_amdgcn_inverse_ballot_w64(pk_half_stage_mask<K, J>()) ? hi : lo;
    if constexpr (J > 1) pk_sort32_half_merge<K, (J >> 1)>(key);
}
template <int K>
__device__ __forceinline__ void pk_sort32_half_levels(unsigned &key) {
    if constexpr (K > 2) pk_sort32_half_levels<(K >> 1)>(key);
    pk_sort32_half_merge<K, (K >> 1)>(key);
}
__device__ __forceinline__ unsigned pk_float_order(float x) {   // a > b  <=>  order(a) > order(b)
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Per-lane state carried between the item-chunk launches of one scoring pass (global memory).
struct LaneState {
    int64_t sp;   // position in the user's seen-tile stream
    float tau;    // current threshold
    int cnt;      // entries in the lane's ring; PK_LANE_DONE once the wave has left the sweep (pruned)
};
#define PK_LANE_DONE (-1)

// The candidate scores come from the bf16 matrix cores at fp32-class accuracy ("split bf16"): every fp32 operand is
// stored as two bf16, x = hi + lo + d with |d| <= 2^-16 |x|, |lo| <= 2^-8 |x| (pk_split_bf16; bf16 has 8 significant bits), and a 16-wide k-step of the product is
// THREE v_mfma_f32_32x32x16_bf16 — hi.hi + hi.lo + lo.hi, fp32 accumulation, the lo.lo term (<= 2^-16 |a||b|) dropped.
// gfx950's bf16 MFMA runs at 16x the rate of v_mfma_f32_32x32x2_f32 (which is the fp32 VECTOR rate, MI355X_MICROARCH.md),
// so a rank-50 tile costs 12 x 32 = 384 MFMA cycles instead of 25 x 64 = 1600, a rank-200 tile 39 x 32 instead of
// 100 x 64, at an error of  |s32 - e.v| <= (3 * 2^-16 + (4 K + 10) * 2^-23) ||e|| ||v||  (operand split, dropped term,
// fp32 conversion of the fp64 inputs, at most 4 K + 10 accumulation roundings counted as truncations) that the exact
// fp64 re-scoring pass certifies against (rescore.hip: `bound`).  The C / D layout of the instruction is that of the
// f32 32x32x2 form, so everything after the MFMAs is unchanged.
// NSTEP = number of 16-wide k-steps (rank padded with zeros to 16 * NSTEP, rounded up to a supported value); the packed
// operands hold KQ = 2 * NSTEP 16-byte groups per lane and tile: the eight hi parts, then the eight lo parts of a step.
__host__ __device__ constexpr bool pk_top_in_lds(int nstep, int kc) { return kc <= 32 || nstep > 8; }
__host__ __device__ constexpr int pk_ring_rows(int kc) { return kc == 16 ? 8 : RING; }
__host__ __device__ constexpr size_t pk_score_lds_bytes(int nstep, int kc) {
    return (size_t)(4 * pk_ring_rows(kc) * 64 + (pk_top_in_lds(nstep, kc) ? 4 * 32 * kc : 1)) * sizeof(uint2);
}
// SHARED instance: the rings and lists of its NW waves + TWO packed V tiles (2 * nstep KB each).  NW = 4 at KC = 16: 8 KB of
// selection state per wave + 16 KB of tiles at rank 50 = 48 KB, THREE workgroups per CU; the four waves of a workgroup sit
// on the four SIMDs, so the three waves of a SIMD belong to three workgroups that drift against each other like the
// free-running waves of the register-fed kernel (a first version with 16 waves = one workgroup per CU kept all waves of a
// SIMD in lock-step and lost the MFMA / VALU overlap that way).  PK_SHARED_WAVES: kernel-tuning builds.
#ifndef PK_SHARED_WAVES
#define PK_SHARED_WAVES 4
#endif
__host__ __device__ constexpr int pk_shared_waves(int kc) { return kc == 16 ? PK_SHARED_WAVES : 8; }
__host__ __device__ constexpr size_t pk_score_lds_bytes_shared(int nstep, int kc) {
    return (size_t)(pk_shared_waves(kc) * pk_ring_rows(kc) * 64 + pk_shared_waves(kc) * 32 * kc) * sizeof(uint2) +
           (size_t)2 * (2 * nstep) * 64 * 16;
}
// instantiated for top-10 lists up to rank 128 (the regime it was meant for; it is opt-in — PK_SCORE_SHARED=1 — because it
// did not pay, see DESIGN.md K3 round 3: at best equal to the register-fed kernel on dense sweeps, slower on pruned ones)
__host__ __device__ constexpr bool pk_shared_ok(int nstep, int kc) {
    return kc == 16 && nstep <= 8 && pk_top_in_lds(nstep, kc) && pk_score_lds_bytes_shared(nstep, kc) <= 160 * 1024 - 64;
}

// Dense seen masks of the head of the catalogue (pk_seen_dense_build): mask[(group * tiles + tile) * 32 + user % 32] =
// the 32-bit seen mask of that user in that tile for tile < tiles, ONE coalesced 128-byte load per tile-wave that is
// requested a tile ahead with the V fragments; skip[user] = the records of the user's seen-tile stream that lie in
// those tiles (the stream cursor starts behind them).  tiles == 0: the stream serves every tile.
struct SeenDense {
    const unsigned *mask;
    const int32_t *skip;
    int tiles;
};

// DENSE: the instance that reads them (the other one is the kernel as it was: in the throughput-bound regimes — full
// sweeps, rank 200 — the extra registers and per-tile tests of a run-time switch cost 10 %).
#ifdef PK_SWEEP_WAVES      // kernel-tuning builds: force the register budget of PK_SWEEP_WAVES waves per SIMD
#define PK_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(PK_SWEEP_WAVES, PK_SWEEP_WAVES)))
#elif defined(PK_SCORE_ROLL4)   // with the rolling buffer the rank <= 64, top-10 instances are 8 registers from four waves per SIMD
#define PK_SWEEP_OCC __attribute__((amdgpu_waves_per_eu((NSTEP <= 4 && KC == 16 && !SHARED) ? 4 : 1)))
#elif defined(PK_SCORE_DEPTH2)  // two rolling buffers: the rank <= 64 instances land at 171-172 registers, 3 short of three waves per SIMD
#define PK_SWEEP_OCC __attribute__((amdgpu_waves_per_eu((NSTEP <= 4 && KC <= 32 && !SHARED) ? 3 : 1)))
#else
#define PK_SWEEP_OCC
#endif
// SHARED (round 3, opt-in): the workgroup is pk_shared_waves() waves (four at KC = 16: one per SIMD) that step through the
// item tiles together; the packed V tile of a step is staged ONCE per workgroup in LDS (global_load_lds_dwordx4: no register
// round trip, two buffers, one workgroup barrier per tile) and every wave feeds its MFMAs from there with two ds_read_b128
// per k-step.  What it buys: the two V-tile register buffers (64 VGPRs at rank 50) leave the register file and the V
// traffic out of L2 falls by the number of waves.  What it costs: a barrier per tile in a kernel that has none, waves
// that idle once their group is pruned until the whole workgroup is, lock-step with the slowest wave of a tile.  Never
// ahead of the register-fed kernel in any regime measured.  Single sweeps only (no item splits),
// lists in LDS.
template <int NSTEP, int KC, bool STRIDED, bool DENSE, bool SHARED = false>
__global__ __launch_bounds__(SHARED ? 64 * pk_shared_waves(KC) : 256) PK_SWEEP_OCC void score_candidates_kernel(
    const float4 *__restrict__ Vp, const float4 *__restrict__ Ep, int64_t n_users, int n_items,
    int n_tiles, int split_tiles, int chunk_begin, int chunk_tiles,
    const int64_t *__restrict__ seen_ptr, const unsigned long long *__restrict__ seen_tiles,
    const int32_t *__restrict__ seen_ntiles,
    float *__restrict__ cand_score, int32_t *__restrict__ cand_idx,
    LaneState *__restrict__ st_lane, uint2 *__restrict__ st_ring,
    const float *__restrict__ user_bound, const float *__restrict__ tile_bound, int ablate, SeenDense dense,
    int tile_base, int slot_base, const LaneState *__restrict__ floor_state, int boot_tiles) {
    constexpr int KQ = 2 * NSTEP;   // 16-byte groups per lane and tile: (hi, lo) x 8 bf16 for every 16-wide k-step
    // KC == 16: rings of 8, so that a user's list + both rings are 32 entries and TWO users are merged
    // per flush, one in each half of the wave (15 sort stages per two users instead of 21 per user).
    constexpr bool PAIRED = (KC == 16);
    constexpr int RG = pk_ring_rows(KC);
    constexpr int SLOTS = (2 * RG + KC + 63) / 64;
    // The running top-KC lists of the wave's 32 users live in LDS when they fit (KC <= 32): a flush
    // then never touches global memory (no vmcnt drain in the middle of the MFMA stream).  They are
    // copied from / to cand_score, cand_idx at the launch boundaries.
    // KC = 64 lists (16 KiB per wave) move to LDS too when the rank is high enough that the fragment
    // registers already limit the SIMD to one wave (NSTEP > 8, i.e. rank > 128: 96 KiB per workgroup, one workgroup per CU).
    constexpr bool TOP_LDS = pk_top_in_lds(NSTEP, KC);
    extern __shared__ __attribute__((aligned(16))) uint2 pk_score_lds[];      // [NW][RG][64] rings, then [NW][32*KC] top lists (TOP_LDS), then (SHARED) 2 V tiles
    constexpr int NW = SHARED ? pk_shared_waves(KC) : 4;           // waves (= user groups) per workgroup
    static_assert(!SHARED || (TOP_LDS && !STRIDED), "SHARED: single sweeps with the lists in LDS");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t group_raw = (int64_t)blockIdx.x * NW + wave;
    // SHARED: a wave without users (the tail of the last workgroup) still stages tiles and meets the barriers; it reads
    // as group 0 and never writes
    const bool participate = group_raw * 32 < n_users;
    if (!SHARED && !participate) return;  // whole wave leaves; this form of the kernel has no workgroup barrier
    const int64_t group = participate ? group_raw : 0;
    bool alive = participate;             // SHARED: this wave still sweeps (not pruned, not finished in an earlier launch)
    PROF_DECL;
    const unsigned long long prof_k0 = PROF_T();
    uint2(*ring)[64] = reinterpret_cast<uint2(*)[64]>(pk_score_lds + wave * (RG * 64));
    uint2 *top = pk_score_lds + NW * RG * 64 + (TOP_LDS ? wave * (32 * KC) : 0);
    float4 *vbuf = reinterpret_cast<float4 *>(pk_score_lds + NW * RG * 64 + (TOP_LDS ? NW * (32 * KC) : 0));   // [3][KQ][64]

    // Item split: blockIdx.y = h of S = gridDim.y owns every S-th tile of the catalogue, h, h+S, h+2S, ...
    // (`split_tiles` = ceil(n_tiles / S) of them at most), with its own threshold, rings, top lists and parked
    // state, so that small user counts still fill the chip (the S partial top-KC lists are merged by the
    // re-scoring kernel).  Interleaved, not contiguous ranges: under the exact pruning every split meets the
    // high-norm head of the catalogue first and builds a threshold nearly as good as the single sweep's, so all
    // S of them leave early (a split owning a contiguous slice of the low-norm tail would have to sweep most of
    // it before its own k-th best score beats the norm bound).  A launch sweeps the tiles number
    // [chunk_begin, chunk_begin + chunk_tiles) of every split, i.e. one contiguous L2-sized piece of V.
    // STRIDED = false is the single-sweep instance (S == 1 at compile time: the tile loop, the stream walk and
    // the checkpoint test fold back to their unit-stride forms)
    const int split = STRIDED ? (int)blockIdx.y : 0;
    const int S = STRIDED ? (int)gridDim.y : 1;
    const int64_t n_groups = (n_users + 31) / 32;
    // Two-phase sweep (pk_score_two_phase_f32): a first launch sweeps the head of the catalogue, tiles [0, tile_base), for
    // every group as a single sweep (this kernel with n_tiles = tile_base; lists and thresholds land in slot 0); the
    // splits of the second phase own the tiles tile_base + h, tile_base + h + S, ... and START from the head's threshold
    // (`floor_state` = the head's parked lane records): a score below it cannot be among the user's KC best whatever
    // the split's own list holds, so a split's tau = max(head tau, its own KC-th best) and its list only receives what
    // beats the head's list.  The dependent chain of a group is head + tail / S tiles instead of head + tail at the same
    // number of tile-waves (tools/probes/two_phase_study.py); the S + 1 lists are merged by merge_candidates_kernel.
    const int t_lo = tile_base + split;
    const int tile_begin = t_lo + chunk_begin * S;
    const int tile_stop = (chunk_begin + chunk_tiles < split_tiles) ? t_lo + (chunk_begin + chunk_tiles) * S : n_tiles;
    const int tile_end = (tile_stop < n_tiles) ? tile_stop : n_tiles;   // tiles tile_begin, +S, ... < tile_end
    const bool first = (chunk_begin == 0);
    const bool last = (tile_stop >= n_tiles);
    if (!first && tile_begin >= n_tiles) return;  // this split finished in an earlier launch

    const int ul = lane & 31, hi = lane >> 5;
    const int64_t user = group * 32 + ul;
    const int64_t slot = (int64_t)(slot_base + split) * n_groups + group;
    float *my_score = cand_score + slot * 32 * KC;  // this wave's [32][KC] top lists of this split
    int32_t *my_idx = cand_idx + slot * 32 * KC;

    float4 e[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) e[q] = Ep[(group * KQ + q) * 64 + lane];

    int64_t sp = 0, se = 0;
    // next three (tile, mask) records of the user's seen-tile stream (prefetch window)
    unsigned long long nxt = PK_TILE_NONE, nxt2 = PK_TILE_NONE, nxt3 = PK_TILE_NONE;
    float tau = -INFINITY;
    int cnt = 0;
    const bool prune = (user_bound != nullptr && tile_bound != nullptr) && !(ablate & 4);
    // padding lanes of the last group never keep the wave in the sweep
    const float en = (prune && user < n_users) ? user_bound[user] : -1.0f;
    bool pruned = false;
    int exit_tile = tile_end;
    const bool has_seen = (seen_ptr != nullptr && user < n_users);
    if (has_seen) {
        sp = seen_ptr[user];
        se = sp + seen_ntiles[user];
    }
    LaneState *my_state = st_lane + slot * 64 + lane;
    uint2 *my_ring_state = st_ring + slot * (RING * 64);
    float tau_floor = -INFINITY;
    if constexpr (STRIDED) {
        if (floor_state) {
            const LaneState fs = floor_state[group * 64 + lane];     // the head sweep's record of my user (slot 0)
            if ((int)fs.sp < tile_base) {
                // the group was pruned INSIDE the head (wave-uniform: every lane of the head wave wrote its exit tile):
                // nothing beyond it can enter its lists — this split's list stays empty
                if (first) {
                    for (int s = lane; s < 32 * KC; s += 64) {
                        my_score[s] = -INFINITY;
                        my_idx[s] = -1;
                    }
                    LaneState ls;
                    ls.sp = fs.sp;
                    ls.tau = fs.tau;
                    ls.cnt = PK_LANE_DONE;
                    *my_state = ls;
                }
                return;
            }
            tau_floor = fs.tau;
            tau = tau_floor;
        }
    }
    const int dense_tiles = (DENSE && seen_ptr != nullptr) ? dense.tiles : 0;
    const unsigned *dense_row = dense_tiles ? dense.mask + ((int64_t)group * dense_tiles) * 32 + ul : nullptr;
    if (DENSE && first && has_seen && dense_tiles) sp += dense.skip[user];   // those tiles are served by the dense masks
    if (first) {
        if (has_seen && t_lo > 0 && !(DENSE && dense_tiles >= t_lo)) {   // (behind the dense window the cursor already is)
            // skip the records before this split's first tile: lower_bound(tile >= t_lo)
            int64_t lo = sp, hi_ = se;
            while (lo < hi_) {
                const int64_t mid = (lo + hi_) >> 1;
                if ((unsigned)(seen_tiles[mid] >> 32) < (unsigned)t_lo) lo = mid + 1; else hi_ = mid;
            }
            sp = lo;
        }
        for (int s = lane; s < 32 * KC; s += 64) {
            if (TOP_LDS) {
                top[s] = make_uint2(__float_as_uint(-INFINITY), 0xffffffffu);
            } else {
                my_score[s] = -INFINITY;
                my_idx[s] = -1;
            }
        }
    } else {
        // resume: restore the lane state and the ring image written by the previous chunk launch
        const LaneState ls = *my_state;
        if (ls.cnt == PK_LANE_DONE) {        // wave-uniform: this group was pruned in an earlier launch
            if constexpr (SHARED) alive = false;
            else return;
        }
        if (TOP_LDS)
            for (int s = lane; s < 32 * KC; s += 64) top[s] = make_uint2(__float_as_uint(my_score[s]), (unsigned)my_idx[s]);
        if (has_seen) sp = ls.sp;
        tau = ls.tau;
        tau_floor = fmaxf(tau_floor, tau);   // a threshold once reached stays a lower bound of the final KC-th best
        cnt = ls.cnt;
        const int cmax = __builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, cnt));
        for (int i = 0; i < cmax; ++i) ring[i][lane] = my_ring_state[i * 64 + lane];
    }
    if (has_seen) {
        if (sp < se) nxt = seen_tiles[sp];
        if (sp + 1 < se) nxt2 = seen_tiles[sp + 1];
        if (sp + 2 < se) nxt3 = seen_tiles[sp + 2];
    }

    // merge the rings of user x (lanes x, x+32) and its top list; refresh list, tau, counters
    auto flush_user = [&](int x) {
        const unsigned long long prof_f0 = PROF_T();
        PROF_INC(6, 1);
        if (TOP_LDS) {
            // LDS only: program order within the wave + the LDS pipe's in-order execution suffice;
            // the barrier keeps the compiler from moving LDS accesses across the hand-off
            __builtin_amdgcn_wave_barrier();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        const int c_lo = __builtin_amdgcn_readlane(cnt, x);
        const int c_hi = __builtin_amdgcn_readlane(cnt, x + 32);
        if (!TOP_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float key[SLOTS];
        int val[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = lane + 64 * s;
            float k = -INFINITY;
            int v = PK_IDX_NONE;
            if (i < RG) {
                if (i < c_lo) {
                    uint2 r = ring[i][x];
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (i < 2 * RG) {
                if (i - RG < c_hi) {
                    uint2 r = ring[i - RG][x + 32];
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (i < 2 * RG + KC) {
                const int t = i - 2 * RG;
                if (TOP_LDS) {
                    const uint2 r = top[x * KC + t];
                    if ((int)r.y >= 0) {
                        k = __uint_as_float(r.x);
                        v = (int)r.y;
                    }
                } else {
                    const int iv = my_idx[x * KC + t];
                    if (iv >= 0) {
                        k = my_score[x * KC + t];
                        v = iv;
                    }
                }
            }
            key[s] = k;
            val[s] = v;
        }
        // key-only sort: low SRC_BITS of the ordered score carry the element's source index (lane + 64*slot);
        // afterwards every position fetches its (score, item) from that source
        constexpr unsigned SRC_MASK = 64u * SLOTS - 1u;
        unsigned k32[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) k32[s] = (pk_float_order(key[s]) & ~SRC_MASK) | (unsigned)(lane + 64 * s);
        pk_sort32_wave_desc<SLOTS>(k32);
        {
            float ok[SLOTS];
            int ov[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int src = (int)(k32[s] & SRC_MASK);
                float fk = __shfl(key[0], src & 63, 64);
                int fv = __shfl(val[0], src & 63, 64);
                if constexpr (SLOTS == 2) {
                    const float fk1 = __shfl(key[1], src & 63, 64);
                    const int fv1 = __shfl(val[1], src & 63, 64);
                    if (src >= 64) {
                        fk = fk1;
                        fv = fv1;
                    }
                }
                ok[s] = fk;
                ov[s] = fv;
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                key[s] = ok[s];
                val[s] = ov[s];
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = lane + 64 * s;
            if (i < KC) {
                const int iv = (val[s] == PK_IDX_NONE) ? -1 : val[s];
                if (TOP_LDS) {
                    top[x * KC + i] = make_uint2(__float_as_uint(key[s]), (unsigned)iv);
                } else {
                    my_score[x * KC + i] = key[s];
                    my_idx[x * KC + i] = iv;
                }
            }
        }
        // new threshold: key of sorted element KC-1 (-inf while the list is not full)
        const float ntau = __int_as_float(
            __builtin_amdgcn_readlane(__float_as_int(key[(KC - 1) / 64]), (KC - 1) & 63));
        if (ul == x) {
            tau = fmaxf(ntau, tau_floor);
            cnt = 0;
        }
        __builtin_amdgcn_wave_barrier();
        PROF_ADD(1, prof_f0);
    };
    // KC == 16: users x (lanes 0..31) and y (lanes 32..63; y < 0: nobody) merged in one 32-wide sort each
    auto flush_pair = [&](int x, int y) {
        const unsigned long long prof_f0 = PROF_T();
        PROF_INC(6, 1);
        __builtin_amdgcn_wave_barrier();
        const int t = lane & 31;
        const int yy = (y >= 0) ? y : x;
        const int cx_lo = __builtin_amdgcn_readlane(cnt, x), cx_hi = __builtin_amdgcn_readlane(cnt, x + 32);
        const int cy_lo = __builtin_amdgcn_readlane(cnt, yy), cy_hi = __builtin_amdgcn_readlane(cnt, yy + 32);
        const int u = hi ? yy : x;
        const int c_lo = hi ? cy_lo : cx_lo, c_hi = hi ? cy_hi : cx_hi;
        const bool act = !hi || y >= 0;
        float k = -INFINITY;
        int v = PK_IDX_NONE;
        if (act) {
            if (t < KC) {
                const uint2 r = top[u * KC + t];
                if ((int)r.y >= 0) {
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (t < KC + RG) {
                if (t - KC < c_lo) {
                    const uint2 r = ring[t - KC][u];
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (t - KC - RG < c_hi) {
                const uint2 r = ring[t - KC - RG][u + 32];
                k = __uint_as_float(r.x);
                v = (int)r.y;
            }
        }
        unsigned key = (pk_float_order(k) & ~31u) | (unsigned)t;   // low 5 bits: the slot the element came from
        pk_sort32_half_levels<32>(key);
        const int src = (lane & 32) | (int)(key & 31u);
        k = __shfl(k, src, 64);
        v = __shfl(v, src, 64);
        if (act && t < KC) top[u * KC + t] = make_uint2(__float_as_uint(k), (unsigned)((v == PK_IDX_NONE) ? -1 : v));
        const float tx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), KC - 1));
        const float ty = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 32 + KC - 1));
        if (ul == x) {
            tau = fmaxf(tx, tau_floor);
            cnt = 0;
        }
        if (y >= 0 && ul == y) {
            tau = fmaxf(ty, tau_floor);
            cnt = 0;
        }
        __builtin_amdgcn_wave_barrier();
        PROF_ADD(1, prof_f0);
    };
    // merge every user of the bit set `um`
    auto flush_set = [&](unsigned um) {
        if constexpr (PAIRED) {
            while (um) {
                const int x = __builtin_ctz(um);
                um &= um - 1;
                int y = -1;
                if (um) {
                    y = __builtin_ctz(um);
                    um &= um - 1;
                }
                flush_pair(x, y);
            }
        } else {
            while (um) {
                const int x = __builtin_ctz(um);
                um &= um - 1;
                flush_user(x);
            }
        }
    };

    // ---- building blocks of the tile pipeline --------------------------------------------------------
    auto load_frags = [&](int tile, float4(&dst)[KQ]) {
        const float4 *vp = Vp + ((int64_t)tile * KQ) * 64 + lane;
#pragma unroll
        for (int q = 0; q < KQ; ++q) dst[q] = vp[q * 64];
    };
    // seen-item mask of a tile for my user (bit b <-> item 32*tile + b); advances the stream cursor.
    // The stream holds ONE record per tile the user has seen items in (pk_seen_tiles_build), so this
    // is at most one step per tile — no inner loop, and the record is requested three records ahead
    // of its use.  (The first version walked the raw item list: in the popular head of the catalogue
    // a user has several seen items per tile, every one a dependent load — the sweep was latency
    // bound there once pruning had cut it down to the head.)
    unsigned m_dense = 0;        // dense mask of the CURRENT tile (requested one tile ahead, see the loop)
    auto walk_mask = [&](int tile) -> unsigned {
        const int j0 = tile * 32, jend = j0 + 32;
        unsigned mask = 0;
        if (ablate & 1) return 0u;   // tuning only: skip the seen-list walk
        if constexpr (DENSE) {
            if (tile < dense_tiles) {
                mask = m_dense;
                if (jend > n_items) mask |= ~0u << (n_items - j0);
                return mask;
            }
        }
        if (S > 1) {
            // records of tiles that belong to the other splits lie between two of mine: step over them
            // (wave-uniform test; the three-record prefetch window keeps the common one-or-two steps cheap)
            while (__any((unsigned)(nxt >> 32) < (unsigned)tile)) {
                if ((unsigned)(nxt >> 32) < (unsigned)tile) {
                    ++sp;
                    nxt = nxt2;
                    nxt2 = nxt3;
                    nxt3 = (sp + 2 < se) ? seen_tiles[sp + 2] : PK_TILE_NONE;
                }
            }
        }
        const bool hit = (unsigned)(nxt >> 32) == (unsigned)tile;
        if (__any(hit)) {
            if (hit) {
                mask = (unsigned)nxt;
                ++sp;
                nxt = nxt2;
                nxt2 = nxt3;
                nxt3 = (sp + 2 < se) ? seen_tiles[sp + 2] : PK_TILE_NONE;
            }
        }
        if (jend > n_items) mask |= ~0u << (n_items - j0);  // padding items of the last tile
        return mask;
    };
    // rare path: append the scores above tau to the lane rings (flushing full rings first)
    auto push_candidates = [&](const float(&acc)[16], int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bool c = acc[r] > tau;
            if (__any(c)) {
                if (__any(c && cnt == RG)) {
                    const unsigned long long full = __ballot(cnt == RG);
                    unsigned um = (unsigned)(full | (full >> 32));
                    if (PAIRED && (__builtin_popcount(um) & 1)) {
                        // the free half of the last sort takes a user whose rings are at least half full
                        const unsigned long long part = __ballot(2 * cnt >= RG);
                        const unsigned cand = (unsigned)(part | (part >> 32)) & ~um;
                        if (cand) um |= 1u << __builtin_ctz(cand);
                    }
                    flush_set(um);
                    c = acc[r] > tau;
                }
                if (c) {
                    ring[cnt][lane] = make_uint2(__float_as_uint(acc[r]),
                                                 (unsigned)(j0 + (r & 3) + 8 * (r >> 2) + 4 * hi));
                    ++cnt;
                }
            }
        }
    };
    // ---- threshold bootstrap ---------------------------------------------------------------------------------------
    // A cold threshold makes the first tiles of a sweep the expensive ones: nearly every score is pushed, the rings fill
    // and the flush sorts run (ML-20M-shaped: 75 of a user's 90 pushes and almost all of its ~7 flushes fall into the
    // first 8 tiles; the 32-tile head took 242 of the sweep's 368 us, profiles/r03_trace_pass_*).  So the first
    // `boot_tiles` tiles are scored once WITHOUT any selection: every lane keeps the KC / 2 largest of the maxima of
    // KC / 8 groups of its 16 scores per tile in a sorted register list (branch-free insertion: a v_max + v_min per
    // list entry and value).  The KC values of a user's two lanes belong to KC different unseen items, so the smallest
    // of them is a lower bound of the user's final KC-th best score: the sweep
    // then starts from that threshold (just below it, so that those items themselves are pushed and the list fills) and
    // re-scores the same tiles — 12 MFMAs each — pushing a quarter of what it pushed from a cold start
    // (tools/probes/warmup_study.py: 27 instead of 86-92 pushes, 1.6 instead of 6-7 flushes per user at 16 tiles).
    // SHARED: wave w of the workgroup brings groups w, w + NW, ... of the packed tile into buffer `buf` (one
    // global_load_lds_dwordx4 per group: a wave writes 64 x 16 contiguous bytes, the [q][lane] layout the MFMA operands
    // are read back in); stage_wait: my loads have landed and everybody's are visible
    auto stage_tile = [&](int tile, int buf) {
        if constexpr (SHARED) {
            const float4 *src = Vp + ((int64_t)tile * KQ) * 64 + lane;
            for (int q = wave; q < KQ; q += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * 64),
                                                 (__attribute__((address_space(3))) void *)(vbuf + (buf * KQ + q) * 64), 16, 0, 0);
        }
    };
    // Two buffers: behind the barrier that ends iteration t - 1 every wave has finished reading tile t - 1 and tile t is
    // complete; a wave then requests its share of tile t + 1 FIRST in iteration t (into the buffer of t - 1), works on
    // tile t, and waits for its own loads only at the end of the iteration — a whole tile later — in front of the barrier.
    auto stage_wait = [&]() {
        if constexpr (SHARED) {
#ifdef PK_SCORE_DIAG       // kernel-tuning builds: ablate & 64 = neither the wait nor the barrier (wrong scores; what would the LDS-fed loop cost if its waves ran free?)
            if (ablate & 64) return;
#endif
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) only: expcnt and lgkmcnt fields left at their maxima (gfx9 encoding)
            __syncthreads();
        }
    };
    // the score tile of one step: split-bf16 product (see the header): per 16-wide k-step  hi.hi + hi.lo + lo.hi, fp32
    // accumulation — ONE instruction sequence for the bootstrap and the sweep, registers or LDS: the same bits
    auto score_tile = [&](const float4(&a)[SHARED ? 1 : KQ], int buf) -> f32x16 {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float4 *vb = vbuf + (buf * KQ) * 64 + lane;
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; ++sidx) {
            float4 fh, fl;
            if constexpr (SHARED) {
                fh = vb[(2 * sidx) * 64];
                fl = vb[(2 * sidx + 1) * 64];
            } else {
                fh = a[2 * sidx];
                fl = a[2 * sidx + 1];
            }
            const bf16x8 vh = __builtin_bit_cast(bf16x8, fh), vl = __builtin_bit_cast(bf16x8, fl);
            const bf16x8 eh = __builtin_bit_cast(bf16x8, e[2 * sidx]), el = __builtin_bit_cast(bf16x8, e[2 * sidx + 1]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh, acc, 0, 0, 0);
        }
        return acc;
    };
#ifdef PK_SCORE_ROLL
    // Round 4: ONE fragment buffer.  The registers of k-step s are re-requested for the next tile right after the three MFMAs
    // that read them: KQ instead of 2 KQ fragment registers (rank 50: 166 -> 136 VGPRs, rank 100: 243 -> 186), no
    // `s_waitcnt vmcnt(0)` + sixteen v_mov_b64 (a <- a_nxt) at the end of every tile — the compiler now waits per k-step with
    // vmcnt(7) / vmcnt(6) on loads issued a whole tile earlier (round 1 saw it drain vmcnt(0) at the loop head with this
    // form; ROCm 7.2 does not).  Measured (bench.py, same box, two-buffer -> rolling): rank 100 / top-20 1.97 -> 1.63 ms per
    // pass (70 -> 85 M users/s), no-prune 71 -> 75 M, pop^0.25 73 -> 80 M, flat-norm 61 -> 63 M users/s, pruned headline sweep
    // 0.378 -> 0.377 ms (it is not bound by the tile loop).  Forcing the rank-50 instance from 136 to 128 registers for four
    // waves per SIMD (PK_SCORE_ROLL=4 builds: 4 spills) made everything slower again (no-prune 75 -> 68 M): occupancy is not
    // what this kernel lacks.  -DPK_SCORE_TWO_BUFFERS builds the old loop.
    auto score_tile_roll = [&](float4(&a)[SHARED ? 1 : KQ], int next_tile) -> f32x16 {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float4 *vp = Vp + ((int64_t)next_tile * KQ) * 64 + lane;
#ifdef PK_SCORE_TWO_CHAINS     // kernel-tuning builds: even and odd k-steps accumulate into two independent chains (rank <= 64)
        // (the rolling buffer left room for a second accumulator at three waves per SIMD: 152 registers.  MEASURED, full sweep of
        // ML-20M-shaped: 1.38 -> 1.41 ms without pushes, 1.61 -> 1.67 with them — the dependent accumulator chain is not what
        // the tile loop waits for either.  Not the default.)
        constexpr bool TWO = (NSTEP >= 2 && NSTEP <= 4);
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
#endif
#pragma unroll
        for (int sidx = 0; sidx < (SHARED ? 0 : NSTEP); ++sidx) {
            const bf16x8 vh = __builtin_bit_cast(bf16x8, a[2 * sidx]), vl = __builtin_bit_cast(bf16x8, a[2 * sidx + 1]);
            const bf16x8 eh = __builtin_bit_cast(bf16x8, e[2 * sidx]), el = __builtin_bit_cast(bf16x8, e[2 * sidx + 1]);
#ifdef PK_SCORE_DIAG       // kernel-tuning builds: ablate & 32 = no products (the fragments are still waited for), & 16 = no re-loads
            if (ablate & 32) {
                acc[sidx] += a[2 * sidx].x + a[2 * sidx + 1].y;
            } else
#endif
            {
#ifdef PK_SCORE_TWO_CHAINS
                if (NSTEP > 8) {
                    // round 5, ranks above 128 (ONE wave per SIMD: nobody else fills the matrix core while a dependent MFMA
                    // waits for its accumulator): consecutive MFMAs alternate between the two chains
                    if (sidx & 1) {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh, acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh, acc1, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh, acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el, acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh, acc, 0, 0, 0);
                    }
                } else if (TWO && (sidx & 1)) {
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh, acc1, 0, 0, 0);
                } else
#endif
                {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh, acc, 0, 0, 0);
                }
            }
#ifdef PK_SCORE_DIAG
            if (ablate & 16) continue;
#endif
            a[2 * sidx] = vp[(2 * sidx) * 64];
            a[2 * sidx + 1] = vp[(2 * sidx + 1) * 64];
        }
#ifdef PK_SCORE_TWO_CHAINS
        if (TWO || NSTEP > 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        }
#endif
        return acc;
    };
#endif
    if (first && boot_tiles > 0 && floor_state == nullptr && !(ablate & 8)) {
        const unsigned long long prof_b0 = PROF_T();
        constexpr int BL = KC / 2;      // values kept per lane: the user's two lanes hold KC of them
        constexpr int BG = KC / 8;      // groups a lane's 16 scores of a tile are cut into (2, 4, 8): BL values after 4 tiles
        float bl[BL];
#pragma unroll
        for (int i = 0; i < BL; ++i) bl[i] = -INFINITY;
        const int64_t sp0 = sp;
        const unsigned long long n0 = nxt, n1 = nxt2, n2 = nxt3;
        float4 a_nxt[SHARED ? 1 : KQ];
        if constexpr (SHARED) {
            stage_tile((tile_begin < n_tiles) ? tile_begin : 0, 0);
            stage_wait();
        } else {
            load_frags((tile_begin < n_tiles) ? tile_begin : 0, a_nxt);
        }
        unsigned m_nxt = 0u;
        if constexpr (DENSE) m_nxt = (tile_begin < dense_tiles) ? dense_row[(int64_t)tile_begin * 32] : 0u;
        // (trip count uniform over the workgroup: SHARED has a barrier per tile)
        for (int i = 0, tile = tile_begin; i < boot_tiles && tile < tile_end; ++i, tile += S) {
#ifndef PK_SCORE_ROLL
            float4 a[SHARED ? 1 : KQ];
#endif
            if constexpr (SHARED) {
                if (i + 1 < boot_tiles && tile + S < tile_end) stage_tile(tile + S, (i + 1) & 1);
            }
#ifndef PK_SCORE_ROLL
            if constexpr (!SHARED) {
#pragma unroll
                for (int q = 0; q < KQ; ++q) a[q] = a_nxt[q];
                load_frags((tile + S < tile_end) ? tile + S : tile, a_nxt);
            }
#endif
            if (!SHARED || alive) {
                if constexpr (DENSE) {
                    m_dense = m_nxt;
                    m_nxt = (tile + S < dense_tiles) ? dense_row[(int64_t)(tile + S) * 32] : 0u;
                }
#ifdef PK_SCORE_ROLL
                f32x16 acc;
                if constexpr (SHARED) acc = score_tile(a_nxt, i & 1);
                else acc = score_tile_roll(a_nxt, (tile + S < tile_end) ? tile + S : tile);
#else
                const f32x16 acc = score_tile(a, i & 1);
#endif
                const unsigned m2 = walk_mask(tile) >> (4 * hi);
#pragma unroll
                for (int g = 0; g < BG; ++g) {
                    float x = -INFINITY;
#pragma unroll
                    for (int r = g * (16 / BG); r < (g + 1) * (16 / BG); ++r)
                        x = fmaxf(x, (m2 & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : acc[r]);
#pragma unroll
                    for (int i2 = 0; i2 < BL; ++i2) {      // sorted insertion, descending
                        const float up = fmaxf(bl[i2], x);
                        x = fminf(bl[i2], x);
                        bl[i2] = up;
                    }
                }
            }
            if constexpr (SHARED) stage_wait();
        }
        // the (KC / 2)-th value of each lane: together at least KC items of the user score that much
        float t0 = bl[KC / 2 - 1];
        t0 = fminf(t0, __int_as_float(pk_lane_xor<32>(__float_as_int(t0))));
        if (t0 > -INFINITY) {
            // strictly below it: the items that define it must pass the `score > tau` test of the sweep
            t0 = fminf(t0 - fabsf(t0) * 2.4e-7f, t0 - 1e-37f);
            tau_floor = fmaxf(tau_floor, t0);
            tau = fmaxf(tau, tau_floor);
        }
        sp = sp0;
        nxt = n0;
        nxt2 = n1;
        nxt3 = n2;
        PROF_ADD(5, prof_b0);
    }
    {
        // One tile per iteration; the fragments of the NEXT tile are requested before this tile's
        // MFMAs so their L2 latency hides behind them.  MFMA/epilogue overlap comes from the other
        // waves of the SIMD (3 per SIMD at 140 VGPRs).  An in-wave software pipeline (MFMAs of tile
        // t+1 interleaved 1:3 with the epilogue VALU of tile t via sched_group_barrier) was measured
        // SLOWER on MI355X (109 ms vs 97 ms per 1M x 100K pass): it drops occupancy to 2 waves/SIMD.
        // Also measured and rejected: two alternating fragment buffers with the tile body unrolled
        // twice (no a <- a_nxt moves): 112 ms; one shared copy of the flush code with a resumable
        // scan instead of 16 inlined copies: 95 ms vs 90 ms; a single rolling fragment buffer (group q
        // of the next tile loaded right after group q's MFMAs, no register moves): 99 vs 81 ms — the
        // compiler drains vmcnt(0) at the loop head, which exposes the latency of the late loads.
        // Forcing five waves per SIMD (amdgpu_waves_per_eu: 96 VGPRs, 12 spilled) on the pruned sweep: 3.47 vs
        // 3.29 ms; four (120 VGPRs, no spill) is what the register allocator picks unprompted.
#if defined(PK_SCORE_ROLL) && defined(PK_SCORE_DEPTH2)
        if constexpr (!SHARED) {
            // Round 4, depth 2 (PK_SCORE_DEPTH2 builds only): TWO rolling fragment buffers that alternate from tile to tile (the
            // tile body is instantiated twice; no copies): the registers a k-step of tile t has just read are re-requested for
            // tile t + 2, so every fragment load has TWO tiles of compute to land in and a wave keeps 16 KB instead of 8 KB in
            // flight (168 VGPRs, three waves per SIMD).  MEASURED against the one-buffer loop: pruned headline sweep 0.377 ->
            // 0.355 ms, but no-prune 74 -> 69 M users/s, flat-norm 63.5 -> 60.4 M, pop^0.25 79 -> 77 M, and the library grows by
            // half (two copies of the push / flush code per instance): the full sweeps already move their 30 GB of fragments at
            // the ~17 TB/s the L1 path gives (the ceiling of the SpMM gathers as well) — more requests in flight only queue.
            // Not the default.
            float4 aA[KQ], aB[KQ];
            {
                const int t0 = (tile_begin < n_tiles) ? tile_begin : 0;
                const int t1 = (tile_begin + S < tile_end) ? tile_begin + S : t0;
                load_frags(t0, aA);
                load_frags(t1, aB);
            }
            unsigned m_nxt = 0u, m_nxt2 = 0u;
            if constexpr (DENSE) {
                m_nxt = (tile_begin < dense_tiles) ? dense_row[(int64_t)tile_begin * 32] : 0u;
                m_nxt2 = (tile_begin + S < dense_tiles) ? dense_row[(int64_t)(tile_begin + S) * 32] : 0u;
            }
            float tb = prune ? tile_bound[(tile_begin < n_tiles) ? tile_begin : 0] : 0.0f;
            PROF_ADD(4, prof_k0);
            auto tile_body = [&](int tile, int step, float4(&a)[KQ]) -> bool {
                if (prune) {
                    const bool open = en * tb > tau;
                    const unsigned long long ob = __ballot(open);
                    if (ob == 0ull) {
                        pruned = true;
                        exit_tile = tile;
                        return false;
                    }
                    if (((STRIDED ? step : tile) & 7) == 7 && __popcll(ob) <= 16) {
                        const unsigned long long pend = __ballot(cnt > 0);
                        flush_set((unsigned)(ob | (ob >> 32)) & (unsigned)(pend | (pend >> 32)));
                    }
                    tb = tile_bound[(tile + S < n_tiles) ? tile + S : tile];
                }
                if constexpr (DENSE) {
                    m_dense = m_nxt;
                    m_nxt = m_nxt2;
                    m_nxt2 = (tile + 2 * S < dense_tiles) ? dense_row[(int64_t)(tile + 2 * S) * 32] : 0u;
                }
                const f32x16 acc = score_tile_roll(a, (tile + 2 * S < tile_end) ? tile + 2 * S : tile);
                const unsigned mask = walk_mask(tile);
                float m_all = fmaxf(acc[0], acc[1]);
#pragma unroll
                for (int r = 2; r < 16; ++r) m_all = fmaxf(m_all, acc[r]);
                if (!(ablate & 2) && __any(m_all > tau)) {
                    float sc[16];
                    float m = m_all;
                    if (__any(mask != 0)) {
                        const unsigned m2 = mask >> (4 * hi);
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            sc[r] = (m2 & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : acc[r];
                        m = fmaxf(sc[0], sc[1]);
#pragma unroll
                        for (int r = 2; r < 16; ++r) m = fmaxf(m, sc[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[r] = acc[r];
                    }
                    if (__any(m > tau)) push_candidates(sc, tile * 32);
                }
                return true;
            };
            int step = 0;
            for (int tile = tile_begin; tile < tile_end;) {
                if (!tile_body(tile, step, aA)) break;
                tile += S;
                ++step;
                if (tile >= tile_end) break;
                if (!tile_body(tile, step, aB)) break;
                tile += S;
                ++step;
            }
        } else
#endif
        {
        float4 a_nxt[SHARED ? 1 : KQ];
        // SHARED: who still sweeps is counted per iteration in one of three LDS counters (waves that stop in iteration
        // `it` add to s_cnt[it % 3] before its barrier, everybody reads it behind the barrier, wave 0 clears the next one
        // while nobody can touch it): the decision to leave the loop is the same in every wave
        __shared__ int s_cnt[4];
        int dead = 0;
        if constexpr (SHARED) {
            if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
            __syncthreads();
            if (!alive && lane == 0) atomicAdd(&s_cnt[3], 1);
            stage_tile((tile_begin < n_tiles) ? tile_begin : 0, 0);
            stage_wait();
            dead = s_cnt[3];
        } else {
            load_frags((tile_begin < n_tiles) ? tile_begin : 0, a_nxt);
        }
        unsigned m_nxt = 0u;
        if constexpr (DENSE) m_nxt = (tile_begin < dense_tiles) ? dense_row[(int64_t)tile_begin * 32] : 0u;
        float tb = prune ? tile_bound[(tile_begin < n_tiles) ? tile_begin : 0] : 0.0f;
        PROF_ADD(4, prof_k0);
        int step = 0;
        for (int tile = tile_begin; tile < tile_end && dead < NW; tile += S, ++step) {
            if constexpr (SHARED) {
                if (threadIdx.x == 0) s_cnt[(step + 1) % 3] = 0;
                if (tile + S < tile_end) stage_tile(tile + S, (step + 1) & 1);
            }
            if ((!SHARED || alive) && prune) {
                // can any item from this tile on still enter a list of this wave?
                const bool open = en * tb > tau;
                const unsigned long long ob = __ballot(open);
                if (ob == 0ull) {
                    pruned = true;
                    exit_tile = tile;
                    if constexpr (SHARED) {
                        alive = false;      // keeps staging and meeting the barriers until the whole workgroup is done
                        if (lane == 0) atomicAdd(&s_cnt[step % 3], 1);
                    } else {
                        break;
                    }
                } else {
                    // tau is only refreshed by a flush; the last few users that keep the wave in the sweep
                    // get their pending ring entries merged so that their tau is exact (checked every 8 tiles)
                    if (((STRIDED ? step : tile) & 7) == 7 && __popcll(ob) <= 16) {
                        const unsigned long long pend = __ballot(cnt > 0);
                        flush_set((unsigned)(ob | (ob >> 32)) & (unsigned)(pend | (pend >> 32)));
                    }
                    tb = tile_bound[(tile + S < n_tiles) ? tile + S : tile];   // suffix maximum: covers my later tiles
                }
            }
#ifndef PK_SCORE_ROLL
            float4 a[SHARED ? 1 : KQ];
            if constexpr (!SHARED) {
#pragma unroll
                for (int q = 0; q < KQ; ++q) a[q] = a_nxt[q];
                load_frags((tile + S < tile_end) ? tile + S : tile, a_nxt);
            }
#endif
            if (!SHARED || alive) {
                if constexpr (DENSE) {
                    m_dense = m_nxt;
                    m_nxt = (tile + S < dense_tiles) ? dense_row[(int64_t)(tile + S) * 32] : 0u;
                }
#ifdef PK_SCORE_PROFILE2
                const unsigned long long prof_m0 = PROF_T();
#endif
#ifdef PK_SCORE_ROLL
                f32x16 acc;
                if constexpr (SHARED) acc = score_tile(a_nxt, step & 1);
                else acc = score_tile_roll(a_nxt, (tile + S < tile_end) ? tile + S : tile);
#else
                const f32x16 acc = score_tile(a, step & 1);
#endif
#ifdef PK_SCORE_PROFILE2
                asm volatile("" ::"v"(acc[0]), "v"(acc[15]));     // the products are done before the clock is read
                PROF_ADD(1, prof_m0);
#endif
                // The list cursor must advance every tile; the mask itself is only needed when some
                // RAW score beats the threshold (f32 MFMA shares the SIMD's FP32 lanes with the VALU, so
                // every VALU instruction here is paid in MFMA time: keep the common path to
                // 8 v_max3 + 1 compare and mask lazily).
                const unsigned long long prof_w0 = PROF_T();
                const unsigned mask = walk_mask(tile);
                PROF_ADD(2, prof_w0);
                float m_all = fmaxf(acc[0], acc[1]);
#pragma unroll
                for (int r = 2; r < 16; ++r) m_all = fmaxf(m_all, acc[r]);
                if (!(ablate & 2) && __any(m_all > tau)) {
                    const unsigned long long prof_p0 = PROF_T();
                    float sc[16];
                    float m = m_all;
                    if (__any(mask != 0)) {
                        const unsigned m2 = mask >> (4 * hi);
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            sc[r] = (m2 & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : acc[r];
                        m = fmaxf(sc[0], sc[1]);
#pragma unroll
                        for (int r = 2; r < 16; ++r) m = fmaxf(m, sc[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[r] = acc[r];
                    }
                    if (__any(m > tau)) push_candidates(sc, tile * 32);
                    PROF_ADD(3, prof_p0);
                }
            }
            if constexpr (SHARED) {
                stage_wait();
                dead += s_cnt[step % 3];
            }
        }
        }
    }
    if constexpr (SHARED) {
        // nothing to write for a wave without users or one whose group had left the sweep before this launch
        if (!participate || (!alive && !pruned)) return;
    }

    if (!last && !pruned) {
        // park the state for the next item chunk (rings stay unsorted: no flush cost per chunk)
        LaneState ls;
        ls.sp = sp;
        ls.tau = tau;
        ls.cnt = cnt;
        *my_state = ls;
        const int cmax = __builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, cnt));
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < cmax; ++i) my_ring_state[i * 64 + lane] = ring[i][lane];
        if (TOP_LDS)
            for (int s = lane; s < 32 * KC; s += 64) {
                const uint2 r = top[s];
                my_score[s] = __uint_as_float(r.x);
                my_idx[s] = (int)r.y;
            }
        PROF_INC(7, (tile_end - tile_begin + S - 1) / S);
        PROF_ADD(0, prof_k0);
        PROF_FLUSH();
        return;
    }
    // final merge of whatever is left in the rings
    {
        const unsigned long long some = __ballot(cnt > 0);
        flush_set((unsigned)(some | (some >> 32)));
    }
    // A sweep that started from a bootstrapped threshold must end with a FULL list (at least KC items beat that threshold,
    // and the sweep re-computes their scores with the same instructions).  Should it ever not — the bound "every
    // non-candidate scores at most the KC-th entry" would silently not hold — the last slot says so (PK_IDX_FLOOR) and
    // the re-scoring kernel sends the user to the exact path.
    const unsigned long long floored = (floor_state == nullptr) ? __ballot(tau_floor > -INFINITY) : 0ull;   // bit x: user x
    if (TOP_LDS) {
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < 32 * KC; s += 64) {
            const uint2 r = top[s];
            int iv = (int)r.y;
            if ((s % KC) == KC - 1 && iv < 0 && ((floored >> (s / KC)) & 1ull)) iv = PK_IDX_FLOOR;
            my_score[s] = __uint_as_float(r.x);
            my_idx[s] = iv;
        }
    } else if (floored) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (hi == 0 && tau_floor > -INFINITY && my_idx[ul * KC + KC - 1] < 0) my_idx[ul * KC + KC - 1] = PK_IDX_FLOOR;
    }
    PROF_INC(7, (exit_tile - tile_begin + S - 1) / S);
    PROF_ADD(0, prof_k0);
    PROF_FLUSH();
    {
        // finished (possibly early): later chunk launches must not touch this group again; sp records
        // the tile at which the group left the sweep (pk_score_state layout: see polara_hip.h)
        LaneState ls;
        ls.sp = exit_tile;
        ls.tau = tau;
        ls.cnt = PK_LANE_DONE;
        *my_state = ls;
    }
}

// ------------------------------------------------------------------------------------------
// score_candidates_pair_kernel (round 4): TWO user groups per wave.
// The dense regimes of the sweep are bound by the fragment loads, not by the matrix cores: a PK_SCORE_DIAG build of the
// full ML-20M-shaped sweep (no pushes) takes 1.98 ms as it is, 1.18 ms without the fragment re-loads, 1.79 ms without the
// PRODUCTS (loads still waited for) and 0.83 ms without either (tools/probes/sweep_floor.py, PK_FLOOR_DIAG) — 3.6 M
// tile-waves x 8 KB = 30 GB through the L1 path at ~17 TB/s, the same ceiling the SpMM gathers hit (64 lanes x 16 B per
// load instruction).  More waves do not help (four forced waves: slower), staging tiles in LDS costs a barrier per tile
// (round 3); what halves the bytes per product is using every loaded fragment for TWO groups of 32 users: this kernel.
// A wave owns groups 2 w and 2 w + 1 (adjacent in activity order: they leave the sweep at similar tiles); per k-step six
// MFMAs on two independent accumulator chains, then the step's fragment registers are re-requested for the next tile
// (the rolling buffer of score_tile_roll); selection state, rings and lists exist once per group (GS), the epilogue of a
// tile runs once per group that is still sweeping.  Single sweeps with the lists in LDS and KC = 16 (top-10 lists: two
// groups' rings and lists are 16 KB per wave, 64 KB per workgroup; KC = 32 would leave one wave per SIMD), same parked
// state and list layout as score_candidates_kernel, same scores bit for bit (the MFMA sequence of a group is unchanged).
// MEASURED (bench.py, ML-20M-shaped, one group -> two groups per wave; 240 VGPRs, two waves per SIMD): pruned headline
// sweep 0.377 -> 0.617 ms, flat-norm catalogue 63.5 -> 50.2 M users/s, no-prune 74 -> 71 M, pop^0.25 79 -> 71 M: SLOWER in
// every regime — half the fragment bytes per product buy nothing, so the "loads" of the diagnostic build are waiting
// time that three free-running waves overlap better than two fat ones, not bytes.  OPT-IN (PK_SCORE_PAIR=1), kept as the
// record of the experiment like the LDS-staged instance; lists identical (tests/test_gpu_kernels.py).
// ------------------------------------------------------------------------------------------
template <int NSTEP, int KC, bool DENSE>
__global__ __launch_bounds__(256) void score_candidates_pair_kernel(
    const float4 *__restrict__ Vp, const float4 *__restrict__ Ep, int64_t n_users, int n_items,
    int n_tiles, int split_tiles, int chunk_begin, int chunk_tiles,
    const int64_t *__restrict__ seen_ptr, const unsigned long long *__restrict__ seen_tiles,
    const int32_t *__restrict__ seen_ntiles,
    float *__restrict__ cand_score, int32_t *__restrict__ cand_idx,
    LaneState *__restrict__ st_lane, uint2 *__restrict__ st_ring,
    const float *__restrict__ user_bound, const float *__restrict__ tile_bound, int ablate, SeenDense dense,
    int slot_base, int boot_tiles) {
    constexpr int KQ = 2 * NSTEP;
    static_assert(KC == 16 && pk_top_in_lds(NSTEP, KC), "pair kernel: top-10 lists (KC = 16) in LDS");
    constexpr int RG = pk_ring_rows(KC);
    constexpr int NG = 2, NW = 4;
    extern __shared__ __attribute__((aligned(16))) uint2 pk_score_lds[];      // [NW * NG][RG][64] rings, then [NW * NG][32 * KC] lists
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_groups = (n_users + 31) / 32;
    const int tile_begin = chunk_begin;
    const int tile_stop = (chunk_begin + chunk_tiles < split_tiles) ? chunk_begin + chunk_tiles : n_tiles;
    const int tile_end = (tile_stop < n_tiles) ? tile_stop : n_tiles;
    const bool first = (chunk_begin == 0);
    const bool last = (tile_stop >= n_tiles);
    if (!first && tile_begin >= n_tiles) return;
    const int ul = lane & 31, hi = lane >> 5;
    const bool prune = (user_bound != nullptr && tile_bound != nullptr) && !(ablate & 4);
    const int dense_tiles = (DENSE && seen_ptr != nullptr) ? dense.tiles : 0;

    struct GS {
        float4 e[KQ];
        int64_t sp, se;
        unsigned long long nxt, nxt2, nxt3;
        float tau, tau_floor, en;
        int cnt, exit_tile;
        bool live, alive, pruned, has_seen;      // live: takes part in this launch; alive: still sweeping
        float *my_score;
        int32_t *my_idx;
        LaneState *my_state;
        uint2 *my_ring_state;
        const unsigned *dense_row;
        unsigned m_dense, m_nxt;
        uint2 (*ring)[64];
        uint2 *top;
    };
    GS gs[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        GS &s = gs[g];
        const int64_t group_raw = ((int64_t)blockIdx.x * NW + wave) * NG + g;
        s.live = group_raw * 32 < n_users;
        const int64_t group = s.live ? group_raw : 0;
        const int64_t user = group * 32 + ul;
        const int64_t slot = (int64_t)slot_base * n_groups + group;
        s.my_score = cand_score + slot * 32 * KC;
        s.my_idx = cand_idx + slot * 32 * KC;
        s.my_state = st_lane + slot * 64 + lane;
        s.my_ring_state = st_ring + slot * (RING * 64);
        s.ring = reinterpret_cast<uint2(*)[64]>(pk_score_lds + (wave * NG + g) * (RG * 64));
        s.top = pk_score_lds + NW * NG * RG * 64 + (wave * NG + g) * (32 * KC);
#pragma unroll
        for (int q = 0; q < KQ; ++q) s.e[q] = Ep[(group * KQ + q) * 64 + lane];
        s.sp = s.se = 0;
        s.nxt = s.nxt2 = s.nxt3 = PK_TILE_NONE;
        s.tau = -INFINITY;
        s.tau_floor = -INFINITY;
        s.cnt = 0;
        s.en = (prune && s.live && user < n_users) ? user_bound[user] : -1.0f;
        s.pruned = false;
        s.exit_tile = tile_end;
        s.has_seen = (seen_ptr != nullptr && s.live && user < n_users);
        if (s.has_seen) {
            s.sp = seen_ptr[user];
            s.se = s.sp + seen_ntiles[user];
        }
        s.dense_row = dense_tiles ? dense.mask + ((int64_t)group * dense_tiles) * 32 + ul : nullptr;
        s.m_dense = s.m_nxt = 0u;
        if (DENSE && first && s.has_seen && dense_tiles) s.sp += dense.skip[user];
        s.alive = s.live;
        if (s.live) {
            if (first) {
                for (int t = lane; t < 32 * KC; t += 64) s.top[t] = make_uint2(__float_as_uint(-INFINITY), 0xffffffffu);
            } else {
                const LaneState ls = *s.my_state;
                if (ls.cnt == PK_LANE_DONE) {              // wave-uniform: this group was pruned in an earlier launch
                    s.live = s.alive = false;
                } else {
                    for (int t = lane; t < 32 * KC; t += 64) s.top[t] = make_uint2(__float_as_uint(s.my_score[t]), (unsigned)s.my_idx[t]);
                    if (s.has_seen) s.sp = ls.sp;
                    s.tau = ls.tau;
                    s.tau_floor = fmaxf(s.tau_floor, s.tau);
                    s.cnt = ls.cnt;
                    const int cmax = __builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, s.cnt));
                    for (int i = 0; i < cmax; ++i) s.ring[i][lane] = s.my_ring_state[i * 64 + lane];
                }
            }
        }
        if (s.has_seen && s.live) {
            if (s.sp < s.se) s.nxt = seen_tiles[s.sp];
            if (s.sp + 1 < s.se) s.nxt2 = seen_tiles[s.sp + 1];
            if (s.sp + 2 < s.se) s.nxt3 = seen_tiles[s.sp + 2];
        }
    }
    if (!gs[0].live && !gs[1].live) return;

    // users x (lanes 0..31) and y (lanes 32..63; y < 0: nobody) of group s merged in one 32-wide sort each (flush_pair of
    // score_candidates_kernel, state through s)
    auto flush_pair = [&](GS &s, int x, int y) {
        __builtin_amdgcn_wave_barrier();
        const int t = lane & 31;
        const int yy = (y >= 0) ? y : x;
        const int cx_lo = __builtin_amdgcn_readlane(s.cnt, x), cx_hi = __builtin_amdgcn_readlane(s.cnt, x + 32);
        const int cy_lo = __builtin_amdgcn_readlane(s.cnt, yy), cy_hi = __builtin_amdgcn_readlane(s.cnt, yy + 32);
        const int u = hi ? yy : x;
        const int c_lo = hi ? cy_lo : cx_lo, c_hi = hi ? cy_hi : cx_hi;
        const bool act = !hi || y >= 0;
        float k = -INFINITY;
        int v = PK_IDX_NONE;
        if (act) {
            if (t < KC) {
                const uint2 r = s.top[u * KC + t];
                if ((int)r.y >= 0) {
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (t < KC + RG) {
                if (t - KC < c_lo) {
                    const uint2 r = s.ring[t - KC][u];
                    k = __uint_as_float(r.x);
                    v = (int)r.y;
                }
            } else if (t - KC - RG < c_hi) {
                const uint2 r = s.ring[t - KC - RG][u + 32];
                k = __uint_as_float(r.x);
                v = (int)r.y;
            }
        }
        unsigned key = (pk_float_order(k) & ~31u) | (unsigned)t;
        pk_sort32_half_levels<32>(key);
        const int src = (lane & 32) | (int)(key & 31u);
        k = __shfl(k, src, 64);
        v = __shfl(v, src, 64);
        if (act && t < KC) s.top[u * KC + t] = make_uint2(__float_as_uint(k), (unsigned)((v == PK_IDX_NONE) ? -1 : v));
        const float tx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), KC - 1));
        const float ty = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k), 32 + KC - 1));
        if (ul == x) {
            s.tau = fmaxf(tx, s.tau_floor);
            s.cnt = 0;
        }
        if (y >= 0 && ul == y) {
            s.tau = fmaxf(ty, s.tau_floor);
            s.cnt = 0;
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_set = [&](GS &s, unsigned um) {
        while (um) {
            const int x = __builtin_ctz(um);
            um &= um - 1;
            int y = -1;
            if (um) {
                y = __builtin_ctz(um);
                um &= um - 1;
            }
            flush_pair(s, x, y);
        }
    };
    auto walk_mask = [&](GS &s, int tile) -> unsigned {
        const int j0 = tile * 32, jend = j0 + 32;
        unsigned mask = 0;
        if (ablate & 1) return 0u;
        if constexpr (DENSE) {
            if (tile < dense_tiles) {
                mask = s.m_dense;
                if (jend > n_items) mask |= ~0u << (n_items - j0);
                return mask;
            }
        }
        const bool hit = (unsigned)(s.nxt >> 32) == (unsigned)tile;
        if (__any(hit)) {
            if (hit) {
                mask = (unsigned)s.nxt;
                ++s.sp;
                s.nxt = s.nxt2;
                s.nxt2 = s.nxt3;
                s.nxt3 = (s.sp + 2 < s.se) ? seen_tiles[s.sp + 2] : PK_TILE_NONE;
            }
        }
        if (jend > n_items) mask |= ~0u << (n_items - j0);
        return mask;
    };
    auto push_candidates = [&](GS &s, const float(&acc)[16], int j0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bool c = acc[r] > s.tau;
            if (__any(c)) {
                if (__any(c && s.cnt == RG)) {
                    const unsigned long long full = __ballot(s.cnt == RG);
                    unsigned um = (unsigned)(full | (full >> 32));
                    if (__builtin_popcount(um) & 1) {
                        const unsigned long long part = __ballot(2 * s.cnt >= RG);
                        const unsigned cand = (unsigned)(part | (part >> 32)) & ~um;
                        if (cand) um |= 1u << __builtin_ctz(cand);
                    }
                    flush_set(s, um);
                    c = acc[r] > s.tau;
                }
                if (c) {
                    s.ring[s.cnt][lane] = make_uint2(__float_as_uint(acc[r]), (unsigned)(j0 + (r & 3) + 8 * (r >> 2) + 4 * hi));
                    ++s.cnt;
                }
            }
        }
    };
    // both groups' score tiles of one step: per k-step three MFMAs per group (two independent accumulator chains), then
    // the step's fragment registers are re-requested for the next tile
    float4 a[KQ];
    auto score_pair = [&](int next_tile, f32x16 &acc0, f32x16 &acc1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.0f;
            acc1[r] = 0.0f;
        }
        const float4 *vp = Vp + ((int64_t)next_tile * KQ) * 64 + lane;
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; ++sidx) {
            const bf16x8 vh = __builtin_bit_cast(bf16x8, a[2 * sidx]), vl = __builtin_bit_cast(bf16x8, a[2 * sidx + 1]);
            const bf16x8 eh0 = __builtin_bit_cast(bf16x8, gs[0].e[2 * sidx]), el0 = __builtin_bit_cast(bf16x8, gs[0].e[2 * sidx + 1]);
            const bf16x8 eh1 = __builtin_bit_cast(bf16x8, gs[1].e[2 * sidx]), el1 = __builtin_bit_cast(bf16x8, gs[1].e[2 * sidx + 1]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, eh1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, el1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, eh1, acc1, 0, 0, 0);
            a[2 * sidx] = vp[(2 * sidx) * 64];
            a[2 * sidx + 1] = vp[(2 * sidx + 1) * 64];
        }
    };
    auto load_frags = [&](int tile) {
        const float4 *vp = Vp + ((int64_t)tile * KQ) * 64 + lane;
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = vp[q * 64];
    };

    // ---- threshold bootstrap (see score_candidates_kernel): the first boot_tiles tiles scored once without selecting ----
    if (first && boot_tiles > 0 && !(ablate & 8)) {
        constexpr int BL = KC / 2, BG = KC / 8;
        float bl[NG][BL];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < BL; ++i) bl[g][i] = -INFINITY;
        int64_t sp0[NG];
        unsigned long long n0[NG], n1[NG], n2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            sp0[g] = gs[g].sp;
            n0[g] = gs[g].nxt;
            n1[g] = gs[g].nxt2;
            n2[g] = gs[g].nxt3;
            if constexpr (DENSE) gs[g].m_nxt = (gs[g].live && tile_begin < dense_tiles) ? gs[g].dense_row[(int64_t)tile_begin * 32] : 0u;
        }
        load_frags((tile_begin < n_tiles) ? tile_begin : 0);
        for (int i = 0, tile = tile_begin; i < boot_tiles && tile < tile_end; ++i, ++tile) {
            if constexpr (DENSE) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    gs[g].m_dense = gs[g].m_nxt;
                    gs[g].m_nxt = (gs[g].live && tile + 1 < dense_tiles) ? gs[g].dense_row[(int64_t)(tile + 1) * 32] : 0u;
                }
            }
            f32x16 acc[NG];
            score_pair((tile + 1 < tile_end) ? tile + 1 : tile, acc[0], acc[1]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                GS &s = gs[g];
                if (!s.live) continue;
                const unsigned m2 = walk_mask(s, tile) >> (4 * hi);
#pragma unroll
                for (int gg = 0; gg < BG; ++gg) {
                    float x = -INFINITY;
#pragma unroll
                    for (int r = gg * (16 / BG); r < (gg + 1) * (16 / BG); ++r)
                        x = fmaxf(x, (m2 & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : acc[g][r]);
#pragma unroll
                    for (int i2 = 0; i2 < BL; ++i2) {
                        const float up = fmaxf(bl[g][i2], x);
                        x = fminf(bl[g][i2], x);
                        bl[g][i2] = up;
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            GS &s = gs[g];
            float t0 = bl[g][KC / 2 - 1];
            t0 = fminf(t0, __int_as_float(pk_lane_xor<32>(__float_as_int(t0))));
            if (t0 > -INFINITY) {
                t0 = fminf(t0 - fabsf(t0) * 2.4e-7f, t0 - 1e-37f);
                s.tau_floor = fmaxf(s.tau_floor, t0);
                s.tau = fmaxf(s.tau, s.tau_floor);
            }
            s.sp = sp0[g];
            s.nxt = n0[g];
            s.nxt2 = n1[g];
            s.nxt3 = n2[g];
        }
    }
    // ---- the sweep ------------------------------------------------------------------------------------------------------------
    load_frags((tile_begin < n_tiles) ? tile_begin : 0);
    if constexpr (DENSE) {
#pragma unroll
        for (int g = 0; g < NG; ++g) gs[g].m_nxt = (gs[g].live && tile_begin < dense_tiles) ? gs[g].dense_row[(int64_t)tile_begin * 32] : 0u;
    }
    float tb = prune ? tile_bound[(tile_begin < n_tiles) ? tile_begin : 0] : 0.0f;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        if (prune) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                GS &s = gs[g];
                if (!s.alive) continue;
                const bool open = s.en * tb > s.tau;
                const unsigned long long ob = __ballot(open);
                if (ob == 0ull) {
                    s.pruned = true;
                    s.exit_tile = tile;
                    s.alive = false;
                } else if ((tile & 7) == 7 && __popcll(ob) <= 16) {
                    const unsigned long long pend = __ballot(s.cnt > 0);
                    flush_set(s, (unsigned)(ob | (ob >> 32)) & (unsigned)(pend | (pend >> 32)));
                }
            }
            if (!gs[0].alive && !gs[1].alive) break;
            tb = tile_bound[(tile + 1 < n_tiles) ? tile + 1 : tile];
        }
        if constexpr (DENSE) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                gs[g].m_dense = gs[g].m_nxt;
                gs[g].m_nxt = (gs[g].live && tile + 1 < dense_tiles) ? gs[g].dense_row[(int64_t)(tile + 1) * 32] : 0u;
            }
        }
        f32x16 acc[NG];
        score_pair((tile + 1 < tile_end) ? tile + 1 : tile, acc[0], acc[1]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            GS &s = gs[g];
            if (!s.alive) continue;
            const unsigned mask = walk_mask(s, tile);
            float m_all = fmaxf(acc[g][0], acc[g][1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) m_all = fmaxf(m_all, acc[g][r]);
            if (!(ablate & 2) && __any(m_all > s.tau)) {
                float sc[16];
                float m = m_all;
                if (__any(mask != 0)) {
                    const unsigned m2 = mask >> (4 * hi);
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = (m2 & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : acc[g][r];
                    m = fmaxf(sc[0], sc[1]);
#pragma unroll
                    for (int r = 2; r < 16; ++r) m = fmaxf(m, sc[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = acc[g][r];
                }
                if (__any(m > s.tau)) push_candidates(s, sc, tile * 32);
            }
        }
    }
    // ---- end of the launch: park, or merge what is left and write the lists ----------------------------------------------------
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        GS &s = gs[g];
        if (!s.live) continue;
        if (!last && !s.pruned) {
            LaneState ls;
            ls.sp = s.sp;
            ls.tau = s.tau;
            ls.cnt = s.cnt;
            *s.my_state = ls;
            const int cmax = __builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, s.cnt));
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < cmax; ++i) s.my_ring_state[i * 64 + lane] = s.ring[i][lane];
            for (int t = lane; t < 32 * KC; t += 64) {
                const uint2 r = s.top[t];
                s.my_score[t] = __uint_as_float(r.x);
                s.my_idx[t] = (int)r.y;
            }
            continue;
        }
        {
            const unsigned long long some = __ballot(s.cnt > 0);
            flush_set(s, (unsigned)(some | (some >> 32)));
        }
        const unsigned long long floored = __ballot(s.tau_floor > -INFINITY);
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < 32 * KC; t += 64) {
            const uint2 r = s.top[t];
            int iv = (int)r.y;
            if ((t % KC) == KC - 1 && iv < 0 && ((floored >> (t / KC)) & 1ull)) iv = PK_IDX_FLOOR;
            s.my_score[t] = __uint_as_float(r.x);
            s.my_idx[t] = iv;
        }
        LaneState ls;
        ls.sp = s.exit_tile;
        ls.tau = s.tau;
        ls.cnt = PK_LANE_DONE;
        *s.my_state = ls;
    }
}

// ------------------------------------------------------------------------------------------
// packing: f64 [n x K] -> f32 MFMA fragments, 32 rows per tile, K padded with zeros to 8*KQ
// ------------------------------------------------------------------------------------------
// NSTEP = number of 16-wide k-steps of the split-bf16 product (rank padded with zeros to 16 * NSTEP)
static const int kNstepSet[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 13, 16};

static int pk_nstep(int K) {
    const int need = (K + 15) / 16;
    for (int n : kNstepSet)
        if (n >= need) return n;
    return 0;
}

extern "C" int32_t pk_pack_kq(int32_t K) {    // 16-byte groups per lane and 32-row tile
    const int n = pk_nstep(K);
    return n ? 2 * n : 0;
}

extern "C" int64_t pk_pack_elems(int64_t n, int32_t K) {
    const int kq = pk_pack_kq(K);
    return pk_ceil_div(n, 32) * kq * 64 * 4;
}

// split of an fp32 value into two bf16 (round to nearest even): x = hi + lo + d, |hi - x| <= 2^-8 |x|, |d| <= 2^-16 |x|
__device__ __forceinline__ unsigned pk_bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void pk_split_bf16(float x, unsigned &hi, unsigned &lo) {
    hi = pk_bf16_rne(x);
    lo = pk_bf16_rne(x - __uint_as_float(hi << 16));
}
// lane (i = lane & 31, h = lane >> 5) of k-step s holds k = 16 s + 8 h + j, j = 0..7: the A / B operand layout of
// v_mfma_f32_32x32x16_bf16.  Group 2 s = the eight hi parts, group 2 s + 1 = the eight lo parts.
__device__ __forceinline__ void pk_pack_step(const double *__restrict__ r, bool live, int K, int s, int h, uint4 &ghi, uint4 &glo,
                                             double *sumsq) {
    unsigned hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * h + j;
        const double x = (live && k < K) ? r[k] : 0.0;
        if (sumsq) *sumsq = fma(x, x, *sumsq);
        pk_split_bf16((float)x, hi[j], lo[j]);
    }
    ghi = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    glo = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

__global__ __launch_bounds__(256) void pack_frag_kernel(int64_t n, int K, int kq, const double *__restrict__ src,
                                                        int64_t ld, uint4 *__restrict__ dst, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (k-step, lane) per thread
    if (t >= total) return;
    const int ns = kq / 2;
    const int lane = (int)(t & 63);
    const int64_t ts = t >> 6;
    const int s = (int)(ts % ns);
    const int64_t tile = ts / ns;
    const int64_t row = tile * 32 + (lane & 31);
    uint4 ghi, glo;
    pk_pack_step(src + (row < n ? row : 0) * ld, row < n, K, s, lane >> 5, ghi, glo, nullptr);
    dst[(tile * kq + 2 * s) * 64 + lane] = ghi;
    dst[(tile * kq + 2 * s + 1) * 64 + lane] = glo;
}

// The same packing of a [n x K] fp64 block with the pruning bound of every row computed on the way (the user
// side of a scoring pass: E is read once instead of twice).  One wave per 32-row tile: lane (i = lane & 31,
// h = lane >> 5) produces its groups of every k-step and accumulates the squares of what it read; the two
// halves of a row meet through one lane exchange.  bound[r] = ||src[r,:]|| (1 + 1e-6) + extra_scale * extra[r]
// (extra: the error weight of an approximate fold-in, or NULL).
__global__ __launch_bounds__(256) void pack_frag_bound_kernel(int64_t n, int K, int kq, const double *__restrict__ src,
                                                              int64_t ld, uint4 *__restrict__ dst,
                                                              float *__restrict__ bound,
                                                              const double *__restrict__ extra, int64_t extra_ld,
                                                              double extra_scale, int64_t n_tiles) {
    __shared__ __attribute__((aligned(16))) double s_tile[4][32 * 16];   // one k-step of a wave's 32 rows
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const int64_t row = tile * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool live = row < n;
    const double *r = src + (live ? row : 0) * ld;
    double ss = 0.0;
    // STAGED (16-byte aligned rows): a lane reading its own row touches 64 different lines per load instruction and the
    // kernel crawls at 1.7 TB/s (0.25 ms per 1M users x rank 50).  Here eight lanes fetch the 128 bytes a row
    // contributes to a k-step with one 16-byte load each — a load instruction covers 8 rows completely, four cover the
    // tile — the k-step goes through LDS (4 KB per wave) and every lane picks up its 64 bytes from there.
    const bool staged = ((ld & 1) == 0) && ((((uintptr_t)src) & 15) == 0);
    for (int s = 0; s < kq / 2; ++s) {
        uint4 ghi, glo;
        if (staged) {
            double *buf = s_tile[wave];
            __builtin_amdgcn_wave_barrier();             // the previous k-step's reads are done (in-order LDS pipe)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rr = 8 * p + (lane >> 3);      // row of the tile this lane fetches from in pass p
                const int kk = 16 * s + 2 * (lane & 7);  // first of its two columns
                const int64_t grow = tile * 32 + rr;
                double2 v = make_double2(0.0, 0.0);
                if (grow < n && kk < K) {
                    const double *g = src + grow * ld + kk;
                    if (kk + 1 < K) v = *reinterpret_cast<const double2 *>(g);
                    else v.x = g[0];
                }
                *reinterpret_cast<double2 *>(buf + rr * 16 + 2 * (lane & 7)) = v;
            }
            __builtin_amdgcn_wave_barrier();
            unsigned hi[8], lo[8];
            const double *mine = buf + (lane & 31) * 16 + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double x = mine[j];
                ss = fma(x, x, ss);
                pk_split_bf16((float)x, hi[j], lo[j]);
            }
            ghi = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
            glo = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
        } else {
            pk_pack_step(r, live, K, s, h, ghi, glo, &ss);
        }
        dst[(tile * kq + 2 * s) * 64 + lane] = ghi;
        dst[(tile * kq + 2 * s + 1) * 64 + lane] = glo;
    }
    ss += pk_lane_xor<32>(ss);
    if (live && h == 0) {
        double b = sqrt(ss) * (1.0 + 1e-6);
        if (extra) b += extra_scale * extra[row * extra_ld];
        bound[row] = (float)(b * (1.0 + 1e-7));   // the conversion rounds to nearest: keep it an upper bound
    }
}

extern "C" int pk_pack_frag_bound_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld,
                                      float *dst_dev, float *bound_dev, const double *extra_dev, int64_t extra_ld,
                                      double extra_scale) {
    const int kq = pk_pack_kq(K);
    PK_REQUIRE(n >= 1 && K >= 1 && kq > 0, "pk_pack_frag_bound_f32: n=%lld K=%d unsupported (K <= 256)", (long long)n, K);
    PK_REQUIRE(ld >= K && ((uintptr_t)dst_dev % 16) == 0 && bound_dev, "pk_pack_frag_bound_f32: bad ld / alignment / pointers");
    const int64_t n_tiles = pk_ceil_div(n, 32);
    hipLaunchKernelGGL(pack_frag_bound_kernel, dim3((unsigned)pk_ceil_div(n_tiles, 4)), dim3(256), 0, pk_stream(stream), n,
                       K, kq, src_dev, ld, reinterpret_cast<uint4 *>(dst_dev), bound_dev, extra_dev, extra_ld,
                       extra_scale, n_tiles);
    PK_CHECK_LAUNCH("pack_frag_bound_kernel");
    return PK_OK;
}

extern "C" int pk_pack_frag_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld,
                                float *dst_dev) {
    const int kq = pk_pack_kq(K);
    PK_REQUIRE(n >= 1 && K >= 1 && kq > 0, "pk_pack_frag_f32: n=%lld K=%d unsupported (K <= 256)", (long long)n, K);
    PK_REQUIRE(ld >= K && ((uintptr_t)dst_dev % 16) == 0, "pk_pack_frag_f32: bad ld / alignment");
    const int64_t total = pk_ceil_div(n, 32) * (kq / 2) * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)pk_ceil_div(total, 256)), dim3(256), 0, pk_stream(stream), n,
                       K, kq, src_dev, ld, reinterpret_cast<uint4 *>(dst_dev), total);
    PK_CHECK_LAUNCH("pack_frag_kernel");
    return PK_OK;
}

// ---- seen-tile stream ----------------------------------------------------------------------------
// One wave per user: the seen-item list [seen_ptr[u], seen_ptr[u+1]) is folded into one 64-bit record
// (tile << 32 | 32-bit item mask) per tile that holds seen items, written compactly from position
// seen_ptr[u] of `tiles` (same indptr as the item list, at most as many records).
// SORTED = false: the list is in arbitrary order (a CSR whose columns were only RENAMED, e.g. to the
// factor-norm order of the serving index); the wave first sorts it in LDS (bitonic, <= PK_SEEN_CAP entries).
#define PK_SEEN_CAP 4096
template <bool SORTED>
__global__ __launch_bounds__(256) void seen_tiles_kernel(int64_t n_users, const int64_t *__restrict__ seen_ptr,
                                                         const int32_t *__restrict__ seen_idx,
                                                         unsigned long long *__restrict__ tiles,
                                                         int32_t *__restrict__ ntiles) {
    __shared__ int sort_buf[SORTED ? 1 : 4][SORTED ? 1 : PK_SEEN_CAP];
    const int lane = threadIdx.x & 63;
    const int64_t user = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (user >= n_users) return;
    const int64_t p0 = seen_ptr[user], p1 = seen_ptr[user + 1];
    int *buf = sort_buf[SORTED ? 0 : (threadIdx.x >> 6)];
    if constexpr (!SORTED) {
        const int n = (int)(p1 - p0);
        int N = 64;
        while (N < n) N <<= 1;
        for (int i = lane; i < N; i += 64) buf[i] = (i < n) ? seen_idx[p0 + i] : 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        // ascending bitonic network in LDS; one wave, so program order + the in-order LDS pipe are the only
        // synchronisation needed (the barriers keep the compiler from moving accesses across stages)
        for (int k = 2; k <= N; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (N >> 1); t += 64) {
                    const int pos = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int x = buf[pos], y = buf[pos + j];
                    const bool up = (pos & k) == 0;
                    if ((x > y) == up) {
                        buf[pos] = y;
                        buf[pos + j] = x;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    int count = 0;
    bool carry = false;             // the last tile of the previous chunk is emitted with the next chunk
    unsigned c_tile = 0, c_mask = 0;
    for (int64_t base = p0; base < p1; base += 64) {
        const int64_t i = base + lane;
        const bool valid = i < p1;
        int idx = 0;
        if (valid) idx = SORTED ? seen_idx[i] : buf[i - p0];
        const unsigned tile = valid ? (unsigned)idx >> 5 : 0xffffffffu;
        unsigned m = valid ? 1u << (idx & 31) : 0u;
        // OR of the bits of my tile over the lanes to my right (sorted list: equal tiles are adjacent)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t2 = __shfl_down(tile, off, 64);
            const unsigned m2 = __shfl_down(m, off, 64);
            if (lane + off < 64 && t2 == tile) m |= m2;
        }
        const unsigned tprev = __shfl_up(tile, 1, 64);
        const bool head = valid && (lane == 0 || tprev != tile);
        const unsigned long long heads = __ballot(head);
        const int last_head = 63 - __builtin_clzll(heads);          // heads != 0: lane 0 is valid
        const unsigned f_tile = __shfl(tile, 0, 64);
        if (carry) {
            if (f_tile == c_tile) {
                if (lane == 0) m |= c_mask;                          // same tile continues in this chunk
            } else {
                if (lane == 0) tiles[p0 + count] = ((unsigned long long)c_tile << 32) | c_mask;
                ++count;
            }
        }
        const int rank = __popcll(heads & ((1ull << lane) - 1ull));
        if (head && lane != last_head) tiles[p0 + count + rank] = ((unsigned long long)tile << 32) | m;
        count += __popcll(heads) - 1;
        c_tile = __shfl(tile, last_head, 64);
        c_mask = __shfl(m, last_head, 64);
        carry = true;
    }
    if (carry) {
        if (lane == 0) tiles[p0 + count] = ((unsigned long long)c_tile << 32) | c_mask;
        ++count;
    }
    if (lane == 0) ntiles[user] = count;
}

extern "C" int32_t pk_seen_tiles_max_unsorted_row(void) { return PK_SEEN_CAP; }

// Dense masks of the first `dense_tiles` tiles from the seen-tile stream: one wave per user copies the masks of its
// records below dense_tiles into the [group][tile][32 users] array (every (user, tile) has at most one record: plain
// stores) and counts them.  The array is zeroed here; built once per test matrix like the stream itself.
__global__ __launch_bounds__(256) void seen_dense_kernel(int64_t n_users, const int64_t *__restrict__ seen_ptr,
                                                         const unsigned long long *__restrict__ tiles,
                                                         const int32_t *__restrict__ ntiles, int dense_tiles,
                                                         unsigned *__restrict__ dense, int32_t *__restrict__ skip) {
    const int lane = threadIdx.x & 63;
    const int64_t user = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (user >= n_users) return;
    const int64_t p0 = seen_ptr[user];
    const int n = ntiles[user];
    unsigned *row = dense + ((user >> 5) * dense_tiles) * 32 + (user & 31);
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
        const unsigned long long rec = tiles[p0 + i];
        const unsigned t = (unsigned)(rec >> 32);
        if (t < (unsigned)dense_tiles) {
            row[(int64_t)t * 32] = (unsigned)rec;
            ++cnt;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) skip[user] = cnt;
}

extern "C" int64_t pk_seen_dense_bytes(int64_t n_users, int32_t dense_tiles) {
    return pk_ceil_div(n_users, 32) * (int64_t)dense_tiles * 32 * 4;
}

extern "C" int pk_seen_dense_build(void *stream, int64_t n_users, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                                   const int32_t *seen_ntiles_dev, int32_t dense_tiles, uint32_t *dense_dev, int32_t *skip_dev) {
    PK_REQUIRE(n_users >= 1 && dense_tiles >= 1 && seen_ptr_dev && seen_ntiles_dev && dense_dev && skip_dev,
               "pk_seen_dense_build: bad arguments");
    hipStream_t st = pk_stream(stream);
    const hipError_t e = hipMemsetAsync(dense_dev, 0, (size_t)pk_seen_dense_bytes(n_users, dense_tiles), st);
    if (e != hipSuccess) {
        pk_set_error("pk_seen_dense_build: memset: %s", hipGetErrorString(e));
        return PK_E_LAUNCH;
    }
    hipLaunchKernelGGL(seen_dense_kernel, dim3((unsigned)pk_ceil_div(n_users, 4)), dim3(256), 0, st, n_users, seen_ptr_dev,
                       reinterpret_cast<const unsigned long long *>(seen_tiles_dev), seen_ntiles_dev, dense_tiles, dense_dev,
                       skip_dev);
    PK_CHECK_LAUNCH("seen_dense_kernel");
    return PK_OK;
}

extern "C" int pk_seen_tiles_build(void *stream, int64_t n_users, const int64_t *seen_ptr_dev,
                                   const int32_t *seen_idx_dev, int32_t rows_sorted, int64_t max_row_len,
                                   uint64_t *tiles_dev, int32_t *ntiles_dev) {
    // seen_idx_dev / tiles_dev may be NULL when the matrix has no entry at all (never dereferenced then)
    PK_REQUIRE(n_users >= 1 && seen_ptr_dev && ntiles_dev, "pk_seen_tiles_build: bad arguments");
    dim3 grid((unsigned)pk_ceil_div(n_users, 4)), block(256);
    unsigned long long *tiles = reinterpret_cast<unsigned long long *>(tiles_dev);
    if (rows_sorted) {
        hipLaunchKernelGGL(seen_tiles_kernel<true>, grid, block, 0, pk_stream(stream), n_users, seen_ptr_dev,
                           seen_idx_dev, tiles, ntiles_dev);
    } else {
        PK_REQUIRE(max_row_len >= 0 && max_row_len <= PK_SEEN_CAP,
                   "pk_seen_tiles_build: unsorted rows longer than %d entries (got %lld): sort the rows first",
                   PK_SEEN_CAP, (long long)max_row_len);
        hipLaunchKernelGGL(seen_tiles_kernel<false>, grid, block, 0, pk_stream(stream), n_users, seen_ptr_dev,
                           seen_idx_dev, tiles, ntiles_dev);
    }
    PK_CHECK_LAUNCH("seen_tiles_kernel");
    return PK_OK;
}

// ---- pruning bounds ------------------------------------------------------------------------------
// out[r] >= ||src[r, :]||_2 as a float (relative head-room 1e-6 covers the double->float rounding and
// the rounding of the fp32 product  user_bound * tile_bound  inside the sweep).
__global__ __launch_bounds__(256) void row_norm_bound_kernel(int64_t n, int K, const double *__restrict__ src,
                                                             int64_t ld, float *__restrict__ out) {
    // 16 lanes per row: coalesced 128-byte pieces of the row, 4-step butterfly inside the 16-lane group
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    double s = 0.0;
    if (r < n) {
        const double *row = src + r * ld;
        for (int k = l; k < K; k += 16) s = fma(row[k], row[k], s);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (r < n && l == 0) out[r] = (float)(sqrt(s) * (1.0 + 1e-6));
}

// out[t] = max(norm[32 t ...]) over ALL rows from tile t to the end (suffix maximum); one wave.
__global__ __launch_bounds__(64) void tile_suffix_max_kernel(int64_t n, int64_t n_tiles, const float *__restrict__ norm,
                                                             float *__restrict__ out) {
    const int lane = threadIdx.x;
    const int64_t seg = (n_tiles + 63) / 64;
    const int64_t t0 = lane * seg, t1 = (t0 + seg < n_tiles) ? t0 + seg : n_tiles;
    float run = 0.0f;
    for (int64_t t = t1 - 1; t >= t0; --t) {
        const int64_t i1 = (32 * t + 32 < n) ? 32 * t + 32 : n;
        for (int64_t i = 32 * t; i < i1; ++i) run = fmaxf(run, norm[i]);
        out[t] = run;
    }
    // carry[lane] = max of the segment totals of all higher lanes
    float carry = 0.0f;
    for (int l = 63; l > 0; --l) {
        const float v = __shfl(run, l);
        if (lane < l) carry = fmaxf(carry, v);
    }
    for (int64_t t = t0; t < t1; ++t) out[t] = fmaxf(out[t], carry);
}

extern "C" int pk_row_norm_bound_f32(void *stream, int64_t n, int32_t K, const double *src_dev, int64_t ld,
                                     float *out_dev) {
    PK_REQUIRE(n >= 1 && K >= 1 && ld >= K, "pk_row_norm_bound_f32: bad sizes");
    hipLaunchKernelGGL(row_norm_bound_kernel, dim3((unsigned)pk_ceil_div(n, 16)), dim3(256), 0, pk_stream(stream), n,
                       K, src_dev, ld, out_dev);
    PK_CHECK_LAUNCH("row_norm_bound_kernel");
    return PK_OK;
}

extern "C" int pk_tile_norm_bound_f32(void *stream, int64_t n_items, int32_t K, const double *V_dev, int64_t ld,
                                      float *work_dev, float *out_dev) {
    PK_REQUIRE(n_items >= 1 && K >= 1 && ld >= K, "pk_tile_norm_bound_f32: bad sizes");
    hipStream_t st = pk_stream(stream);
    hipLaunchKernelGGL(row_norm_bound_kernel, dim3((unsigned)pk_ceil_div(n_items, 16)), dim3(256), 0, st, n_items, K,
                       V_dev, ld, work_dev);
    PK_CHECK_LAUNCH("row_norm_bound_kernel");
    hipLaunchKernelGGL(tile_suffix_max_kernel, dim3(1), dim3(64), 0, st, n_items, pk_ceil_div(n_items, 32), work_dev,
                       out_dev);
    PK_CHECK_LAUNCH("tile_suffix_max_kernel");
    return PK_OK;
}

extern "C" int32_t pk_candidate_capacity(int32_t topk) {
    if (topk < 1) return 0;
    if (topk <= 10) return 16;
    if (topk <= 24) return 32;
    if (topk <= 52) return 64;
    return 0;
}

// L2 per XCD is 4 MiB: each launch streams an item chunk whose packed image (KQ KiB per 32-item
// tile) stays L2-resident while every resident workgroup sweeps it.
#define PK_CHUNK_BYTES (2560 * 1024)

// where a launch sequence sits in a two-phase sweep (all zero / null: the plain sweep)
struct SweepPhase {
    int tile_base;                  // first tile of the range the splits deal out (phase 2), else 0
    int slot_base;                  // first list / state slot of this launch sequence (phase 2: 1, the head owns slot 0)
    const LaneState *floor_state;   // phase 2: the head's lane records (threshold to start from), else nullptr
    int boot_tiles;                 // tiles of the threshold bootstrap in front of a sweep that starts cold (0: none)
    int shared;                     // single sweeps: the eight-wave workgroup with the V tiles staged in LDS
    int pair;                       // single sweeps with KC = 16: two user groups per wave (score_candidates_pair_kernel)
};

template <int NSTEP>
static int launch_candidates_n(hipStream_t st, int KC, int ablate, dim3 grid, const float4 *Vp, const float4 *Ep,
                               int64_t n_users, int n_items, int n_tiles, int split_tiles, int tiles_per_chunk,
                               const int64_t *seen_ptr, const unsigned long long *seen_tiles,
                               const int32_t *seen_ntiles, float *cs, int32_t *ci,
                               LaneState *st_lane, uint2 *st_ring, const float *user_bound, const float *tile_bound,
                               SeenDense dense, SweepPhase ph) {
    // Item chunks: tiles_per_chunk each while every group sweeps (the chunk's packed image stays in L2);
    // with pruning, groups leave the sweep early, so the chunks DOUBLE from launch to launch — the catalogue
    // is covered in O(log) launches and the few groups still sweeping late (low bandwidth demand) are not
    // cut into hundreds of launches (rank 200: 100-tile chunks, 157 launches for 500K items otherwise).
    // the dense seen masks: instances up to rank 128 — the single sweep, and the splits of a two-phase sweep (independent
    // item splits and higher ranks let the stream serve every tile; a dense request is then simply not used — the stream
    // cursor starts at the user's first record)
    constexpr bool DENSE_OK = (NSTEP <= 8);
    const bool use_dense = dense.tiles > 0 && DENSE_OK && (grid.y == 1 || ph.floor_state != nullptr);
    const SeenDense no_dense{nullptr, nullptr, 0};
    const SeenDense dn = use_dense ? dense : no_dense;
    int chunk_tiles = tiles_per_chunk;
    for (int chunk_begin = 0; chunk_begin < split_tiles; chunk_begin += chunk_tiles, chunk_tiles = (user_bound ? 2 * chunk_tiles : chunk_tiles)) {
#define PK_LAUNCH(KCV)                                                                                          \
    if (pk_score_lds_bytes(NSTEP, KCV) > 64 * 1024) {                                                           \
        static PkDeviceOnce attr_set;      /* one flag per (NSTEP, KC) instance of this macro expansion */        \
        if (attr_set.pending()) {                                                                                        \
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_candidates_kernel<NSTEP, KCV, false, false>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,                     \
                                                (int)pk_score_lds_bytes(NSTEP, KCV));                           \
            if (e1 == hipSuccess)                                                                               \
                e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_candidates_kernel<NSTEP, KCV, true, DENSE_OK>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,                            \
                                         (int)pk_score_lds_bytes(NSTEP, KCV));                                  \
            if (e1 != hipSuccess) {                                                                             \
                pk_set_error("pk_score_candidates_f32: cannot raise the LDS limit: %s", hipGetErrorString(e1)); \
                return PK_E_LAUNCH;                                                                             \
            }                                                                                                   \
            attr_set.done();                                                                                       \
        }                                                                                                       \
    }                                                                                                           \
    if constexpr (pk_shared_ok(NSTEP, KCV)) {                                                                   \
        if (ph.shared && grid.y == 1 && ph.floor_state == nullptr) {                                            \
            static PkDeviceOnce attr_set_s;                                                                        \
            if (attr_set_s.pending()) {                                                                                  \
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_candidates_kernel<NSTEP, KCV, false, DENSE_OK, true>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk_score_lds_bytes_shared(NSTEP, KCV)); \
                if (e2 != hipSuccess) {                                                                         \
                    pk_set_error("pk_score_candidates_f32: cannot raise the LDS limit (shared): %s", hipGetErrorString(e2)); \
                    return PK_E_LAUNCH;                                                                         \
                }                                                                                               \
                attr_set_s.done();                                                                                 \
            }                                                                                                   \
            hipLaunchKernelGGL((score_candidates_kernel<NSTEP, KCV, false, DENSE_OK, true>), dim3((grid.x * 4 + pk_shared_waves(KCV) - 1) / pk_shared_waves(KCV)), dim3(64 * pk_shared_waves(KCV)), \
                               pk_score_lds_bytes_shared(NSTEP, KCV), st, Vp, Ep, n_users,                      \
                               n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring,  \
                               user_bound, tile_bound, ablate, dn, 0, ph.slot_base, nullptr, ph.boot_tiles);    \
            continue;                                                                                           \
        }                                                                                                       \
    }                                                                                                           \
    if constexpr (KCV == 16 && NSTEP <= 8) {                                                                    \
        if (ph.pair && grid.y == 1 && ph.floor_state == nullptr) {                                              \
            const dim3 pgrid((grid.x + 1) / 2);                                                                 \
            const size_t plds = 2 * pk_score_lds_bytes(NSTEP, KCV);                                             \
            if (use_dense)                                                                                      \
                hipLaunchKernelGGL((score_candidates_pair_kernel<NSTEP, KCV, true>), pgrid, dim3(256), plds, st, Vp, Ep, n_users, \
                                   n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring, \
                                   user_bound, tile_bound, ablate, dn, ph.slot_base, ph.boot_tiles);            \
            else                                                                                                \
                hipLaunchKernelGGL((score_candidates_pair_kernel<NSTEP, KCV, false>), pgrid, dim3(256), plds, st, Vp, Ep, n_users, \
                                   n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring, \
                                   user_bound, tile_bound, ablate, no_dense, ph.slot_base, ph.boot_tiles);      \
            continue;                                                                                           \
        }                                                                                                       \
    }                                                                                                           \
    if (grid.y > 1 || ph.floor_state != nullptr)                                                                \
        hipLaunchKernelGGL((score_candidates_kernel<NSTEP, KCV, true, DENSE_OK>), grid, dim3(256), pk_score_lds_bytes(NSTEP, KCV), st, Vp, Ep, n_users, \
                           n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring,  \
                           user_bound, tile_bound, ablate, dn, ph.tile_base, ph.slot_base, ph.floor_state, ph.boot_tiles); \
    else if (use_dense)                                                                                         \
        hipLaunchKernelGGL((score_candidates_kernel<NSTEP, KCV, false, DENSE_OK>), grid, dim3(256), pk_score_lds_bytes(NSTEP, KCV), st, Vp, Ep, n_users, \
                           n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring,  \
                           user_bound, tile_bound, ablate, dn, 0, ph.slot_base, nullptr, ph.boot_tiles);        \
    else                                                                                                        \
        hipLaunchKernelGGL((score_candidates_kernel<NSTEP, KCV, false, false>), grid, dim3(256), pk_score_lds_bytes(NSTEP, KCV), st, Vp, Ep, n_users, \
                           n_items, n_tiles, split_tiles, chunk_begin, chunk_tiles, seen_ptr, seen_tiles, seen_ntiles, cs, ci, st_lane, st_ring,  \
                           user_bound, tile_bound, ablate, no_dense, 0, ph.slot_base, nullptr, ph.boot_tiles)
#ifdef PK_FAST_BUILD
        if (KC != 16) return PK_E_UNSUPPORTED;
        PK_LAUNCH(16);
#else
        switch (KC) {
            case 16:
                PK_LAUNCH(16);
                break;
            case 32:
                PK_LAUNCH(32);
                break;
            case 64:
                PK_LAUNCH(64);
                break;
            default:
                pk_set_error("pk_score_candidates_f32: KC=%d unsupported (16, 32, 64)", KC);
                return PK_E_UNSUPPORTED;
        }
#endif
#undef PK_LAUNCH
    }
    return PK_OK;
}

extern "C" int64_t pk_score_state_bytes(int64_t n_users, int32_t splits) {
    const int64_t slots = pk_ceil_div(n_users, 32) * (splits < 1 ? 1 : splits);
    return slots * 64 * (int64_t)sizeof(LaneState) + slots * RING * 64 * (int64_t)sizeof(uint2);
}

// How many item splits to use.  Every split pays its own threshold warm-up (measured on the
// ML-20M-shaped workload: 2 splits at 2.1 rounds of wave slots are NOT faster), so the catalogue is
// only dealt out when the user groups alone cannot fill the chip's ~2048-3072 wave slots even once
// (small request batches, the per-GPU share of a small user set at 8 GPUs); limited by the 64
// candidates the re-scoring wave can merge.  Used with and without the pruning bound.
extern "C" int32_t pk_score_splits(int64_t n_users, int32_t KC) {
    const int64_t groups = pk_ceil_div(n_users, 32);
    const int64_t slots = 256 * 8;
    int64_t s = (groups > 0) ? slots / groups : 1;
    const int smax = 64 / (KC > 0 ? KC : 64);
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int32_t)s;
}

// number of kernel launches pk_score_candidates_f32 issues for these arguments (item chunks; bench.py's
// per-pass traffic accounting multiplies the per-launch PMC averages by it)
extern "C" int32_t pk_score_chunk_launches(int64_t n_items, int32_t K, int32_t splits, int32_t tiles_per_chunk,
                                           int32_t pruned) {
    const int kq = pk_pack_kq(K);
    if (kq <= 0 || n_items < 1 || splits < 1) return 0;
    const int n_tiles = (int)pk_ceil_div(n_items, 32);
    const int split_tiles = (int)pk_ceil_div(n_tiles, splits);
    if (tiles_per_chunk <= 0) {
        tiles_per_chunk = PK_CHUNK_BYTES / (kq * 1024) / splits;
        if (tiles_per_chunk < 8) tiles_per_chunk = 8;
    }
    int n = 0, ct = tiles_per_chunk;
    for (int b = 0; b < split_tiles; b += ct, ct = (pruned ? 2 * ct : ct)) ++n;
    return n;
}

// one launch sequence (item chunks) of the candidate sweep over the lists / state slots [ph.slot_base, ph.slot_base + splits)
static int pk_sweep_launches(hipStream_t st, int64_t n_users, int64_t n_items, int n_tiles, int32_t K, const float *Vp_dev,
                             const float *Ep_dev, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                             const int32_t *seen_ntiles_dev, int32_t KC, int32_t splits, int32_t total_slots,
                             float *cand_score_dev, int32_t *cand_idx_dev, void *state_dev, int32_t tiles_per_chunk,
                             const float *user_bound_dev, const float *tile_bound_dev, SeenDense dense, SweepPhase ph) {
    const int kq = pk_pack_kq(K);
    const int nstep = pk_nstep(K);
    const int64_t groups = pk_ceil_div(n_users, 32);
    const int split_tiles = (int)pk_ceil_div(n_tiles - ph.tile_base, splits);
    if (tiles_per_chunk <= 0) {
        tiles_per_chunk = PK_CHUNK_BYTES / (kq * 1024) / splits;   // the S chunks of a launch share the L2
        if (tiles_per_chunk < 8) tiles_per_chunk = 8;
    }
    LaneState *st_lane = static_cast<LaneState *>(state_dev);
    uint2 *st_ring = reinterpret_cast<uint2 *>(st_lane + groups * total_slots * 64);
    if (ph.floor_state) ph.floor_state = st_lane;      // the head's records: slot 0
    // threshold bootstrap in front of every sweep that starts cold: 16 tiles (PK_SCORE_BOOT_TILES overrides, 0 = off)
    const char *boot_env = getenv("PK_SCORE_BOOT_TILES");
    ph.boot_tiles = ph.floor_state ? 0 : (boot_env ? atoi(boot_env) : 16);
    const char *shared_env = getenv("PK_SCORE_SHARED");     // tuning: 1 = the LDS-staged eight-wave instance for single sweeps
    ph.shared = shared_env ? atoi(shared_env) : 0;
    const char *pair_env = getenv("PK_SCORE_PAIR");         // tuning: 1 = two user groups per wave for single sweeps with KC = 16
    ph.pair = pair_env ? atoi(pair_env) : 0;
    dim3 grid((unsigned)pk_ceil_div(groups, 4), (unsigned)splits);
    const float4 *Vp = reinterpret_cast<const float4 *>(Vp_dev);
    const float4 *Ep = reinterpret_cast<const float4 *>(Ep_dev);
    int rc = PK_E_UNSUPPORTED;
#define PK_N_CASE(Q)                                                                                          \
    case Q:                                                                                                   \
        rc = launch_candidates_n<Q>(st, KC, ablate, grid, Vp, Ep, n_users, (int)n_items, n_tiles,            \
                                    split_tiles, tiles_per_chunk, seen_ptr_dev,                                   \
                                    reinterpret_cast<const unsigned long long *>(seen_tiles_dev), seen_ntiles_dev, \
                                    cand_score_dev, cand_idx_dev,                                              \
                                    st_lane, st_ring, user_bound_dev, tile_bound_dev, dense, ph);              \
        break;
    const char *abl_env = getenv("PK_SCORE_ABLATE");   // kernel-tuning knob, never set in production
    const int ablate = abl_env ? atoi(abl_env) : 0;
    switch (nstep) {
#ifdef PK_FAST_BUILD
        PK_N_CASE(4)
#else
        PK_N_CASE(1)
        PK_N_CASE(2)
        PK_N_CASE(3)
        PK_N_CASE(4)
        PK_N_CASE(5)
        PK_N_CASE(6)
        PK_N_CASE(7)
        PK_N_CASE(8)
        PK_N_CASE(10)
        PK_N_CASE(13)
        PK_N_CASE(16)
#endif
    }
#undef PK_N_CASE
    if (rc != PK_OK) return rc;
    PK_CHECK_LAUNCH("score_candidates_kernel");
    return PK_OK;
}

static int pk_score_check_args(const char *who, int64_t n_users, int64_t n_items, int32_t K, const float *Vp_dev,
                               const float *Ep_dev, const int64_t *seen_ptr_dev, const uint64_t *seen_tiles_dev,
                               const int32_t *seen_ntiles_dev, void *state_dev, const float *user_bound_dev,
                               const float *tile_bound_dev, const uint32_t *seen_dense_dev, const int32_t *seen_skip_dev,
                               int32_t dense_tiles) {
    PK_REQUIRE(n_users >= 1 && n_items >= 1 && n_items < 0x7fffff00LL, "%s: bad sizes", who);
    PK_REQUIRE(dense_tiles >= 0 && (dense_tiles == 0 || (seen_dense_dev && seen_skip_dev && seen_ptr_dev)),
               "%s: dense seen masks need seen_dense, seen_skip and the seen-tile stream", who);
    PK_REQUIRE(pk_pack_kq(K) > 0, "%s: K=%d unsupported (K <= 256)", who, K);
    PK_REQUIRE(((uintptr_t)Vp_dev % 16) == 0 && ((uintptr_t)Ep_dev % 16) == 0, "%s: alignment", who);
    PK_REQUIRE(state_dev != nullptr && ((uintptr_t)state_dev % 16) == 0, "%s: state buffer", who);
    PK_REQUIRE((seen_ptr_dev == nullptr) == (seen_tiles_dev == nullptr) &&
                   (seen_ptr_dev == nullptr) == (seen_ntiles_dev == nullptr),
               "%s: seen_ptr, seen_tiles, seen_ntiles go together (all or none)", who);
    PK_REQUIRE((user_bound_dev == nullptr) == (tile_bound_dev == nullptr),
               "%s: user_bound and tile_bound go together (both or neither)", who);
    return PK_OK;
}

extern "C" int pk_score_candidates_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K,
                                       const float *Vp_dev, const float *Ep_dev, const int64_t *seen_ptr_dev,
                                       const uint64_t *seen_tiles_dev, const int32_t *seen_ntiles_dev,
                                       int32_t KC, int32_t splits,
                                       float *cand_score_dev, int32_t *cand_idx_dev, void *state_dev,
                                       int32_t tiles_per_chunk, const float *user_bound_dev,
                                       const float *tile_bound_dev, const uint32_t *seen_dense_dev,
                                       const int32_t *seen_skip_dev, int32_t dense_tiles) {
    const int rc = pk_score_check_args("pk_score_candidates_f32", n_users, n_items, K, Vp_dev, Ep_dev, seen_ptr_dev,
                                       seen_tiles_dev, seen_ntiles_dev, state_dev, user_bound_dev, tile_bound_dev,
                                       seen_dense_dev, seen_skip_dev, dense_tiles);
    if (rc != PK_OK) return rc;
    PK_REQUIRE(splits >= 1 && splits * KC <= 64, "pk_score_candidates_f32: need 1 <= splits and splits*KC <= 64");
    SeenDense dense{seen_dense_dev, seen_skip_dev, seen_ptr_dev ? dense_tiles : 0};
    return pk_sweep_launches(pk_stream(stream), n_users, n_items, (int)pk_ceil_div(n_items, 32), K, Vp_dev, Ep_dev,
                             seen_ptr_dev, seen_tiles_dev, seen_ntiles_dev, KC, splits, splits, cand_score_dev, cand_idx_dev,
                             state_dev, tiles_per_chunk, user_bound_dev, tile_bound_dev, dense, SweepPhase{0, 0, nullptr, 0, 0});
}

// ---- two-phase sweep -------------------------------------------------------------------------------------------------
// The S + 1 candidate lists of a user (head + splits, work arrays [list][n_pad users][KC]) -> ONE list of the KC best, by
// (score descending, item ascending) with exact comparisons: one wave per user, bitonic network over 64 * SLOTS >=
// (S + 1) * KC elements.  The lists cover disjoint tiles, so no item appears twice.  What the merged list bounds: an item
// no list kept scored at most the KC-th entry of the list that dropped it (or the head's, which every split starts from),
// and the merged KC-th entry is at least each of those — the re-scoring kernel certifies against it as against the
// list of a single sweep (same key-sort tolerance: the merge itself compares exactly).
template <int SLOTS>
__global__ __launch_bounds__(256) void merge_candidates_kernel(int64_t n_users, int64_t n_pad, int KC, int lists,
                                                               const float *__restrict__ work_score,
                                                               const int32_t *__restrict__ work_idx,
                                                               float *__restrict__ out_score, int32_t *__restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t user = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (user >= n_pad) return;
    float key[SLOTS];
    int val[SLOTS];
    const int total = lists * KC;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        key[s] = -INFINITY;
        val[s] = PK_IDX_NONE;
        if (i < total && user < n_users) {
            const int64_t at = ((int64_t)(i / KC) * n_pad + user) * KC + (i % KC);
            const int iv = work_idx[at];
            if (iv >= 0) {
                key[s] = work_score[at];
                val[s] = iv;
            }
        }
    }
    // the common case: nothing beat the head's list in the tail (ML-20M-shaped: 0.1 pushes per user beyond tile 32) — the
    // head's list goes out as it is (the re-scoring kernel orders by exact scores and reads the bound from the last slot,
    // exactly as it does for the list a single sweep leaves)
    bool extra = false;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) extra = extra || (lane + 64 * s >= KC && val[s] != PK_IDX_NONE);
    if (!__any(extra)) {
        if (lane < KC) {
            out_score[user * KC + lane] = (user < n_users) ? work_score[user * KC + lane] : -INFINITY;
            out_idx[user * KC + lane] = (user < n_users) ? work_idx[user * KC + lane] : -1;
        }
        return;
    }
    // a precedes b: larger score, then smaller item id (empty entries: -inf, PK_IDX_NONE -> last)
    auto before = [](float ka, int va, float kb, int vb) { return ka > kb || (ka == kb && va < vb); };
    constexpr int N = 64 * SLOTS;
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int ds = j >> 6;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    if ((s & ds) == 0 && (s | ds) < SLOTS) {
                        const int i = lane + 64 * s;
                        const bool desc = (i & k) == 0;
                        const int s2 = s | ds;
                        const bool in_order = before(key[s], val[s], key[s2], val[s2]);
                        if (in_order != desc) {
                            const float tk = key[s];
                            const int tv = val[s];
                            key[s] = key[s2];
                            val[s] = val[s2];
                            key[s2] = tk;
                            val[s2] = tv;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int i = lane + 64 * s;
                    const float ok = __shfl_xor(key[s], j, 64);
                    const int ov = __shfl_xor(val[s], j, 64);
                    const bool desc = (i & k) == 0;
                    const bool lower = (lane & j) == 0;                 // I hold the smaller index of the pair
                    const bool mine_first = before(key[s], val[s], ok, ov);
                    // the smaller index keeps the element that precedes in a descending run, the other one otherwise
                    const bool keep = (mine_first == (lower == desc));
                    if (!keep) {
                        key[s] = ok;
                        val[s] = ov;
                    }
                }
            }
        }
    }
    if (lane < KC) {     // KC <= 64: the best KC sit in slot 0
        int iv = (val[0] == PK_IDX_NONE) ? -1 : val[0];
        // the head's "not full although bootstrapped" mark survives the merge (only the head sweep can set it)
        if (lane == KC - 1 && iv < 0 && user < n_users && work_idx[user * KC + KC - 1] == PK_IDX_FLOOR) iv = PK_IDX_FLOOR;
        out_score[user * KC + lane] = key[0];
        out_idx[user * KC + lane] = iv;
    }
}

// Default shape of the two-phase sweep: *head_tiles = 0 when a single sweep is the better choice.
// Measured (profiles/r03_sweep_variants_*.txt, ML-20M-shaped rank 50 / top-10): a 17 312-user shard (541 groups: the share
// of one GPU of eight) sweeps in 0.274 ms as one sweep per group, 0.246 with the threshold bootstrap, 0.181 with a
// 32-tile head + 3 seeded splits, 0.162 with 7 — its pass goes from 0.367 to 0.254 ms; at full size (4 328 groups, the
// wave slots are full 1.4 times over) the same shapes are SLOWER (0.37 -> 0.43 - 0.48 ms; S-1M 1.71 -> 2.51 ms): there the
// kernel is bound by the work of the head tiles, where the lists are built, not by its longest chain, and the extra
// launches, waves and the merge only cost.  So the scheme is used when the groups alone leave wave slots idle:
//   groups <= 1024 (32K users);  head = 32 tiles;  splits = 3 / 7 / 15 for > 768 / > 256 / fewer groups, capped by the
//   (splits + 1) * KC <= 256 entries the merge holds (KC = 16: 15, 32: 7, 64: 3).
// PK_SCORE_HEAD_TILES / PK_SCORE_PHASE2_SPLITS override (tuning and tests: any n_users); PK_SCORE_HEAD_TILES=0 switches it off.
extern "C" int pk_score_two_phase_plan(int64_t n_users, int64_t n_items, int32_t KC, int32_t *head_tiles, int32_t *splits) {
    PK_REQUIRE(head_tiles && splits, "pk_score_two_phase_plan: null output");
    const int64_t n_tiles = pk_ceil_div(n_items, 32);
    const int64_t groups = pk_ceil_div(n_users, 32);
    int h = (groups <= 1024) ? 32 : 0;
    int s = groups > 768 ? 3 : (groups > 256 ? 7 : 15);
    if (const char *e = getenv("PK_SCORE_HEAD_TILES")) h = atoi(e);
    if (const char *e = getenv("PK_SCORE_PHASE2_SPLITS")) s = atoi(e);
    if (s < 1) s = 1;
    while (s > 1 && (s + 1) * KC > 256) --s;
    if (KC < 1 || KC > 64 || h < 1 || n_tiles < 4 * (int64_t)h) h = 0;     // short catalogues: the tail is no longer than the head
    *head_tiles = h;
    *splits = h ? s : 0;
    return PK_OK;
}

// The pruned candidate sweep in two phases (see the kernel header): tiles [0, head_tiles) by one sweep per group, the rest
// dealt round-robin to `splits` sweeps per group that start from the head's thresholds, then the merge.  Needs the
// pruning bounds (without them nobody leaves early and a single sweep is the shortest chain there is).
//   work_score / work_idx  [(splits + 1) * n_pad * KC]   the raw lists (slot 0: head), n_pad = n_users rounded up to 32
//   cand_score / cand_idx  [n_pad * KC]                   the merged list: what pk_rescore_topk_* takes with splits = 1
//   state                  pk_score_state_bytes(n_users, splits + 1)
extern "C" int pk_score_two_phase_f32(void *stream, int64_t n_users, int64_t n_items, int32_t K,
                                      const float *Vp_dev, const float *Ep_dev, const int64_t *seen_ptr_dev,
                                      const uint64_t *seen_tiles_dev, const int32_t *seen_ntiles_dev,
                                      int32_t KC, int32_t head_tiles, int32_t splits,
                                      float *work_score_dev, int32_t *work_idx_dev,
                                      float *cand_score_dev, int32_t *cand_idx_dev, void *state_dev,
                                      int32_t tiles_per_chunk, const float *user_bound_dev,
                                      const float *tile_bound_dev, const uint32_t *seen_dense_dev,
                                      const int32_t *seen_skip_dev, int32_t dense_tiles) {
    int rc = pk_score_check_args("pk_score_two_phase_f32", n_users, n_items, K, Vp_dev, Ep_dev, seen_ptr_dev,
                                 seen_tiles_dev, seen_ntiles_dev, state_dev, user_bound_dev, tile_bound_dev,
                                 seen_dense_dev, seen_skip_dev, dense_tiles);
    if (rc != PK_OK) return rc;
    const int n_tiles = (int)pk_ceil_div(n_items, 32);
    PK_REQUIRE(user_bound_dev && tile_bound_dev, "pk_score_two_phase_f32: needs the pruning bounds");
    PK_REQUIRE(KC >= 1 && KC <= 64 && splits >= 1 && (splits + 1) * KC <= 256 && head_tiles >= 1 && head_tiles < n_tiles,
               "pk_score_two_phase_f32: need 1 <= head_tiles < n_tiles and (splits + 1) * KC <= 256");
    PK_REQUIRE(work_score_dev && work_idx_dev && cand_score_dev && cand_idx_dev, "pk_score_two_phase_f32: null list buffers");
    hipStream_t st = pk_stream(stream);
    SeenDense dense{seen_dense_dev, seen_skip_dev, seen_ptr_dev ? dense_tiles : 0};
    const int total_slots = splits + 1;
    // phase 1: the head, a complete single sweep over a catalogue that ends at tile head_tiles (ONE launch: it finalises —
    // rings merged, lists written, exit tile and threshold in the lane records — so the head must fit one item chunk)
    rc = pk_sweep_launches(st, n_users, n_items, head_tiles, K, Vp_dev, Ep_dev, seen_ptr_dev, seen_tiles_dev, seen_ntiles_dev,
                           KC, 1, total_slots, work_score_dev, work_idx_dev, state_dev, head_tiles, user_bound_dev,
                           tile_bound_dev, dense, SweepPhase{0, 0, nullptr, 0, 0});
    if (rc != PK_OK) return rc;
    // phase 2: the splits, from the head's thresholds
    LaneState *st_lane = static_cast<LaneState *>(state_dev);
    rc = pk_sweep_launches(st, n_users, n_items, n_tiles, K, Vp_dev, Ep_dev, seen_ptr_dev, seen_tiles_dev, seen_ntiles_dev,
                           KC, splits, total_slots, work_score_dev, work_idx_dev, state_dev, tiles_per_chunk, user_bound_dev,
                           tile_bound_dev, dense, SweepPhase{head_tiles, 1, st_lane, 0, 0});
    if (rc != PK_OK) return rc;
    const int64_t n_pad = pk_ceil_div(n_users, 32) * 32;
    const int slots = (int)pk_ceil_div((int64_t)total_slots * KC, 64);
    dim3 grid((unsigned)pk_ceil_div(n_pad, 4));
#define PK_MERGE(SL)                                                                                               \
    hipLaunchKernelGGL((merge_candidates_kernel<SL>), grid, dim3(256), 0, st, n_users, n_pad, KC, total_slots,     \
                       work_score_dev, work_idx_dev, cand_score_dev, cand_idx_dev)
    if (slots <= 1) PK_MERGE(1);
    else if (slots == 2) PK_MERGE(2);
    else PK_MERGE(4);
#undef PK_MERGE
    PK_CHECK_LAUNCH("merge_candidates_kernel");
    return PK_OK;
}

// The rows entries of the product (round 5: the sweep builds the users' fragments from the rows of E) do not exist in this
// frozen tree; the Python layer keeps the packed route under a probe library (ops.sweep_takes_rows), the symbols are here so
// that the product's driver.hip links.
extern "C" int pk_sweep_takes_rows(void) { return 0; }      // the experiment tree takes packed fragments only

extern "C" int pk_score_candidates_rows_f32(void *, int64_t, int64_t, int32_t, const float *, const double *, int64_t, const double *,
                                            int64_t, double, const int64_t *, const uint64_t *, const int32_t *, int32_t, int32_t,
                                            float *, int32_t *, void *, int32_t, const float *, const uint32_t *, const int32_t *,
                                            int32_t) {
    pk_set_error("pk_score_candidates_rows_f32: not part of the experiment tree (probe library)");
    return PK_E_UNSUPPORTED;
}
extern "C" int pk_score_two_phase_rows_f32(void *, int64_t, int64_t, int32_t, const float *, const double *, int64_t, const double *,
                                           int64_t, double, const int64_t *, const uint64_t *, const int32_t *, int32_t, int32_t,
                                           int32_t, float *, int32_t *, float *, int32_t *, void *, int32_t, const float *,
                                           const uint32_t *, const int32_t *, int32_t) {
    pk_set_error("pk_score_two_phase_rows_f32: not part of the experiment tree (probe library)");
    return PK_E_UNSUPPORTED;
}

// eager load of this translation unit's code object (pk_warm_up, api.cpp): the runtime loads a code object at the first
// launch of one of its kernels — or when a kernel's attributes are asked for, which costs no launch
hipError_t pk_tu_load_score() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&pack_frag_kernel));
}
