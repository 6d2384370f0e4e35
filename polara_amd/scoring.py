"""Recommendation pipeline on device: fold-in -> fused MFMA scoring/masking/top-k candidates ->
exact fp64 re-scoring -> (rare) exact rows.  Replaces the chunk loop of
`RecommenderModel.get_recommendations` (models.py:391-405) and `_slice_recommender`
(models.py:359-371); no `[chunk x n_items]` score matrix is ever materialised, so there is no
chunking by host memory (utils.py:16-53) either.

Inputs follow the reference's protocol: the test triplet of `_get_test_data` (models.py:227-257)
turned into ONE canonical CSR whose explicit zeros are kept — zero-feedback entries contribute
nothing to the fold-in (the reference drops them from `test_matrix`, models.py:198-203) but still
count as seen (they stay in `slice_data`, models.py:494-519).
"""
import threading

import numpy as np
import torch

from . import _lib
from .csr import coo_to_csr

EXACT_ROWS_BYTES = 2 << 30  # work-buffer budget per launch of the exact path
MAX_FUSED_RANK = 256        # largest rank the MFMA candidate sweep is instantiated for (csrc/score.hip)
PACKED_FOLD_IN = True       # the approximate fold-in gathers the packed (one line per rank-50 row) image where one exists
# ... for lists of at most this many entries: the packed image's error weights are ~40x an fp32 rounding, and every one of a
# list's topk gaps must clear them — at top-10 0.9 % of the users are re-folded exactly (ML-20M-shaped), at rank 100 / top-20
# 6.5 % (a wash: 1.62 ms per pass either way), at rank 200 / top-50 46 % (S-50M shard: fold-in 4.9 -> 2.9 ms, but re-fold and
# second re-scoring 1.7 -> 8.2 ms: 45.7 -> 50.3 ms per pass) — so longer lists keep the fp32 image
PACKED_MAX_TOPK = 20
# the sweep reads the users' side from the rows of E when they are aligned (no packing launch); False: the packing launch (A/B runs, tests)
SWEEP_FROM_ROWS = True


class FactorImage:
    """Item factors resident in HBM in both forms the scoring kernels read: fp64 row-major
    (re-scoring, fold-in) and fp32 MFMA-fragment packed (candidate pass)."""

    def __init__(self, ops, V):
        self.ops = ops
        self.V = V.contiguous()
        self.n_items, self.K = self.V.shape
        # the fused sweep's MFMA instances stop at rank 256; beyond it every user goes through the exact fp64 row
        # kernel (any rank) — the same lists, no fragment image needed
        self.fused = self.K <= MAX_FUSED_RANK
        self.Vp = ops.pack_frag(self.V) if self.fused else None
        if not self.fused:
            self.vmax = float(torch.linalg.vector_norm(self.V, dim=1).max().item())
            if not np.isfinite(self.vmax) or (self.vmax != 0.0 and not 1e-30 < self.vmax < 1e30):
                raise ValueError('item factors with max row norm %g are outside the range the scoring kernels work in' % self.vmax)
            self.Q20 = self.tile_bound = self.V32x = self.vnorm = None
            self.Kx = self.K
            return
        self.vnorm = ops.row_norm_bound(self.V)       # fp32 upper bounds of the rows' norms (fold-in weight column, certification)
        # fp32 image for the approximate fold-in: columns 0..K-1 = fl32(V), column K = an upper bound of the
        # row norm (so the same product also yields w_u = sum_j a_uj ||V_j||, the weight of the fold-in's
        # rounding error), zero padding to a multiple of 4 columns (one 16-byte load = 4 columns).
        # Rows start on cache-line boundaries (stride = Kx rounded up to 32 floats = 128 B, or to a power of two
        # below that): the fold-in is bound by the number of lines its row gathers touch, and a 208-byte row at
        # stride 208 straddles 2.6 lines on average instead of 2 (measured on S-1M: 2.00 -> 1.86 ms per fold-in).
        # ONE launch builds it and returns the two numbers of the range check (ops.v32_image): the largest row-norm bound
        # and whether everything is finite.
        self.Kx = -(-(self.K + 1) // 4) * 4
        ld = -(-self.Kx // 32) * 32 if self.Kx > 16 else (4 if self.Kx <= 4 else 8 if self.Kx <= 8 else 16)
        if hasattr(ops, 'v32_image'):
            image, self.vmax, finite = ops.v32_image(self.V, self.vnorm, ld)
        else:          # (the CPU double of the tests)
            image = torch.zeros(self.n_items, ld, dtype=torch.float32, device=self.V.device)
            image[:, :self.K] = self.V.to(torch.float32)
            image[:, self.K] = self.vnorm
            self.vmax = float(torch.linalg.vector_norm(self.V, dim=1).max().item())
            finite = bool(np.isfinite(self.vmax))
        if not finite or (self.vmax != 0.0 and not 1e-30 < self.vmax < 1e30):
            # the candidate sweep and the approximate fold-in work on fp32 images of the factors; their error
            # bounds are norm-wise (2^-24 * max||V_i||) and hold as long as that scale is an fp32 NORMAL number
            raise ValueError('item factors with max row norm %g are outside the range the fp32 candidate sweep '
                             'works in (rescale the factors)' % (self.vmax if finite else float('nan')))
        self.Q20 = None
        self.tile_bound = ops.tile_norm_bound(self.V)   # exact pruning bound of the candidate sweep
        self.V32x = image[:, :self.Kx]
        # packed image for the approximate fold-in (csrc/foldq.hip): 20-bit block fixed point, half the bytes and lines
        # per gathered entry of the fp32 image; its own error weights D_j (exact, from the bits written) take the
        # place of the norm column, so column K of the product is again a w_u with ||E' - E|| <= 2^-24 w_u
        self.Q20 = ops.q20_encode(self.V) if (PACKED_FOLD_IN and hasattr(ops, 'q20_encode')) else None


def test_csr_from_triplet(test_data, shape, weights=None):
    """(user_idx, item_idx, feedback) sorted by user -> canonical CSR arrays keeping zeros.
    `weights`, if given, replaces the feedback values (CoFFee's per-entry coefficient)."""
    users, items, fdbk = test_data
    vals = np.asarray(fdbk if weights is None else weights, dtype=np.float64)
    return coo_to_csr(users, items, vals, shape, sum_duplicates=True)


ORDER_USERS_MIN = 8192      # below this a pass is a handful of workgroups: nothing to balance
HEAD_USERS = 0   # users of the head batch of an activity-ordered pass (0: none; measured slower, kept for the tests of the batching)
_in_pass = threading.local()


def renumbered_test_rows(ops, users):
    """Device int64 row number of every test entry: `users` (host array, sorted by user) as they are when they run
    0, 1, 2, ... without gaps, else renumbered that way — what models.py:244-255 does with `np.unique(..., return_inverse)` on
    the host.  Raises like the protocol when the set is not sorted by user.  One flag pair comes back from the device."""
    ud = ops.to_device(np.ascontiguousarray(users, dtype=np.int64))
    if ud.numel() < 2:
        return torch.zeros_like(ud)
    step = ud[1:] - ud[:-1]
    unsorted, gaps = torch.stack([(step < 0).any(), (step > 1).any() | (ud[0] != 0)]).tolist()
    if unsorted:
        raise AssertionError('the test set must be sorted by users')
    if not gaps:
        return ud
    rows = torch.zeros_like(ud)
    rows[1:] = torch.cumsum((step > 0).to(torch.int64), 0)
    return rows


def recommend(ops, factors, T, topk, filter_seen=True, return_scores=False, stats=None, prune=True, batches=None,
              approx_fold_in=None, order_users=True, head_users=None, two_phase_ok=True, out=None):
    """factors: FactorImage; T: ops-level CSR of the test users [n_users x n_items].
    Returns int64 device tensor [n_users x topk] (+ fp64 scores), rows in test-user order,
    columns by descending score — the contract of models.py:400-405.
    out (ids only): where the lists are to end up — a device tensor or a PINNED HOST tensor [n_users x topk] int64; the last
    kernel of the pass writes it (ops.scatter_rows: the host-side array of the reference's contract without a copy-engine
    transfer behind the pass) and it is what the call returns.

    approx_fold_in (default: on when only the ids are asked for and the feedback is non-negative): the fold-in
    E = T V gathers the fp32 image of V (half the bytes of the product that is bound by them), and so does the
    first re-scoring of the candidates; the re-scoring kernel then knows every score to within
    delta_u = 2^-24 (w_u + ||E'_u||) max||V_i|| and certifies the ORDER only where consecutive scores are further
    apart than 2 delta_u; the (few) other users get their E row recomputed from the fp64 factors and are re-scored
    exactly, against the fp64 item rows.  The returned ids are those of the exact pipeline either way."""
    lock = getattr(ops, 'pass_lock', None)
    if lock is not None and not getattr(_in_pass, 'held', False):
        # two host threads driving one ops object (the reference parallelises its chunk loop with a thread pool,
        # models.py:374-382) would interleave their launches on the same stream and share the sweep's parked state and
        # the exact path's work buffer: a pass is enqueued as a whole (enqueueing costs ~0.2 ms; the GPU is not waited for)
        with lock:
            _in_pass.held = True
            try:
                return recommend(ops, factors, T, topk, filter_seen, return_scores, stats, prune, batches, approx_fold_in,
                                 order_users, head_users, two_phase_ok, out)
            finally:
                _in_pass.held = False
    n_users, n_items = T.shape
    if out is not None and (return_scores or tuple(out.shape) != (n_users, topk) or out.dtype != torch.int64 or not out.is_contiguous()
                            or not (out.is_cuda or out.is_pinned())):
        raise ValueError('recommend: `out` takes the ids only: a contiguous int64 [n_users x topk] device or pinned host tensor')
    if n_items != factors.n_items:
        raise ValueError('test matrix and item factors disagree on the number of items')
    if topk > n_items:
        raise ValueError('kth(=%d) out of bounds (%d)' % (n_items - topk, n_items))  # numpy argpartition's error
    if order_users and prune and factors.fused and n_users >= ORDER_USERS_MIN and hasattr(T, 'by_activity'):
        # A wave sweeps the catalogue for 32 consecutive users until the LAST of them can be pruned, so users are
        # grouped by activity (the row order is a cached image of the test matrix); rows go back to their places at
        # the end.  ML-20M-shaped: 12.3 -> 9.5 % of the tiles scored, sweep -14 %, fold-in -17 % (long rows first).
        Tp, perm = T.by_activity()
        if head_users is None:
            head_users = HEAD_USERS
        res = recommend(ops, factors, Tp, topk, filter_seen, return_scores, stats, prune, batches, approx_fold_in,
                        order_users=False, head_users=head_users, two_phase_ok=two_phase_ok)
        if return_scores:
            idx_p, sc_p = res
            out_idx, out_s = torch.empty_like(idx_p), torch.empty_like(sc_p)
            out_idx[perm] = idx_p
            out_s[perm] = sc_p
            return out_idx, out_s
        if out is not None:
            return ops.scatter_rows(res, perm, out=out)
        return ops.scatter_rows(res, perm) if hasattr(ops, 'scatter_rows') else torch.empty_like(res).index_copy_(0, perm, res)
    KC = ops.candidate_capacity(topk) if factors.fused else 0
    K = factors.K
    if KC == 0:
        E = ops.spmm(T, factors.V)                   # fold-in, fp64 (K4)
        # topk beyond the fused kernel's 52, or a rank beyond its 256: every user goes through the exact fp64 row
        # kernel (all items scored, two-class key) — slow but the same contract
        seen_ptr = T.indptr if filter_seen else None
        seen_idx = T.indices if filter_seen else None
        out_idx = torch.empty(n_users, topk, dtype=torch.int64, device=E.device)
        out_s = torch.empty(n_users, topk, dtype=torch.float64, device=E.device)
        per = max(1, int(EXACT_ROWS_BYTES // (n_items * 9 + 16)))
        for s0 in range(0, n_users, per):
            sub = torch.arange(s0, min(n_users, s0 + per), dtype=torch.int32, device=E.device)
            ex_idx, ex_s = ops.score_exact_rows(sub, factors.V, E, n_items, seen_ptr, seen_idx, topk)
            out_idx[s0:s0 + len(sub)] = ex_idx
            out_s[s0:s0 + len(sub)] = ex_s
        if stats is not None:
            stats.update(flagged_users=n_users, candidate_capacity=0, item_splits=0)
        return (out_idx, out_s) if return_scores else out_idx
    if approx_fold_in is None:
        approx_fold_in = not return_scores
    approx_fold_in = bool(approx_fold_in) and not return_scores and factors.Kx <= 256 and T.nonneg()
    Kx = factors.Kx if approx_fold_in else K
    Ex = ops.empty(n_users, Kx)
    E = Ex[:, :K]                       # row stride Kx: every kernel below takes a leading dimension
    seen_ptr = T.indptr if filter_seen else None
    seen_idx = T.indices if filter_seen else None
    seen_tiles = T.seen_tiles() if filter_seen else None
    # (masks, skip counts, tiles) of the head of the catalogue: where a PRUNED sweep spends its time (a full sweep would
    # only pay for the second kernel instance: 1.69 -> 1.80 ms on ML-20M-shaped)
    seen_dense = T.seen_dense() if (filter_seen and prune and hasattr(T, 'seen_dense')) else None
    splits = ops.score_splits(n_users, KC, prune)    # item ranges per user group (1 when pruning / users fill the chip)
    use_two_phase = bool(prune and two_phase_ok and hasattr(ops, 'two_phase_plan') and not getattr(ops, 'score_splits_override', 0))
    two_phase = ops.two_phase_plan(n_users, n_items, KC) if use_two_phase else (0, 0)
    out_idx = torch.empty(n_users, topk, dtype=torch.int64, device=E.device)
    out_s = torch.empty(n_users, topk, dtype=torch.float64, device=E.device)
    flags = torch.empty(n_users, dtype=torch.int32, device=E.device)

    refolded = []                        # device counters of the users re-done with an exact fold-in
    # The lists of users to re-do are built by the re-scoring kernel itself, where the flags are (ops.rescore_topk
    # `flagged`): counter 0 belongs to the pass's final list (users for the exact-row kernel, global ids, every batch
    # appends), counter 1 + b to batch b's re-fold list.  One launch zeroes them all; the flag compactions (two launches
    # per list) are gone from the pass.  Backends without the fused form (test doubles) keep `flag_compact`.
    fused_lists = hasattr(ops, 'zero_counters')        # (the CPU double of the tests keeps the separate compactions)
    final_list = final_cnt = counters = None
    batch_no = [0]

    def run_batch(u0, u1):
        """fold-in -> bounds/pack -> candidate sweep -> exact re-scoring of users [u0, u1) on the current stream"""
        nb = u1 - u0
        splits = ops.score_splits(nb, KC, prune)     # of THIS batch: a small head batch is dealt out over item splits
        two_phase = ops.two_phase_plan(nb, n_items, KC) if use_two_phase else (0, 0)     # ... or swept in two phases
        if approx_fold_in:
            if factors.Q20 is not None and PACKED_FOLD_IN and topk <= PACKED_MAX_TOPK:
                ops.fold_q20(T, factors.Q20, K, out=Ex, rows=(u0, u1))     # fold-in against the packed image (K4q)
            else:
                ops.spmm(T, factors.V32x, out=Ex, rows=(u0, u1))           # fold-in against fl32(V) (K4)
            w = Ex[u0:u1, K]                                               # w_u: ||E' - E|| <= 2^-24 w_u (strided view)
        else:
            ops.spmm(T, factors.V, out=Ex, rows=(u0, u1))                  # fold-in, fp64 (K4)
            w = None
        Eb = E[u0:u1]
        # fragments of E for the MFMA sweep + the users' side of the exact Cauchy-Schwarz pruning bound (a group
        # of 32 users leaves the sweep once no later item can beat any of its thresholds), one pass over E;
        # with the approximate fold-in ||E|| <= ||E'|| + 2^-24 w.  `prune=False` forces the full sweep (same
        # result, tuning / tests only)
        # Round 5: when the rows of E are aligned (the approximate fold-in's [n x Kx] block always is) the SWEEP builds both in
        # the prologue of its waves, bit for bit what the packing kernel writes: no packing launch, no packed copy of E.
        rows_kw = {}
        # (pruned passes: they are ONE launch; a full sweep is cut into item chunks and every launch would rebuild the
        # fragments — S-1M unpruned: 39.1 -> 39.7 ms — so it keeps the packed copy)
        if SWEEP_FROM_ROWS and prune and hasattr(ops, 'sweep_takes_rows') and ops.sweep_takes_rows(Eb):
            Ep = ub = None
            rows_kw = {'E_rows': (Eb, w, 1.2e-7)}
        else:
            Ep, ub = ops.pack_frag_bound(Eb, extra=w, extra_scale=1.2e-7)
        if not prune:
            ub = None
        sp = seen_ptr[u0:u1 + 1] if filter_seen else None
        st = (seen_tiles[0], seen_tiles[1][u0:u1]) if seen_tiles is not None else None
        assert u0 % 32 == 0, 'user batches start on a 32-user group boundary (dense seen masks are indexed by group)'
        sd = (seen_dense[0][u0 // 32:], seen_dense[1][u0:u1], seen_dense[2]) if seen_dense is not None else None
        extra = {} if sd is None else {'seen_dense': sd}
        if two_phase[0] and prune:
            # pruned sweep in two phases: head of the catalogue for every group, then item splits that start from the
            # head's thresholds, lists merged (K3; the chain of a group is head + tail / S tiles instead of head + tail)
            cs, ci = ops.score_two_phase(factors.Vp, Ep, nb, n_items, K, sp, KC, two_phase[0], two_phase[1], ub,
                                         factors.tile_bound, seen_tiles=st, seen_dense=sd, **rows_kw)
            splits = 1                                                                        # ONE merged list per user
        else:
            cs, ci = ops.score_candidates(factors.Vp, Ep, nb, n_items, K, sp, seen_idx, KC, splits,
                                          user_bound=ub, tile_bound=factors.tile_bound if prune else None,
                                          seen_tiles=st, **extra, **rows_kw)                  # K3
        outs = (out_idx[u0:u1], out_s[u0:u1], flags[u0:u1])
        to_final = (final_list, final_cnt, u0) if fused_lists else None
        if approx_fold_in and fused_lists:
            b = batch_no[0]
            batch_no[0] += 1
            lst, cnt = torch.empty(nb, dtype=torch.int32, device=E.device), counters[1 + b:2 + b]
            first = (lst, cnt, 0)
        else:
            first = to_final
        ops.rescore_topk(factors.V, Eb, n_items, sp, KC, cs, ci, topk, factors.vmax, want_scores=True,
                         splits=splits, out=outs, e_err=w, v32=factors.V32x if approx_fold_in else None, **(
                             {'flagged': first, 'item_norm': factors.vnorm if approx_fold_in else None} if fused_lists else {}))
        if approx_fold_in:
            # every flagged user — order not certified at the accuracy of the approximate fold-in (bit 4), or
            # bound for the exact-row kernel anyway (bits 1, 2), which must not see an approximate E — gets its
            # E row recomputed from the fp64 factors and is re-scored; the list of those users never leaves the
            # device (no host round trip inside the pass).  Who is STILL flagged after that goes to the final list.
            if not fused_lists:
                lst, cnt = ops.flag_compact(outs[2], 7)
            if hasattr(ops, 'spmm_flagged') and ops.spmm_flagged_ok(factors.V):
                # the product's own row tasks (mapping, summation order) on the LISTED users: thousands of them (the packed
                # image's weights are ~40x an fp32 rounding) cost what their entries cost — not a workgroup with barriers
                # per row (fold_rows: 86 us for 3 320 users of an ML-20M-shaped pass), nor a wave per task of EVERY row
                # (the flag-predicated launch of the whole plan: 0.23 ms of early exits on S-1M)
                if hasattr(ops, 'spmm_rows_list'):
                    ops.spmm_rows_list(T, factors.V, Ex, lst, cnt, flags, 7, rows=(u0, u1))
                else:
                    ops.spmm_flagged(T, factors.V, Ex, flags, 7, rows=(u0, u1))
            else:
                ops.fold_rows(T, lst, cnt, factors.V, Ex, row_offset=u0)
            ops.rescore_topk(factors.V, Eb, n_items, sp, KC, cs, ci, topk, factors.vmax, want_scores=True,
                             splits=splits, out=outs, rows=lst, n_rows_dev=cnt, e_err=w, e_exact=True, **(
                                 {'flagged': to_final} if fused_lists else {}))
            refolded.append(cnt)

    # User batches are independent: with B > 1 they run round-robin on two side streams.  Measured on
    # MI355X (S-1M): the fold-in SpMM of one batch and the MFMA sweep of another hardly overlap (B = 2:
    # -4 %, B = 4: -1 %, B = 8: +55 % per pass — each kernel fills the chip on its own), so batching is
    # only used to bound the temporaries of very large user sets (4M users per batch).
    B = int(batches) if batches else -(-n_users // (1 << 22))
    B = max(1, min(B, n_users // 4096)) if n_users >= 4096 else 1
    # batch starts must be multiples of 128 users (one workgroup; the dense seen masks are addressed by 32-user group)
    head = (int(head_users or 0) // 128) * 128
    if stats is not None:
        B, head = 1, 0                       # sweep statistics are read from the (single) state buffer
    if fused_lists:
        counters = ops.zero_counters(2 + max(B, 2))          # on the calling stream, before any batch stream forks from it
        final_cnt = counters[0:1]
        final_list = torch.empty(n_users, dtype=torch.int32, device=E.device)
    if B == 1 and not (0 < head and 4 * head <= n_users):
        run_batch(0, n_users)
    else:
        if B == 1:
            # the first `head` users on their own: with the users in activity order they are the ones whose groups stay
            # longest in the sweep (mean exit tile 129 against 70, max 271 against ~130 of 836 on ML-20M-shaped) and the
            # sweep kernel lasts as long as its slowest wave; a batch this small gets item splits (ops.score_splits), its
            # chains are S times shorter and run next to the bulk of the users on the other stream
            ranges = [(0, head), (head, n_users)]
        else:
            per = -(-(-(-n_users // B)) // 128) * 128          # batch size, multiple of 128 users (one workgroup)
            ranges = [(u0, min(n_users, u0 + per)) for u0 in range(0, n_users, per)]
        main = torch.cuda.current_stream(E.device)
        side = ops.aux_streams(2)
        for sdev in side:
            sdev.wait_stream(main)
        for b, (u0, u1) in enumerate(ranges):
            with torch.cuda.stream(side[b % 2]):
                run_batch(u0, u1)
        for sdev in side:
            main.wait_stream(sdev)
    # users still flagged (fewer than k unseen items, or not certifiable) are re-done by the exact-row kernel from a
    # device-side list: nothing of the pass visits the host, so consecutive passes queue up without a gap
    lst, cnt = (final_list, final_cnt) if fused_lists else ops.flag_compact(flags, 0x7fffffff)
    ops.score_exact_list(lst, cnt, factors.V, E, n_items, seen_ptr, seen_idx, topk, out_idx, out_s)
    if stats is not None:
        stats['flagged_users'] = int(cnt.item())
        stats['refolded_users'] = int(sum(int(c.item()) for c in refolded))
        stats['approx_fold_in'] = bool(approx_fold_in)
        stats['candidate_capacity'] = KC
        stats['item_splits'] = splits
        # tiles actually scored by the candidate sweep (pruning), for the roofline accounting
        n_tiles = -(-n_items // 32)
        if two_phase[0]:
            # slot 0: the head sweep (tiles [0, H)); slots 1..S: split h owns tiles H + h, H + h + S, ...
            H, S2 = two_phase
            stats['item_splits'] = 1
            stats['two_phase'] = {'head_tiles': H, 'splits': S2}
            ex = ops.score_exit_tiles(n_users, S2 + 1)
            first = H + torch.arange(S2, device=ex.device, dtype=torch.int64)[:, None]
            tail = torch.div((ex[1:] - first).clamp_min(0) + S2 - 1, S2, rounding_mode='floor')
            tail = torch.where(ex[:1] < H, torch.zeros_like(tail), tail)      # pruned inside the head: the splits did not run
            scored = ex[0].clamp_max(H) + tail.sum(dim=0)                    # tiles scored per group, all sweeps together
            chain = ex[0].clamp_max(H) + tail.max(dim=0).values              # the group's dependent chain (tile steps)
            q2 = torch.quantile(chain.double(), torch.tensor([0.5, 0.99, 1.0], dtype=torch.float64, device=ex.device))
            stats['two_phase']['chain_quantiles'] = dict(zip(('p50', 'p99', 'max'), [float(v) for v in q2.tolist()]))
        else:
            ex = ops.score_exit_tiles(n_users, splits)                      # absolute tile index, per split and group
            first = torch.arange(splits, device=ex.device, dtype=torch.int64)[:, None]   # split h owns tiles h, h+S, ...
            scored = torch.div((ex - first).clamp_min(0) + splits - 1, splits, rounding_mode='floor')
        stats['tiles_scored'] = int(scored.sum().item())
        stats['tiles_total'] = int(ex.shape[1]) * n_tiles
        q = torch.quantile(scored.flatten().double(),
                           torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=ex.device))
        stats['exit_tile_quantiles'] = dict(zip(('p50', 'p90', 'p99', 'p999', 'max'), [float(v) for v in q.tolist()]))
    if return_scores:
        return out_idx, out_s
    if out is not None:
        return ops.scatter_rows(out_idx, None, out=out)
    return out_idx


class CapturedPass:
    """One scoring pass over a FIXED (factors, test matrix, topk) captured in a hipGraph (torch.cuda.CUDAGraph) and
    replayed: the ~14 kernel launches and the temporaries of `recommend` cost one graph launch.  The pass has no host
    round trip inside (the lists of users to re-fold / re-do stay on the device), so the capture is the pass itself.
    What it is for: user sets small enough that a pass is bound by its launch sequence instead of its kernels — a
    rank's shard of ML-20M at 8 GPUs is 17K users, 0.49 ms per pass launched from Python against ~0.3 ms of kernels —
    scored repeatedly: periodic re-scoring, the timed loop of bench.py.  A new test matrix, new factors or another
    topk need a new capture (every launch's grid and arguments derive from them).
    `replay()` returns the SAME device tensor every time (the graph's output buffer): copy it before the next replay."""

    def __init__(self, ops, factors, T, topk, filter_seen=True, prune=True):
        self.ops = ops
        dev = ops.device
        T.nonneg()
        if filter_seen:
            T.seen_tiles()
        _ = T.plan
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):          # warm-up on the capture side: lazily created buffers exist afterwards
            for _ in range(2):
                recommend(ops, factors, T, topk, filter_seen, prune=prune, batches=1)
        main.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = recommend(ops, factors, T, topk, filter_seen, prune=prune, batches=1)
        # everything the captured launches point at must outlive the graph: the operands, and the per-stream scratch
        # buffers the capture created (they stay registered under the capture stream's key, which nobody else uses)
        self._keep = (factors, T, dict(ops._score_states or {}), dict(getattr(ops, '_exact_work', None) or {}))

    def replay(self):
        self.graph.replay()
        return self.out


class _CallRecorder:
    """stands in for `ops.lib` while a pass is recorded: every library call goes through unchanged and is written down"""

    def __init__(self, lib):
        self.lib = lib
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self.lib, name)
        calls = self.calls

        def call(*args):
            calls.append((name, fn, args))
            return fn(*args)
        return call


class RecordedPass:
    """One scoring pass over a FIXED (factors, test matrix, topk) as the LIST OF LIBRARY CALLS it consists of, recorded once —
    entry point and converted arguments, the stream handle among them — and issued again by `replay()`: the dozen launches of
    a pass without the Python around them (argument conversion, per-stream scratch look-ups, the temporaries' allocation:
    168 us of host time per pass through `recommend`, against ~9 launches of ~5 us).  What it is for is what `CapturedPass`
    is for — user sets whose pass is shorter on the GPU than on the host: a rank's shard of ML-20M at 8 GPUs is 17 K users,
    0.25 ms per pass on the device, and two such passes in flight are bound by the host — without the hipGraph, whose
    replay costs this runtime ~70 us per kernel node.  The pass has no host round trip inside, so its calls and their
    arguments do not depend on the data; only entries that take the stream as their first argument are replayed (the
    planning queries of a pass — splits, capacities, work sizes — are pure host functions and were answered at the
    recording).  Bound to the stream it was recorded on.  `replay()` returns the SAME device tensor every time (the pass's
    output buffer, like every temporary of the recorded pass, is kept with the recording): consume it, or order the next
    replay behind its consumer, before replaying again."""

    def __init__(self, ops, factors, T, topk, filter_seen=True, prune=True, host_out=None, hand_over='copy'):
        """host_out: a pinned host tensor [n_users x topk] int64 the lists are handed to by the recording itself; `replay()` then
        returns that tensor, valid once the stream has passed the pass.  hand_over='copy' (default): one more recorded call,
        pk_copy_to_host_async, in stream order behind the kernels; 'mapped': the pass's last kernel writes the pinned buffer
        itself (`recommend(out=...)`: no copy call at all) — measured slower for a pass in a loop (the kernel stays on the stream
        until its PCIe writes are through: 0.1225 against 0.099-0.101 ms per pass on a 17 K-user shard, and against 0.099-0.117
        for torch's `copy_` under the pass's stream; tools/probes/recorded_handover.py), the form for a single hand-over."""
        if hand_over not in ('copy', 'mapped'):
            raise ValueError("hand_over must be 'copy' or 'mapped'")
        from . import ops as ops_module
        self.ops = ops
        T.nonneg()
        if filter_seen:
            T.seen_tiles()
        _ = T.plan
        for _ in range(2):                     # lazily created buffers (per-stream scratch, cached images) exist afterwards
            recommend(ops, factors, T, topk, filter_seen, prune=prune, batches=1)
        self.stream = ops.stream_key()
        recorder = _CallRecorder(ops.lib)
        keep = []
        with ops.pass_lock:
            ops_module._PTR_KEEP, ops.lib = keep, recorder
            try:
                self.out = recommend(ops, factors, T, topk, filter_seen, prune=prune, batches=1, out=host_out if hand_over == 'mapped' else None)
            finally:
                ops_module._PTR_KEEP, ops.lib = None, recorder.lib
        import ctypes
        self.calls = []
        for name, fn, args in recorder.calls:
            first = args[0] if args else None
            handle = (first.value or 0) if isinstance(first, ctypes.c_void_p) else None
            if handle is not None and handle == (self.stream or 0) and fn.restype is ctypes.c_int and name != 'pk_ctx_create':
                self.calls.append((name, fn, args))
        if not self.calls:
            raise RuntimeError('RecordedPass: the pass made no library call on its stream')
        self.host_out = host_out
        if host_out is not None and hand_over != 'mapped':
            out = self.out
            if (host_out.is_cuda or not host_out.is_pinned() or host_out.shape != out.shape or host_out.dtype != out.dtype
                    or not host_out.is_contiguous() or not out.is_contiguous()):
                raise ValueError('RecordedPass: host_out must be a contiguous pinned host tensor of the output\'s shape and dtype')
            self.calls.append(('pk_copy_to_host_async', ops.lib.pk_copy_to_host_async,
                               (ctypes.c_void_p(self.stream), ctypes.c_void_p(host_out.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                out.numel() * out.element_size())))
            self.out = host_out
        # everything the recorded arguments point at outlives the recording: operands, temporaries, per-stream scratch
        self._keep = (factors, T, keep, dict(ops._score_states or {}), dict(getattr(ops, '_exact_work', None) or {}))

    def replay(self):
        for name, fn, args in self.calls:
            rc = fn(*args)
            if rc:
                _lib.check(rc, name)
        return self.out


def dense_scores(ops, factors, T, start, stop):
    """Dense fp64 scores of test users [start, stop) — kept for `slice_recommendations` /
    `_user_scores` (models.py:277-291, 857-861); not used by get_recommendations."""
    E = ops.spmm(T, factors.V)
    return ops.dense_scores(factors.V, E[start:stop].contiguous())
