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
import numpy as np
import torch

from .csr import coo_to_csr

EXACT_ROWS_BYTES = 2 << 30  # work-buffer budget per launch of the exact path


class FactorImage:
    """Item factors resident in HBM in both forms the scoring kernels read: fp64 row-major
    (re-scoring, fold-in) and fp32 MFMA-fragment packed (candidate pass)."""

    def __init__(self, ops, V):
        self.ops = ops
        self.V = V.contiguous()
        self.n_items, self.K = self.V.shape
        self.Vp = ops.pack_frag(self.V)
        self.vmax = float(torch.linalg.vector_norm(self.V, dim=1).max().item())
        self.tile_bound = ops.tile_norm_bound(self.V)   # exact pruning bound of the candidate sweep


def test_csr_from_triplet(test_data, shape, weights=None):
    """(user_idx, item_idx, feedback) sorted by user -> canonical CSR arrays keeping zeros.
    `weights`, if given, replaces the feedback values (CoFFee's per-entry coefficient)."""
    users, items, fdbk = test_data
    vals = np.asarray(fdbk if weights is None else weights, dtype=np.float64)
    return coo_to_csr(users, items, vals, shape, sum_duplicates=True)


def recommend(ops, factors, T, topk, filter_seen=True, return_scores=False, stats=None, prune=True):
    """factors: FactorImage; T: ops-level CSR of the test users [n_users x n_items].
    Returns int64 device tensor [n_users x topk] (+ fp64 scores), rows in test-user order,
    columns by descending score — the contract of models.py:400-405."""
    n_users, n_items = T.shape
    if n_items != factors.n_items:
        raise ValueError('test matrix and item factors disagree on the number of items')
    if topk > n_items:
        raise ValueError('kth(=%d) out of bounds (%d)' % (n_items - topk, n_items))  # numpy argpartition's error
    KC = ops.candidate_capacity(topk)
    K = factors.K
    E = ops.spmm(T, factors.V)                       # fold-in, fp64 (K4)
    if KC == 0:
        # topk beyond the fused kernel's 52: every user goes through the exact fp64 row kernel
        # (all items scored, two-class key) — slow but the same contract
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
    Ep = ops.pack_frag(E)
    seen_ptr = T.indptr if filter_seen else None
    seen_idx = T.indices if filter_seen else None
    splits = ops.score_splits(n_users, KC, prune)    # item ranges per user group (1 when pruning / users fill the chip)
    # exact Cauchy-Schwarz pruning: a group of 32 users leaves the sweep once no later item can beat
    # any of its thresholds (`prune=False` forces the full sweep: same result, tuning / tests only)
    ub = ops.row_norm_bound(E) if prune else None
    cs, ci = ops.score_candidates(factors.Vp, Ep, n_users, n_items, K, seen_ptr, seen_idx, KC, splits,
                                  user_bound=ub, tile_bound=factors.tile_bound if prune else None)   # K3
    out_idx, out_s, flags = ops.rescore_topk(factors.V, E, n_items, seen_ptr, KC, cs, ci, topk,
                                             factors.vmax, want_scores=True, splits=splits)
    rows = torch.nonzero(flags, as_tuple=False).flatten().to(torch.int32)
    n_flag = int(rows.numel())
    if stats is not None:
        stats['flagged_users'] = n_flag
        stats['candidate_capacity'] = KC
        stats['item_splits'] = splits
        # tiles actually scored by the candidate sweep (pruning), for the roofline accounting
        n_tiles = -(-n_items // 32)
        split_tiles = -(-n_tiles // splits)
        ex = ops.score_exit_tiles(n_users, splits)
        lo = torch.arange(splits, device=ex.device, dtype=torch.int64)[:, None] * split_tiles
        stats['tiles_scored'] = int((ex - lo).clamp_min(0).sum().item())
        stats['tiles_total'] = int(ex.shape[1]) * n_tiles
    if n_flag:
        per = max(1, int(EXACT_ROWS_BYTES // (n_items * 9 + 16)))
        for s in range(0, n_flag, per):
            sub = rows[s:s + per].contiguous()
            ex_idx, ex_s = ops.score_exact_rows(sub, factors.V, E, n_items, seen_ptr, seen_idx, topk)
            out_idx[sub.long()] = ex_idx
            out_s[sub.long()] = ex_s
    if return_scores:
        return out_idx, out_s
    return out_idx


def dense_scores(ops, factors, T, start, stop):
    """Dense fp64 scores of test users [start, stop) — kept for `slice_recommendations` /
    `_user_scores` (models.py:277-291, 857-861); not used by get_recommendations."""
    E = ops.spmm(T, factors.V)
    return ops.dense_scores(factors.V, E[start:stop].contiguous())
