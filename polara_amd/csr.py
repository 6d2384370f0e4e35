"""Host-side sparse containers: canonical CSR/CSC construction and the per-row task plans the
SpMM / TTM kernels consume (include/polara_hip.h, pk_spmm_csr_f64).

The COO->CSR step restates `RecommenderModel.get_training_matrix` (models.py:160-177:
`coo_matrix(...).tocsr()`, duplicates summed, indices sorted) in NumPy; it is index bookkeeping, not
arithmetic on the hot path, and runs once per build.
"""
import numpy as np

SPLIT_NNZ = 1024  # rows longer than this are cut into several wave tasks


def coo_to_csr(rows, cols, vals, shape, sum_duplicates=True):
    """Canonical CSR (row-sorted, column-sorted, duplicates summed).  Returns indptr int64,
    indices int32, values (dtype of vals)."""
    n_rows, n_cols = int(shape[0]), int(shape[1])
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals)
    if rows.size:
        if rows.min() < 0 or rows.max() >= n_rows or cols.min() < 0 or cols.max() >= n_cols:
            raise ValueError('index out of bounds')
    key = rows * n_cols + cols
    order = np.argsort(key, kind='stable')
    key = key[order]
    vals = vals[order]
    if sum_duplicates and key.size:
        first = np.r_[True, key[1:] != key[:-1]]
        if not first.all():
            starts = np.flatnonzero(first)
            vals = np.add.reduceat(vals, starts)
            key = key[starts]
    r = key // n_cols
    c = (key - r * n_cols).astype(np.int32)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    np.cumsum(indptr, out=indptr)
    return indptr, c, np.ascontiguousarray(vals)


def csr_transpose(indptr, indices, values, n_cols):
    """CSR of A^T (= CSC of A), canonical."""
    n_rows = len(indptr) - 1
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(indptr))
    order = np.argsort(indices, kind='stable')  # stable: rows stay ascending within a column
    t_indices = rows[order].astype(np.int32)
    t_values = values[order]
    t_indptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.add.at(t_indptr, indices.astype(np.int64) + 1, 1)
    np.cumsum(t_indptr, out=t_indptr)
    return t_indptr, t_indices, np.ascontiguousarray(t_values)


def build_row_tasks(indptr, split=SPLIT_NNZ):
    """Task plan for one-wave-per-task row kernels.

    Every row gets at least one task (empty rows too, so outputs are fully defined).  Rows with
    more than `split` nnz are cut into ceil(nnz/split) near-equal tasks writing partial results
    to consecutive slots, summed in slot order by the fix-up kernel.
    Returns dict of NumPy arrays: task_row i32, task_begin i64, task_end i64, task_slot i32,
    long_row i32, long_slot_begin i32, long_slot_end i32, n_slots int.
    """
    indptr = np.asarray(indptr, dtype=np.int64)
    n_rows = len(indptr) - 1
    counts = np.diff(indptr)
    n_chunks = np.maximum(1, -(-counts // split))
    n_tasks = int(n_chunks.sum())
    task_row = np.repeat(np.arange(n_rows, dtype=np.int64), n_chunks)
    first_task = np.cumsum(n_chunks) - n_chunks
    k = np.arange(n_tasks, dtype=np.int64) - np.repeat(first_task, n_chunks)
    chunk_len = -(-counts // n_chunks)  # ceil(count / n_chunks)
    begin = indptr[task_row] + k * chunk_len[task_row]
    end = np.minimum(begin + chunk_len[task_row], indptr[task_row + 1])
    begin = np.minimum(begin, end)
    is_long = n_chunks > 1
    task_slot = np.full(n_tasks, -1, dtype=np.int32)
    long_rows = np.flatnonzero(is_long)
    long_chunks = n_chunks[long_rows]
    slot_end = np.cumsum(long_chunks)
    slot_begin = slot_end - long_chunks
    if long_rows.size:
        long_task_mask = is_long[task_row]
        task_slot[long_task_mask] = np.arange(int(long_chunks.sum()), dtype=np.int32)
    # host-side index of the plan by row (tasks and long rows are in row order): lets a caller run a
    # contiguous row range of the matrix as its own launch (user batches on separate streams)
    row_first_task = np.concatenate(([0], np.cumsum(n_chunks))).astype(np.int64)
    return dict(row_first_task=row_first_task,
                task_row=task_row.astype(np.int32), task_begin=begin, task_end=end, task_slot=task_slot,
                long_row=long_rows.astype(np.int32), long_slot_begin=slot_begin.astype(np.int32),
                long_slot_end=slot_end.astype(np.int32), n_slots=int(long_chunks.sum()) if long_rows.size else 0)


def nnz_balanced_row_partition(indptr, parts):
    """Contiguous row ranges with ~equal nnz.  Returns int64[parts+1] boundaries."""
    indptr = np.asarray(indptr, dtype=np.int64)
    n_rows = len(indptr) - 1
    nnz = indptr[-1]
    targets = (np.arange(1, parts, dtype=np.float64) * nnz / parts)
    cuts = np.searchsorted(indptr, targets, side='left')
    bounds = np.r_[0, cuts, n_rows].astype(np.int64)
    return np.maximum.accumulate(bounds)


def popularity_order(cols, n_items, counts=None):
    """Internal item order of the device path: items by DESCENDING interaction count (ties by id).

    Why: the fused scoring kernel sweeps the catalogue once per user group keeping a running top-k
    threshold; visiting popular items first raises the threshold to (nearly) its final value within
    the first few percent of the sweep, and the users' seen items — mostly popular ones — cluster in
    the first tiles, so the candidate-push and seen-mask paths go quiet for the rest of the sweep.
    Pure relabelling: factors and recommendations are mapped back to external ids by the model layer.
    `counts`: the per-item interaction counts when the caller already has them (a user-sharded dataset sums
    them over ranks first, so that every rank derives the same order); `cols` is ignored then.
    Returns (rank int32[n_items]: external id -> internal position, inv int32[n_items]: internal -> external).
    """
    if counts is None:
        counts = np.bincount(np.asarray(cols, dtype=np.int64), minlength=n_items)
    inv = np.argsort(-counts, kind='stable').astype(np.int32)
    rank = np.empty(n_items, dtype=np.int32)
    rank[inv] = np.arange(n_items, dtype=np.int32)
    return rank, inv
