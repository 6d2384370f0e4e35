"""CPU oracle: a NumPy/SciPy restatement of the evfro/polara PureSVD + CoFFee hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `polara_amd/` may import this module; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do (and, in the same two roles, the
side benchmarks `tools/bench_coffee.py` / `tools/bench_s50m_shard.py` that `bench.py`'s contract has no
room for), and only as the checker / the timed CPU baseline, never as the thing shipped
(`tests/test_abi.py::test_no_cpu_fallback_in_package` greps the package for it).

Every function cites the reference file:line (relative to /root/reference) it restates.  The
arithmetic that lives outside the reference tree is SciPy's `scipy.sparse.linalg.svds` (ARPACK,
unpinned in the reference's conda_req.txt:11, scipy 1.15.3 here), NumPy/LAPACK `qr`/`svd`,
NumPy `argpartition`/`argsort` and BLAS `dot`; the oracle calls the *same* third-party functions
with the *same* arguments as the reference does.

Pinning: the reference's own tests hold no vectors for this path (tests/preprocessing_test.py:5-11
is its only test).  The oracle is therefore pinned against outputs of the reference itself run in
the build container (imported from /root/reference through the test-only numba shim) — see
`tests/golden/make_golden.py`, which asserts bit-equality oracle == reference on every fixture
before writing it, and `tests/test_oracle_golden.py`, which re-checks the oracle against the
committed fixtures on any machine.
"""
import numpy as np
import scipy as sp
import scipy.sparse
from scipy.sparse import coo_matrix, csr_matrix
from scipy.sparse.linalg import svds


# ----------------------------------------------------------------------------------------------
# chunk sizing: polara/recommender/utils.py:7-53, polara/recommender/defaults.py:51
# ----------------------------------------------------------------------------------------------
MEMORY_HARD_LIMIT = 1  # defaults.py:51 ("in gigabytes")


def range_division(length, fit_size):
    """utils.py:7-13."""
    n_chunks = length // fit_size + int((length % fit_size) > 0)
    chunk_size, remainder = divmod(length, n_chunks)
    chunk_sizes = ([0] + remainder * [chunk_size + 1] + (n_chunks - remainder) * [chunk_size])
    return np.cumsum(chunk_sizes)


def get_chunk_size(shp, result_width, scores_multiplier, dtypes=None,
                   available_memory=None, memory_hard_limit=MEMORY_HARD_LIMIT):
    """utils.py:16-47.  `available_memory` replaces systools.get_available_memory()
    (tools/systools.py:52-58, psutil bytes); it only matters when below the hard limit."""
    chunk_size = shp[0]
    shp = [s / 1024 if i < 2 else s for i, s in enumerate(shp)]
    if dtypes:
        result_itemsize = np.dtype(dtypes[0]).itemsize / 1024
        scores_itemsize = np.dtype(dtypes[1]).itemsize / 1024
    else:
        result_itemsize = np.dtype(np.int64).itemsize / 1024
        scores_itemsize = np.dtype(np.float64).itemsize / 1024
    result_memory = shp[0] * (result_width / 1024) * result_itemsize
    scores_memory = np.prod(shp[:2]) * scores_multiplier * scores_itemsize
    if available_memory is None:
        import psutil
        available_memory = psutil.virtual_memory().available
    memory_limit = 0.8 * available_memory
    if memory_hard_limit:
        memory_limit = min(memory_limit, memory_hard_limit)
    required_memory = scores_memory + result_memory
    if required_memory > memory_limit:
        chunk_size = min(int((memory_limit - result_memory) /
                             (shp[1] * scores_itemsize * (scores_multiplier / 1024) +
                              result_itemsize / (1024 ** 2)) - 1),
                         chunk_size)
        if chunk_size <= 0:
            raise MemoryError()
    return chunk_size


def array_split(shp, result_width, scores_multiplier, dtypes=None, **kw):
    """utils.py:50-53."""
    chunk_size = get_chunk_size(shp, result_width, scores_multiplier, dtypes=dtypes, **kw)
    return range_division(shp[0], chunk_size)


# ----------------------------------------------------------------------------------------------
# matrices: polara/recommender/models.py:160-211, 260-270
# ----------------------------------------------------------------------------------------------
def get_training_matrix(idx, val, shp, dtype=None, ignore_feedback=False):
    """models.py:160-177 (csr branch).  idx int[nnz,2], val float[nnz]."""
    dtype = dtype or val.dtype
    if ignore_feedback:
        val = np.ones_like(val, dtype=dtype)
    matrix = coo_matrix((val, (idx[:, 0], idx[:, 1])), shape=shp, dtype=dtype)
    return matrix.tocsr()


def slice_test_data(test_data, start, stop):
    """models.py:260-270."""
    user_coo, item_coo, fdbk_coo = test_data
    slicer = (user_coo >= start) & (user_coo < stop)
    return (user_coo[slicer] - start, item_coo[slicer], fdbk_coo[slicer])


def get_test_matrix(test_data, shape, user_slice=None, dtype=None, ignore_feedback=False):
    """models.py:180-211."""
    num_users_all = shape[0]
    if user_slice:
        start, stop = user_slice
        stop = min(stop, num_users_all)
        num_users = stop - start
        coo_data = slice_test_data(test_data, start, stop)
    else:
        num_users = num_users_all
        coo_data = test_data
    user_coo, item_coo, fdbk_coo = coo_data
    valid_fdbk = fdbk_coo != 0
    if not valid_fdbk.all():
        user_coo = user_coo[valid_fdbk]
        item_coo = item_coo[valid_fdbk]
        fdbk_coo = fdbk_coo[valid_fdbk]
    dtype = dtype or fdbk_coo.dtype
    if ignore_feedback:
        fdbk_coo = np.ones_like(fdbk_coo, dtype=dtype)
    num_items = shape[1]
    test_matrix = csr_matrix((fdbk_coo, (user_coo, item_coo)), shape=(num_users, num_items), dtype=dtype)
    return test_matrix, coo_data


def rebase_test_users(user_idx, n_test_users):
    """models.py:244-255: contiguous re-basing of (sorted) test user ids."""
    idx_diff = np.diff(user_idx)
    assert (idx_diff >= 0).all()
    if (idx_diff > 1).any() or (user_idx.min() != 0):
        test_users = user_idx[np.r_[0, np.where(idx_diff)[0] + 1]]
        user_idx = np.r_[0, np.cumsum(idx_diff > 0)].astype(user_idx.dtype)
    else:
        test_users = np.arange(n_test_users)
    return user_idx, test_users


# ----------------------------------------------------------------------------------------------
# PureSVD: polara/recommender/models.py:835-861
# ----------------------------------------------------------------------------------------------
def svd_build(svd_matrix, rank, return_factors='vh'):
    """models.py:835-855.  Returns (user_factors|None, sigma desc, item_factors [n_items x rank],
    F-ordered exactly like the reference's `ascontiguousarray(vh[::-1]).T`)."""
    user_factors, sigma, item_factors = svds(svd_matrix, k=rank, return_singular_vectors=return_factors)
    if user_factors is not None:
        user_factors = np.ascontiguousarray(user_factors[:, ::-1])
    if item_factors is not None:
        item_factors = np.ascontiguousarray(item_factors[::-1, :]).T
    if sigma is not None:
        sigma = np.ascontiguousarray(sigma[::-1])
    return user_factors, sigma, item_factors


def rescale_matrix(matrix, scaling, axis, binary=True):
    """preprocessing/matrices.py:71-93 (scaling == 1 still multiplies by ones, like the reference)."""
    from scipy.sparse import diags
    from scipy.sparse.linalg import norm as spnorm
    if binary:
        norm = np.sqrt(matrix.getnnz(axis=axis))
    else:
        norm = spnorm(matrix, axis=axis, ord=2)
    scaling_values = np.power(norm, scaling - 1, where=norm != 0)
    scaling_matrix = diags(scaling_values)
    if axis == 0:
        return matrix.dot(scaling_matrix)
    return scaling_matrix.dot(matrix)


def scaled_training_matrix(idx, val, shp, col_scaling=0.4, row_scaling=1, dtype=np.float64):
    """ScaledMatrixMixin.get_training_matrix, models.py:891-895."""
    m = get_training_matrix(idx, val, shp, dtype=dtype)
    m = rescale_matrix(m, row_scaling, 1)
    m = rescale_matrix(m, col_scaling, 0)
    return m


def svd_slice_recommendations(v, test_data, shape, start, stop):
    """models.py:857-861."""
    test_matrix, slice_data = get_test_matrix(test_data, shape, (start, stop))
    scores = (test_matrix.dot(v)).dot(v.T)
    return scores, slice_data


# ----------------------------------------------------------------------------------------------
# seen-item down-voting and top-k: polara/recommender/models.py:488-564
# ----------------------------------------------------------------------------------------------
def topsort(a, topk):
    """models.py:488-491."""
    parted = np.argpartition(a, -topk)[-topk:]
    return parted[np.argsort(-a[parted])]


def downvote_seen_items(recs, idx_seen):
    """models.py:494-519, dense branch (in-place on `recs`)."""
    idx_seen = idx_seen[:2]
    try:
        idx_seen_flat = np.ravel_multi_index(idx_seen, recs.shape)
    except ValueError:
        idx_seen_flat = idx_seen
    seen_data = recs.flat[idx_seen_flat]
    lowered = recs.min() - (seen_data.max() - seen_data) - 1
    recs.flat[idx_seen_flat] = lowered


def get_topk_elements(scores, topk):
    """models.py:561-563, dense branch."""
    return np.apply_along_axis(topsort, 1, scores, topk)


def get_recommendations(slice_fn, test_data, test_shape, topk, filter_seen=True,
                        scores_multiplier=1, chunk_kw=None, return_scores=False):
    """models.py:359-405 (sequential path): chunk -> slice_recommendations -> downvote -> top-k.
    `slice_fn(test_data, shape, start, stop) -> (scores, slice_data)`."""
    slices_idx = array_split(test_shape, topk, scores_multiplier, **(chunk_kw or {}))
    top_recs = np.empty((test_shape[0], topk), dtype=np.int64)
    top_scores = np.empty((test_shape[0], topk), dtype=np.float64) if return_scores else None
    for start, stop in zip(slices_idx[:-1], slices_idx[1:]):
        scores, slice_data = slice_fn(test_data, test_shape, start, stop)
        if filter_seen:
            downvote_seen_items(scores, slice_data)
        recs = get_topk_elements(scores, topk)
        top_recs[start:stop, :] = recs
        if return_scores:
            top_scores[start:stop, :] = np.take_along_axis(scores, recs, axis=1)
    if return_scores:
        return top_recs, top_scores
    return top_recs


def svd_recommendations(v, test_data, test_shape, topk, filter_seen=True, **kw):
    return get_recommendations(lambda td, sh, a, b: svd_slice_recommendations(v, td, sh, a, b),
                               test_data, test_shape, topk, filter_seen, **kw)


# ----------------------------------------------------------------------------------------------
# sparse tensor kernels: polara/lib/sparse.py:172-216 ; polara/lib/tensor.py:7-96
# ----------------------------------------------------------------------------------------------
def inverse_permutation(p):
    """sparse.py:172-175."""
    s = np.empty(p.size, p.dtype)
    s[p] = np.arange(p.size)
    return s


def dttm_seq_loops(idx, val, u, v, mode0, mode1, mode2, res):
    """sparse.py:203-216 verbatim loop nest (slow; small cases only)."""
    new_shape1 = u.shape[1]
    new_shape2 = v.shape[1]
    for i in range(len(val)):
        i0 = idx[i, mode0]
        i1 = idx[i, mode1]
        i2 = idx[i, mode2]
        vv = val[i]
        for j in range(new_shape1):
            uij = u[i1, j]
            for k in range(new_shape2):
                vik = v[i2, k]
                res[i0, j, k] += vv * uij * vik


def dttm_seq(idx, val, u, v, mode0, mode1, mode2, res):
    """sparse.py:203-216, vectorised: identical summation ORDER per output element
    (np.add.at applies updates in nnz order, one product `vv*uij*vik` evaluated left-to-right)."""
    # blocks of nnz bound the [block x r1 x r2] temporary; the order of updates per output element is unchanged
    block = max(1, (64 << 20) // max(1, 8 * u.shape[1] * v.shape[1]))
    for a in range(0, len(val), block):
        b = min(len(val), a + block)
        contrib = (val[a:b, None] * u[idx[a:b, mode1], :])[:, :, None] * v[idx[a:b, mode2], :][:, None, :]
        np.add.at(res, idx[a:b, mode0], contrib)


def ttm3d_seq(idx, val, shape, U, V, modes, dtype=None, loops=False):
    """tensor.py:7-19."""
    mode1, mat_mode1 = modes[0]
    mode2, mat_mode2 = modes[1]
    u = U.T if mat_mode1 == 1 else U
    v = V.T if mat_mode2 == 1 else V
    mode0, = [x for x in (0, 1, 2) if x not in (mode1, mode2)]
    new_shape = (shape[mode0], U.shape[1 - mat_mode1], V.shape[1 - mat_mode2])
    res = np.zeros(new_shape, dtype=dtype)
    (dttm_seq_loops if loops else dttm_seq)(idx, val, u, v, mode0, mode1, mode2, res)
    return res


def hooi(idx, val, shape, core_shape, return_core=True, num_iters=25,
         growth_tol=0.01, seed=None, trace=None):
    """tensor.py:37-96 with parallel_ttm=False (defaults.py:29).  `trace`, if a list, collects the
    per-iteration core norms."""
    random_state = np.random if seed is None else np.random.RandomState(seed)
    r0, r1, r2 = core_shape
    u1 = random_state.rand(shape[1], r1)
    u1 = np.linalg.qr(u1, mode='reduced')[0]
    u2 = random_state.rand(shape[2], r2)
    u2 = np.linalg.qr(u2, mode='reduced')[0]

    g_norm_old = 0
    return_core_vectors = True if return_core else 'u'
    for i in range(num_iters):
        u0 = ttm3d_seq(idx, val, shape, u2, u1, ((2, 0), (1, 0))).reshape(shape[0], r1 * r2)
        uu, ss, _ = svds(u0, k=r0, return_singular_vectors='u')
        u0 = np.ascontiguousarray(uu[:, ::-1])

        u1 = ttm3d_seq(idx, val, shape, u2, u0, ((2, 0), (0, 0))).reshape(shape[1], r0 * r2)
        uu, ss, _ = svds(u1, k=r1, return_singular_vectors='u')
        u1 = np.ascontiguousarray(uu[:, ::-1])

        u2 = ttm3d_seq(idx, val, shape, u1, u0, ((1, 0), (0, 0))).reshape(shape[2], r0 * r1)
        uu, ss, vv = svds(u2, k=r2, return_singular_vectors=return_core_vectors)
        u2 = np.ascontiguousarray(uu[:, ::-1])

        g_norm_new = np.linalg.norm(ss)
        g_growth = (g_norm_new - g_norm_old) / g_norm_new
        g_norm_old = g_norm_new
        if trace is not None:
            trace.append(float(g_norm_new))
        if g_growth < growth_tol:
            break

    if return_core:
        g = np.ascontiguousarray((ss[:, np.newaxis] * vv)[::-1, :])
        g = g.reshape(r2, r1, r0).transpose(2, 1, 0)
    else:
        g = None
    return u0, u1, u2, g


# ----------------------------------------------------------------------------------------------
# CoFFee scoring: polara/recommender/models.py:983-1054 ; polara/lib/sparse.py:190-200
# ----------------------------------------------------------------------------------------------
def tensor_outer_at(val, v, w, i, j):
    """sparse.py:190-200 (gufunc '(),(i,m),(j,n),(),()->(m,n)'): res[n,:,:] = val*v[i_n,:,None]*w[j_n,None,:]."""
    return (val * v[i, :])[:, :, None] * w[j, :][:, None, :]


def flatten_scores(tensor_scores, flattener=None):
    """models.py:983-1006."""
    flattener = flattener or slice(None)
    if isinstance(flattener, str):
        slicer = slice(None)
        flatten = getattr(np, flattener)
        matrix_scores = flatten(tensor_scores[..., slicer], axis=-1)
    elif isinstance(flattener, int):
        slicer = flattener
        matrix_scores = tensor_scores[..., slicer]
    elif isinstance(flattener, (list, slice)):
        slicer = flattener
        flatten = np.sum
        matrix_scores = flatten(tensor_scores[..., slicer], axis=-1)
    elif isinstance(flattener, tuple):
        slicer, flatten_method = flattener
        slicer = slicer or slice(None)
        flatten = getattr(np, flatten_method)
        matrix_scores = flatten(tensor_scores[..., slicer], axis=-1)
    elif callable(flattener):
        matrix_scores = flattener(tensor_scores)
    else:
        raise ValueError('Unrecognized value for flattener attribute')
    return matrix_scores


def coffee_slice_recommendations(v, w, test_data, shape, start, stop, flattener=slice(0, None)):
    """models.py:1042-1054."""
    slice_idx = slice_test_data(test_data, start, stop)
    scores = tensor_outer_at(1.0, v, w, slice_idx[1], slice_idx[2])
    scores = np.add.reduceat(scores, np.r_[0, np.where(np.diff(slice_idx[0]))[0] + 1])
    wt_flat = flatten_scores(w.T, flattener)
    scores = np.tensordot(scores, wt_flat, axes=(2, 0)).dot(v.T)
    return scores, slice_idx


def coffee_recommendations(v, w, test_data, test_shape, topk, filter_seen=True,
                           flattener=slice(0, None), **kw):
    """models.py:214-224 sets scores_multiplier = r2 for tensor models."""
    return get_recommendations(
        lambda td, sh, a, b: coffee_slice_recommendations(v, w, td, sh, a, b, flattener),
        test_data, test_shape, topk, filter_seen, scores_multiplier=w.shape[1], **kw)


# ----------------------------------------------------------------------------------------------
# checker helpers (not restatements): tie flags and set comparison
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------
# CoffeeModel extras: unfolded test slices, holdout slices, feedback prediction
# polara/lib/sparse.py:178-187, polara/recommender/models.py:1027-1039, 1056-1092
# ----------------------------------------------------------------------------------------------
def unfold_tensor_coordinates(index, shape, mode):
    """lib/sparse.py:178-187: coordinates of the mode-`mode` unfolding with `mode` as the COLUMN index and the two
    other modes (in their order) flattened C-style into the row index."""
    modes = [m for m in (0, 1, 2) if m != mode] + [mode]
    mode_shape = tuple(shape[m] for m in modes)
    mode_index = tuple(index[m] for m in modes)
    flat_index = np.ravel_multi_index(mode_index, mode_shape)
    unfold_shape = (mode_shape[0] * mode_shape[1], mode_shape[2])
    return np.unravel_index(flat_index, unfold_shape), unfold_shape


def unfold_test_tensor_slice(test_data, shape, start, stop, mode):
    """models.py:1027-1039: the binary test tensor of users [start, stop) unfolded along `mode` (uint8 CSR)."""
    slice_idx = slice_test_data(test_data, start, stop)
    slice_shp = (stop - start, shape[1], shape[2])
    idx, shp = unfold_tensor_coordinates(slice_idx, slice_shp, mode)
    val = np.ones_like(slice_idx[2], dtype=np.uint8)
    return csr_matrix((val, idx), shape=shp, dtype=val.dtype), slice_idx


def get_holdout_slice(holdout_users, holdout_items, start, stop):
    """models.py:1056-1065: the holdout entries of users [start, stop), user ids re-based to the slice."""
    holdout_users = np.asarray(holdout_users)
    sel = (holdout_users >= start) & (holdout_users < stop)
    return holdout_users[sel].astype(np.int64) - start, np.asarray(holdout_items)[sel].astype(np.int64)


def coffee_predict_feedback(u, v, w, g, holdout_users, holdout_items):
    """models.py:1068-1091: for every holdout (user, item) the feedback level with the largest reconstructed score,
    scores[h, f] = sum_abc g[a, b, c] u[user_h, a] v[item_h, b] w[f, c]; returns the level INDICES (the reference maps
    them to feedback values through data.index.feedback)."""
    holdout_users = np.asarray(holdout_users, dtype=np.int64)
    holdout_items = np.asarray(holdout_items, dtype=np.int64)
    gv = np.tensordot(g, v[holdout_items, :], (1, 1))
    gu = (gv * u[holdout_users, None, :].T).sum(axis=0)
    scores = w.dot(gu).T
    return np.argmax(scores, axis=-1), scores


def boundary_gap(scores, topk):
    """Per-row gap between the k-th and (k+1)-th largest value of a dense (already down-voted)
    score block.  Rows with gap == 0 have an implementation-defined reference top-k (introselect)."""
    part = -np.partition(-scores, topk, axis=1)[:, :topk + 1]
    part.sort(axis=1)
    return part[:, 1] - part[:, 0]


def topk_sets_equal(a, b):
    a = np.sort(np.asarray(a), axis=1)
    b = np.sort(np.asarray(b), axis=1)
    return (a == b).all(axis=1)
