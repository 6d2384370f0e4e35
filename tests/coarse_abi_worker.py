"""ctypes-only driver of the coarse C entry points (include/polara_hip.h: pk_ctx_* / pk_mat_* / pk_svd_build /
pk_score_topk): NO torch, no polara_amd import — what a host in another language would do.  Checks the factors
against a dense SVD / the golden fixture and the lists against the oracle (the reference path restated)."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert 'torch' not in sys.modules
from oracle import polara_oracle as orc      # checker only (NumPy / SciPy)

lib = C.CDLL(os.path.join(ROOT, 'polara_amd', 'libpolarahip.so'))
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double


class Stats(C.Structure):
    _fields_ = [('outer', i32), ('gramian_steps', i32), ('block', i32), ('converged', i32), ('final_rel_residual', f64)]


lib.pk_ctx_create.argtypes, lib.pk_ctx_create.restype = [i32, C.POINTER(vp)], C.c_int
lib.pk_ctx_destroy.argtypes, lib.pk_ctx_destroy.restype = [vp], None
lib.pk_ctx_error.argtypes, lib.pk_ctx_error.restype = [vp], C.c_char_p
lib.pk_ctx_set_option.argtypes, lib.pk_ctx_set_option.restype = [vp, C.c_char_p, i32], C.c_int
lib.pk_mat_from_csr.argtypes, lib.pk_mat_from_csr.restype = [vp, i64, i64, i64, vp, vp, vp, i32, C.POINTER(vp)], C.c_int
lib.pk_mat_from_coo.argtypes, lib.pk_mat_from_coo.restype = [vp, i64, i64, i64, vp, vp, i64, vp, i32, C.POINTER(vp)], C.c_int
lib.pk_mat_free.argtypes, lib.pk_mat_free.restype = [vp, vp], None
lib.pk_mat_nnz.argtypes, lib.pk_mat_nnz.restype = [vp], i64
lib.pk_svd_build.argtypes = [vp, vp, i32, i32, f64, i32, C.c_uint64, vp, vp, vp, C.POINTER(Stats)]
lib.pk_svd_build.restype = C.c_int
lib.pk_score_topk.argtypes, lib.pk_score_topk.restype = [vp, i64, i32, vp, vp, i32, i32, vp, vp], C.c_int
lib.pk_serving_create.argtypes, lib.pk_serving_create.restype = [vp, i64, i32, vp, vp, C.POINTER(vp)], C.c_int
lib.pk_serving_score.argtypes, lib.pk_serving_score.restype = [vp, vp, i32, i32, vp, vp], C.c_int
lib.pk_serving_free.argtypes, lib.pk_serving_free.restype = [vp, vp], None
lib.pk_hooi.argtypes = [vp, i64, vp, vp, vp, vp, i32, f64, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, vp]
lib.pk_hooi.restype = C.c_int


def ptr(a):
    return a.ctypes.data_as(vp)


def check(ctx, rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, lib.pk_ctx_error(ctx).decode()))


def planted(n_users, n_items, mean, rank, seed):
    rng = np.random.RandomState(seed)
    P, Q = rng.randn(n_users, rank), rng.randn(n_items, rank)
    pop = 0.8 * np.log(1.0 / (rng.permutation(n_items) + 1.0))
    rows, cols, vals = [], [], []
    for u in range(n_users):
        k = int(np.clip(rng.lognormal(np.log(mean), 0.6), 5, n_items // 2))
        s = P[u] @ Q.T + pop + rng.gumbel(size=n_items)
        it = np.sort(np.argpartition(-s, k)[:k])
        rows.append(np.full(k, u)); cols.append(it); vals.append(rng.randint(1, 6, k).astype(np.float64))
    return np.concatenate(rows).astype(np.int64), np.concatenate(cols).astype(np.int64), np.concatenate(vals)


def main():
    ctx = vp()
    rc = lib.pk_ctx_create(0, C.byref(ctx))
    assert rc == 0, rc
    n_users, n_items, rank, topk = 3000, 700, 12, 10
    r, c, v = planted(n_users, n_items, 40, 8, 5)
    # COO route: the interleaved [nnz x 2] index array, with duplicates (summed)
    idx = np.ascontiguousarray(np.stack([np.r_[r, r[:50]], np.r_[c, c[:50]]], axis=1))
    vv = np.r_[v, v[:50]].astype(np.float32)
    A = vp()
    check(ctx, lib.pk_mat_from_coo(ctx, n_users, n_items, len(vv), ptr(idx), C.c_void_p(idx.ctypes.data + 8), 2, ptr(vv), 0, C.byref(A)),
          'pk_mat_from_coo')
    S = sps.coo_matrix((vv.astype(np.float64), (idx[:, 0], idx[:, 1])), shape=(n_users, n_items)).tocsr()
    assert lib.pk_mat_nnz(A) == S.nnz
    sigma = np.empty(rank); V = np.empty((n_items, rank), order='F'); U = np.empty((n_users, rank), order='F')
    st = Stats()
    check(ctx, lib.pk_svd_build(ctx, A, rank, 0, 0.0, 0, 0, ptr(sigma), ptr(V), ptr(U), C.byref(st)), 'pk_svd_build')
    s_ref = np.linalg.svd(S.toarray(), compute_uv=False)[:rank]
    assert st.converged == 1 and st.final_rel_residual <= 1e-12, (st.converged, st.final_rel_residual)
    assert np.allclose(sigma, s_ref, rtol=1e-9), np.abs(sigma / s_ref - 1).max()
    assert np.abs(V.T @ V - np.eye(rank)).max() < 1e-10 and np.abs(U.T @ U - np.eye(rank)).max() < 1e-9
    assert np.abs(S @ V - U * sigma).max() < 1e-8 * sigma[0]
    # the same build by block Lanczos (what pk_svd_build picks by itself once a Gramian step is heavy; forced here): the
    # factors of the filtered subspace iteration to the solver tolerance, in fewer Gramian steps, residual VERIFIED
    check(ctx, lib.pk_ctx_set_option(ctx, b'svd_method', 1), 'pk_ctx_set_option')      # 1 = block Lanczos
    sigma_l = np.empty(rank); V_l = np.empty((n_items, rank), order='F'); st_l = Stats()
    check(ctx, lib.pk_svd_build(ctx, A, rank, 0, 0.0, 0, 0, ptr(sigma_l), ptr(V_l), None, C.byref(st_l)), 'pk_svd_build(lanczos)')
    check(ctx, lib.pk_ctx_set_option(ctx, b'svd_method', 2), 'pk_ctx_set_option')      # 2 = filtered subspace iteration
    st_s = Stats()
    check(ctx, lib.pk_svd_build(ctx, A, rank, 0, 0.0, 0, 0, ptr(sigma), ptr(V), ptr(U), C.byref(st_s)), 'pk_svd_build(subspace)')
    check(ctx, lib.pk_ctx_set_option(ctx, b'svd_method', 0), 'pk_ctx_set_option')
    assert lib.pk_ctx_set_option(ctx, b'svd_method', 7) != 0 and lib.pk_ctx_set_option(ctx, b'no_such_option', 0) != 0
    assert st_l.converged == 1 and st_l.final_rel_residual <= 1e-12 and st_l.gramian_steps < st_s.gramian_steps, (
        st_l.gramian_steps, st_s.gramian_steps, st_l.final_rel_residual)
    assert np.allclose(sigma_l, s_ref, rtol=1e-10) and np.abs(V_l @ V_l.T - V @ V.T).max() < 1e-8
    print('coarse build: %d Gramian steps (block Lanczos) / %d (subspace iteration)' % (st_l.gramian_steps, st_s.gramian_steps))
    # CSR route for the test rows (= the training rows here), ids only (approximate fold-in route) and with scores
    T = vp()
    ind = S.indices.astype(np.int32); ptr64 = S.indptr.astype(np.int64); dat = S.data.astype(np.float64)
    check(ctx, lib.pk_mat_from_csr(ctx, n_users, n_items, S.nnz, ptr(ptr64), ptr(ind), ptr(dat), 1, C.byref(T)), 'pk_mat_from_csr')
    recs = np.empty((n_users, topk), dtype=np.int64)
    check(ctx, lib.pk_score_topk(ctx, n_items, rank, ptr(V), T, topk, 1, ptr(recs), None), 'pk_score_topk')
    coo = S.tocoo()
    order = np.lexsort((coo.col, coo.row))
    td = (coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order])
    Vc = np.ascontiguousarray(V)
    want = orc.svd_recommendations(Vc, td, (n_users, n_items), topk, True)
    full, sd = orc.svd_slice_recommendations(Vc, td, (n_users, n_items), 0, n_users)
    orc.downvote_seen_items(full, sd)
    clear = orc.boundary_gap(full, topk) > 1e-9
    assert clear.mean() > 0.99 and np.array_equal(recs[clear], want[clear]), int((recs[clear] != want[clear]).any(axis=1).sum())
    recs2 = np.empty_like(recs); sc = np.empty((n_users, topk))
    check(ctx, lib.pk_score_topk(ctx, n_items, rank, ptr(V), T, topk, 1, ptr(recs2), ptr(sc)), 'pk_score_topk(scores)')
    assert np.array_equal(recs2, recs) and (np.diff(sc, axis=1) <= 0).all()
    E = S @ Vc
    assert np.allclose(sc, np.take_along_axis(E @ Vc.T, recs, axis=1), rtol=1e-12, atol=1e-12)
    # filter_seen = 0 and the exact-row route (topk beyond the fused sweep)
    recs3 = np.empty((n_users, 60), dtype=np.int64)
    check(ctx, lib.pk_score_topk(ctx, n_items, rank, ptr(V), T, 60, 0, ptr(recs3), None), 'pk_score_topk(top-60)')
    top = np.argsort(-(E @ Vc.T), axis=1, kind='stable')[:, :60]
    sorted_scores = -np.sort(-(E @ Vc.T), axis=1)[:, :61]
    ok = (np.diff(-sorted_scores, axis=1) > 1e-9).all(axis=1)
    assert ok.mean() > 0.9 and np.array_equal(recs3[ok], top[ok])
    # the serving handle: factors, images and the renamed test rows stay on the device between calls; the test matrix may
    # be freed once the handle exists; every call returns what pk_score_topk returns
    T2 = vp()
    check(ctx, lib.pk_mat_from_csr(ctx, n_users, n_items, S.nnz, ptr(ptr64), ptr(ind), ptr(dat), 1, C.byref(T2)), 'pk_mat_from_csr')
    sv = vp()
    check(ctx, lib.pk_serving_create(ctx, n_items, rank, ptr(V), T2, C.byref(sv)), 'pk_serving_create')
    lib.pk_mat_free(ctx, T2)
    for rep in range(3):
        r4 = np.full((n_users, topk), -5, dtype=np.int64)
        check(ctx, lib.pk_serving_score(ctx, sv, topk, 1, ptr(r4), None), 'pk_serving_score')
        assert np.array_equal(r4, recs), rep
    r5 = np.empty_like(recs); sc5 = np.empty((n_users, topk))
    check(ctx, lib.pk_serving_score(ctx, sv, topk, 1, ptr(r5), ptr(sc5)), 'pk_serving_score(scores)')
    assert np.array_equal(r5, recs2) and np.array_equal(sc5, sc)
    r6 = np.empty((n_users, 60), dtype=np.int64)
    check(ctx, lib.pk_serving_score(ctx, sv, 60, 0, ptr(r6), None), 'pk_serving_score(top-60)')
    assert np.array_equal(r6, recs3)
    assert lib.pk_serving_score(ctx, sv, n_items + 1, 1, ptr(r6), None) != 0 and b'out of bounds' in lib.pk_ctx_error(ctx)
    lib.pk_serving_free(ctx, sv)
    # errors are codes + messages, not crashes
    bad = np.empty((n_users, topk), dtype=np.int64)
    rc = lib.pk_score_topk(ctx, n_items + 1, rank, ptr(V), T, topk, 1, ptr(bad), None)
    assert rc != 0 and b'number of items' in lib.pk_ctx_error(ctx)
    rc = lib.pk_svd_build(ctx, A, rank, 0, 1e-30, 1, 0, ptr(sigma), ptr(V), None, C.byref(st))
    assert rc == -4 and st.converged == 0 and b'not converged' in lib.pk_ctx_error(ctx)      # PK_E_NOCONV
    lib.pk_mat_free(ctx, A); lib.pk_mat_free(ctx, T)
    # ---- pk_hooi against the reference's golden CoFFee fixture (same start block as lib/tensor.py:57-63) ----
    for name in ('coffee_small', 'coffee_warm'):
        g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        idx = np.ascontiguousarray(g['train_idx'], dtype=np.int64)
        val = np.ascontiguousarray(g['train_val'], dtype=np.float64)
        shape = np.ascontiguousarray(g['train_shape'], dtype=np.int64)
        mlrank = np.ascontiguousarray(g['mlrank'], dtype=np.int32)
        rs = np.random.RandomState(int(g['seed']))
        u1s = np.ascontiguousarray(np.linalg.qr(rs.rand(shape[1], mlrank[1]), mode='reduced')[0])
        u2s = np.ascontiguousarray(np.linalg.qr(rs.rand(shape[2], mlrank[2]), mode='reduced')[0])
        n_it = int(g['num_iters'])
        u0 = np.empty((shape[0], mlrank[0])); u1 = np.empty((shape[1], mlrank[1])); u2 = np.empty((shape[2], mlrank[2]))
        core = np.empty(tuple(mlrank)); trace = np.zeros(n_it); iters = i32(0)
        check(ctx, lib.pk_hooi(ctx, len(val), ptr(idx), ptr(val), ptr(shape), ptr(mlrank), n_it, float(g['growth_tol']), ptr(u1s), ptr(u2s),
                               0, ptr(u0), ptr(u1), ptr(u2), ptr(core), ptr(trace), C.byref(iters)), 'pk_hooi')
        want = g['core_norm_trace']
        assert iters.value == len(want) and np.allclose(trace[:iters.value], want, rtol=1e-9), (iters.value, trace, want)
        for a, ref in ((u0, g['u0']), (u1, g['u1']), (u2, g['u2'])):
            assert np.abs(a @ a.T - ref @ ref.T).max() < 1e-8
        assert np.isclose(np.linalg.norm(core), np.linalg.norm(g['core']), rtol=1e-9)
        # the core IS the tensor contracted with the factors
        dense = np.zeros(tuple(shape)); np.add.at(dense, (idx[:, 0], idx[:, 1], idx[:, 2]), val)
        assert np.allclose(core, np.einsum('uif,ua,ib,fc->abc', dense, u0, u1, u2, optimize=True), atol=1e-9 * np.abs(core).max())
        # the default route factors the mode products (SpMM over two unfoldings + fp64-MFMA contractions); the per-entry
        # kernel (pk_ttm_f64 = dttm_seq restated) must give the same iteration
        check(ctx, lib.pk_ctx_set_option(ctx, b'hooi_ttm', 1), 'pk_ctx_set_option')
        v0 = np.empty_like(u0); v1 = np.empty_like(u1); v2 = np.empty_like(u2); core2 = np.empty_like(core); trace2 = np.zeros(n_it); it2 = i32(0)
        check(ctx, lib.pk_hooi(ctx, len(val), ptr(idx), ptr(val), ptr(shape), ptr(mlrank), n_it, float(g['growth_tol']), ptr(u1s), ptr(u2s),
                               0, ptr(v0), ptr(v1), ptr(v2), ptr(core2), ptr(trace2), C.byref(it2)), 'pk_hooi (per-entry products)')
        check(ctx, lib.pk_ctx_set_option(ctx, b'hooi_ttm', 0), 'pk_ctx_set_option')
        assert it2.value == iters.value and np.allclose(trace2[:it2.value], trace[:iters.value], rtol=1e-10)
        for a, b in ((u0, v0), (u1, v1), (u2, v2)):
            assert np.abs(a @ a.T - b @ b.T).max() < 1e-8
    # internal start (no start blocks given): a valid Tucker fit all the same
    check(ctx, lib.pk_hooi(ctx, len(val), ptr(idx), None, ptr(shape), ptr(mlrank), n_it, float(g['growth_tol']), None, None, 7,
                           ptr(u0), ptr(u1), ptr(u2), ptr(core), ptr(trace), C.byref(iters)), 'pk_hooi(seeded start)')
    assert np.abs(u1.T @ u1 - np.eye(mlrank[1])).max() < 1e-9 and iters.value >= 1 and trace[iters.value - 1] > 0
    lib.pk_ctx_destroy(ctx)
    print('COARSE_ABI_OK steps', st.gramian_steps)


if __name__ == '__main__':
    main()
