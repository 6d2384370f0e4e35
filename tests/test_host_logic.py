"""Host-side logic on CPU: CSR/task-plan builders, the ArrayData protocol, and the backend-agnostic
orchestration (solver, scoring pipeline, HOOI driver, model classes) driven through the TEST-ONLY
NumPy double of the device operator set (tests/numpy_ops.py) against the reference's golden vectors.
This proves the orchestration; the HIP kernels themselves are proven by the -m gpu tests."""
import os
import numpy as np
import pytest
import scipy.sparse as sps

from conftest import check_coffee_extras, load_golden, GoldenData
from numpy_ops import NumpyOps
from oracle import polara_oracle as orc
from polara_amd import csr as pcsr
from polara_amd.data import ArrayData
from polara_amd.models import SVDModel, CoffeeModel, ScaledSVD
from polara_amd.solver import svd_topk


def test_coo_to_csr_matches_scipy_and_sums_duplicates():
    rng = np.random.RandomState(0)
    r = rng.randint(0, 50, 3000)
    c = rng.randint(0, 40, 3000)
    v = rng.rand(3000)
    ip, ix, vl = pcsr.coo_to_csr(r, c, v, (50, 40))
    A = sps.coo_matrix((v, (r, c)), shape=(50, 40)).tocsr()
    assert np.array_equal(ip, A.indptr) and np.array_equal(ix, A.indices) and np.allclose(vl, A.data)
    tp, ti, tv = pcsr.csr_transpose(ip, ix, vl, 40)
    At = A.T.tocsr()
    At.sort_indices()
    assert np.array_equal(tp, At.indptr) and np.array_equal(ti, At.indices) and np.allclose(tv, At.data)
    with pytest.raises(ValueError):
        pcsr.coo_to_csr([0, 60], [0, 1], [1.0, 1.0], (50, 40))


@pytest.mark.parametrize('split', [4, 16, 1000])
def test_row_tasks_cover_every_nnz_once(split):
    rng = np.random.RandomState(1)
    counts = np.r_[0, rng.randint(0, 60, 30), 0, 500, 0]
    indptr = np.r_[0, np.cumsum(counts)].astype(np.int64)
    plan = pcsr.build_row_tasks(indptr, split=split)
    cov = np.zeros(indptr[-1], int)
    for row, b, e in zip(plan['task_row'], plan['task_begin'], plan['task_end']):
        assert indptr[row] <= b <= e <= indptr[row + 1] and e - b <= max(split, 1)
        cov[b:e] += 1
    assert (cov == 1).all()
    assert set(plan['task_row']) == set(range(len(counts)))  # empty rows get a task too
    slots = plan['task_slot'][plan['task_slot'] >= 0]
    assert np.array_equal(slots, np.arange(plan['n_slots']))
    for row, s0, s1 in zip(plan['long_row'], plan['long_slot_begin'], plan['long_slot_end']):
        assert (plan['task_row'][np.isin(plan['task_slot'], np.arange(s0, s1))] == row).all()


def test_nnz_balanced_partition():
    indptr = np.r_[0, np.cumsum(np.r_[np.full(10, 100), np.full(90, 1)])].astype(np.int64)
    b = pcsr.nnz_balanced_row_partition(indptr, 4)
    assert b[0] == 0 and b[-1] == 100 and (np.diff(b) >= 0).all()
    shard_nnz = np.diff(indptr[b])
    assert shard_nnz.max() <= 1.5 * indptr[-1] / 4


def test_arraydata_protocol_threshold_and_recovery():
    u = np.array([0, 0, 1, 1, 2, 2, 2])
    i = np.array([0, 1, 1, 2, 0, 2, 3])
    f = np.array([5., 2., 4., 1., 3., 5., 2.])
    hold = (np.array([0, 2]), np.array([3, 1]), np.array([4., 4.]))
    d = ArrayData((u, i, f), n_users=3, n_items=4, holdout=hold, warm_start=False)
    idx, val, shp = d.to_coo(feedback_threshold=3)
    assert shp == (3, 4) and len(val) == 4 and (val >= 3).all()       # data.py:777-791 filter_values=True
    tu, ti, tf = d.test_to_coo(feedback_threshold=3)
    assert np.array_equal(tu, [0, 0, 2, 2, 2]) and np.array_equal(tf, [5, 0, 3, 5, 0])  # zeroed, not dropped
    assert d.get_test_shape() == (2, 4)
    idx3, val3, shp3 = d.to_coo(tensor_mode=True)
    assert shp3 == (3, 4, 5) and (val3 == 1).all() and idx3[:, 2].max() == 4
    calls = []

    class M:
        def cb(self):
            calls.append(1)
    m = M()
    d.subscribe(d.on_update_event, m.cb)
    d.set_test_data(holdout=hold)
    assert calls == [1]


@pytest.mark.parametrize('name', ['svd_warm', 'svd_known', 'svd_fewunseen', 'svd_nofilter'])
def test_svd_model_orchestration_matches_reference(name):
    g = load_golden(name)
    m = SVDModel(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
    m.collect_recommend_stats = True
    m.build()
    assert len(m.training_time) == 1 and m._is_ready
    assert np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9)
    V = m.factors[m.data.fields.itemid]
    assert V.flags.f_contiguous and np.allclose(V @ V.T, g['V'] @ g['V'].T, atol=1e-8)
    assert m.factors[m.data.fields.userid] is None
    recs = m.recommendations
    notie = g['boundary_gap'] > 0
    assert recs.dtype == np.int64 and recs.shape == g['recs'].shape
    assert np.array_equal(recs[notie], g['recs'][notie])
    if name == 'svd_fewunseen':
        assert m.recommend_stats['flagged_users'] > 0   # exercised the exact two-class path
    if name == 'svd_warm':   # rank setter truncates, does not rebuild (models.py:812-832)
        m.rank = 5
        assert m._is_ready and m.factors['singular_values'].shape == (5,)
        assert np.array_equal(m.recommendations, g['recs_rank5'])
        m.rank = 9
        assert not m._is_ready


def test_scaled_svd_matches_reference():
    g = load_golden('svd_scaled')
    m = ScaledSVD(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    m.col_scaling, m.row_scaling = float(g['col_scaling']), float(g['row_scaling'])
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    m.build()
    assert m.method == 'PureSVD-s'
    assert np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9)
    notie = g['boundary_gap'] > 0
    assert np.array_equal(m.recommendations[notie], g['recs'][notie])
    # the scaled matrix itself equals the oracle's (= the reference's rescale_matrix chain)
    ip, ix, vals, shp = m._training_csr()
    ref = orc.scaled_training_matrix(g['train_idx'], g['train_val'], tuple(g['train_shape']),
                                     float(g['col_scaling']), float(g['row_scaling']))
    assert np.array_equal(ix, ref.indices) and np.allclose(vals, ref.data, rtol=1e-15)


@pytest.mark.parametrize('name', ['coffee_small', 'coffee_warm'])
def test_coffee_model_orchestration_matches_reference(name):
    g = load_golden(name)
    m = CoffeeModel(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    m.mlrank, m.topk, m.seed = tuple(int(x) for x in g['mlrank']), int(g['topk']), int(g['seed'])
    m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
    m.build()
    assert np.allclose(m.core_norm_trace, g['core_norm_trace'], rtol=1e-9)
    f = m.data.fields
    for key, ref in ((f.userid, g['u0']), (f.itemid, g['u1']), (f.feedback, g['u2'])):
        a = m.factors[key]
        assert np.allclose(a @ a.T, ref @ ref.T, atol=1e-8)
    assert m.factors['core'].shape == tuple(g['mlrank'])
    notie = g['boundary_gap'] > 0
    assert np.array_equal(m.recommendations[notie], g['recs'][notie])
    check_coffee_extras(m, g)          # models.py:1027-1092: unfolded slices, holdout slice, predict_feedback


def _check_full_feedback_mode(m, g):
    """mlrank[2] == number of feedback levels (BASELINE.json configs[3] asks for (30, 30, 5) on 5 rating levels):
    the reference raises here (svds needs k < min(shape), lib/tensor.py:79); the device path takes the full
    eigen-decomposition of the 5 x 5 Gram matrix instead.  Checked through properties of a Tucker fit."""
    f = m.data.fields
    u0, u1, u2, core = (m.factors[k] for k in (f.userid, f.itemid, f.feedback, 'core'))
    n2 = u2.shape[0]
    assert u2.shape == (n2, n2) and core.shape[2] == n2
    for u in (u0, u1, u2):
        assert np.abs(u.T @ u - np.eye(u.shape[1])).max() < 1e-9          # orthonormal factors
    # the core is the tensor contracted with the three factors; its norm is what hooi monitors
    idx, val, shp = m.data.to_coo(tensor_mode=True)
    dense = np.zeros(shp)
    np.add.at(dense, (idx[:, 0], idx[:, 1], idx[:, 2]), val)
    want = np.einsum('uif,ua,ib,fc->abc', dense, u0, u1, u2)
    assert np.allclose(core, want, atol=1e-9 * np.abs(want).max())
    assert np.isclose(np.linalg.norm(core), m.core_norm_trace[-1], rtol=1e-9)
    assert all(b >= a * (1 - 1e-12) for a, b in zip(m.core_norm_trace, m.core_norm_trace[1:]))   # monotone fit
    # a complete feedback-mode basis loses nothing along that mode: same fit as the (r0, r1, n2-1) model or better
    assert m.recommendations.shape == (g['recs'].shape[0], m.topk)


def test_coffee_full_feedback_mode_rank():
    g = load_golden('coffee_small')
    m = CoffeeModel(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    n_fdbk = int(m.data.to_coo(tensor_mode=True)[2][2])
    m.mlrank, m.topk, m.seed = (int(g['mlrank'][0]), int(g['mlrank'][1]), n_fdbk), int(g['topk']), int(g['seed'])
    m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
    m.build()
    _check_full_feedback_mode(m, g)


def test_topk_cache_rules_and_errors():
    g = load_golden('svd_nofilter')
    m = SVDModel(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    m.rank = int(g['rank'])
    m.topk = 10
    r10 = m.recommendations
    m.topk = 5                      # smaller k keeps the cache (models.py:123-128)
    assert m._recommendations is r10
    m.topk = 12
    assert m._recommendations is None
    m.topk = 10 ** 6
    with pytest.raises(ValueError):      # same failure class as numpy argpartition in models.py:490
        m.get_recommendations()
    with pytest.raises(AttributeError):   # not a LinearOperator (scipy's svds fails the same way in models.py:844)
        SVDModel.build(m, operator=object())


def test_solver_converges_on_planted_and_degenerate_inputs():
    from polara_amd.synth import planted_csr, csr_to_numpy
    c = csr_to_numpy(planted_csr(900, 400, 30, 12, seed=5, min_items=8, max_items=150))
    ops = NumpyOps()
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    _, s, V, st = svd_topk(ops, A, 12)
    ref = np.linalg.svd(A.m.toarray(), compute_uv=False)[:12]
    assert st['converged'] and np.allclose(s.numpy(), ref, rtol=1e-10)
    assert np.allclose(V.numpy().T @ V.numpy(), np.eye(12), atol=1e-10)
    # rank-deficient matrix with k beyond the numerical rank and block == n_items
    B = sps.csr_matrix(np.outer(np.arange(1, 9.), np.arange(1, 7.)))
    A2 = ops.csr(B.indptr, B.indices, B.data, B.shape)
    _, s2, _, st2 = svd_topk(ops, A2, 3)
    assert np.isclose(s2.numpy()[0], np.linalg.svd(B.toarray(), compute_uv=False)[0]) and s2.numpy()[1] < 1e-5 * s2.numpy()[0]


def test_get_test_matrix_matches_oracle():
    """models.py:180-211 surface: same matrix and slice triplet as the oracle's restatement."""
    g = load_golden('svd_known')
    m = SVDModel(GoldenData(g), ops=NumpyOps())
    td, shp = (g['test_user'], g['test_item'], g['test_fdbk']), tuple(int(x) for x in g['test_shape'])
    for sl in (None, (3, 40), (0, shp[0] + 5)):
        got, got_td = m.get_test_matrix(td, shp, sl)
        want, want_td = orc.get_test_matrix(td, shp, sl)
        assert got.shape == want.shape and (got != want).nnz == 0
        assert all(np.array_equal(a, b) for a, b in zip(got_td, want_td))
    auto, _ = m.get_test_matrix()
    assert auto.shape == shp
    with pytest.raises(ValueError):
        m.get_test_matrix(td)


def test_build_with_linear_operator_matches_svds():
    """build(operator=...) (models.py:835-844): a SciPy LinearOperator replaces the training matrix; checked
    against scipy's own svds of the same operator, including the recommendations that follow from it."""
    from scipy.sparse.linalg import aslinearoperator, svds
    import scipy.sparse as sps
    g = load_golden('svd_known')
    data = GoldenData(g)
    m = SVDModel(data, ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    A = m.get_training_matrix(dtype=np.float64)
    d = 1.0 / np.sqrt(1.0 + np.asarray(A.getnnz(axis=0)).ravel())          # an item-side scaling, as EIGENREC does
    op = aslinearoperator(A @ sps.diags(d))
    m.build(operator=op, return_factors=True)
    assert m._is_ready and len(m.training_time) == 1
    u, s, vt = svds(op, k=m.rank)
    order = np.argsort(-s)
    assert np.allclose(m.factors['singular_values'], s[order], rtol=1e-9)
    V = m.factors[data.fields.itemid]
    U = m.factors[data.fields.userid]
    assert np.abs(V @ V.T - vt.T @ vt).max() < 1e-8
    assert U.shape == (A.shape[0], m.rank) and np.abs(U @ U.T - u @ u.T).max() < 1e-8
    td, shp = (g['test_user'], g['test_item'], g['test_fdbk']), tuple(int(x) for x in g['test_shape'])
    Vref = np.ascontiguousarray(vt.T[:, order])
    want = orc.svd_recommendations(Vref, td, shp, m.topk, True)
    scores, slice_data = orc.svd_slice_recommendations(Vref, td, shp, 0, shp[0])
    orc.downvote_seen_items(scores, slice_data)
    # rows whose k-th/(k+1)-th scores (or any two inside the top-k) tie are implementation-defined
    top = -np.sort(-scores, axis=1)[:, :m.topk + 1]
    clear = (np.diff(-top, axis=1) > 1e-9 * np.abs(top[:, :1])).all(axis=1)
    got = m.recommendations
    assert clear.mean() > 0.5 and np.array_equal(got[clear], want[clear])
    with pytest.raises(ValueError):
        m.build(operator=aslinearoperator(A[:, :-1]))


def test_build_with_sparse_operator_forms_match_svds():
    """build(operator=...) with the device-resident forms: a sparse matrix (HybridSVD's precomputed auxiliary
    matrix, hybrid/models.py:357-363) and a SparseProduct of factors (the chained form, :364-381) give the
    factorization scipy's svds gives for the multiplied-out matrix."""
    from scipy.sparse.linalg import svds
    import scipy.sparse as sps
    from polara_amd.operator import SparseProduct
    g = load_golden('svd_known')
    data = GoldenData(g)
    A = SVDModel(data, ops=NumpyOps()).get_training_matrix(dtype=np.float64)
    rng = np.random.RandomState(5)
    n_users, n_items = A.shape
    # sparse lower-triangular "Cholesky-like" factors on both sides (unit diagonal + a few sub-diagonal entries)
    Ls = (sps.eye(n_items) + 0.3 * sps.tril(sps.random(n_items, n_items, 0.02, random_state=rng), -1)).tocsr()
    Lk = (sps.eye(n_users) + 0.3 * sps.tril(sps.random(n_users, n_users, 0.01, random_state=rng), -1)).tocsr()
    full = (Lk.T @ A @ Ls).tocsr()
    rank = int(g['rank'])
    u, s, vt = svds(full, k=rank)
    order = np.argsort(-s)
    out = []
    for op in (full, SparseProduct(Lk.T, A, Ls), SparseProduct(A, Ls.tocoo())):
        m = SVDModel(data, ops=NumpyOps())
        m.verbose = False
        m.rank, m.topk = rank, int(g['topk'])
        m.build(operator=op, return_factors=True)
        out.append(m)
    for m in out[:2]:
        assert np.allclose(m.factors['singular_values'], s[order], rtol=1e-9)
        V, U = m.factors[data.fields.itemid], m.factors[data.fields.userid]
        assert np.abs(V @ V.T - vt.T @ vt).max() < 1e-8 and np.abs(U @ U.T - u @ u.T).max() < 1e-8
    assert np.array_equal(out[0].recommendations, out[1].recommendations)
    s2 = svds((A @ Ls).tocsr(), k=rank, return_singular_vectors=False)
    assert np.allclose(out[2].factors['singular_values'], np.sort(s2)[::-1], rtol=1e-9)
    with pytest.raises(ValueError):
        SparseProduct(A, Lk)                      # shapes do not chain
    with pytest.raises(ValueError):
        out[0].build(operator=SparseProduct(A, Ls[:, :-1]))   # an item short


def test_factor_image_rejects_factors_outside_the_fp32_range():
    """The fp32 images behind the sweep and the fold-in need the factor scale to be an fp32 normal number: anything
    else fails loudly instead of producing uncertified lists."""
    import torch
    from polara_amd import scoring
    ops = NumpyOps()
    rng = np.random.RandomState(0)
    V = rng.randn(40, 5)
    scoring.FactorImage(ops, torch.from_numpy(V))                       # fine
    scoring.FactorImage(ops, torch.zeros(40, 5, dtype=torch.float64))   # a degenerate model is still a model
    for scale in (1e-40, 1e35, np.inf, np.nan):
        with pytest.raises(ValueError, match='fp32'):
            scoring.FactorImage(ops, torch.from_numpy(V * scale))


def test_single_user_conveniences_on_array_data():
    """models.py:277-356, 488-563 on ArrayData (ids are internal): `_user_scores`, `show_recommendations` for a test
    user and for ad-hoc users, `downvote_seen_items` / `get_topk_elements` / `topsort` against the oracle."""
    g = load_golden('svd_known')
    data = GoldenData(g)
    m = SVDModel(data, ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    m.build()
    rng = np.random.RandomState(1)
    sc = rng.randn(9, 60)
    seen = (np.repeat(np.arange(9), 4), rng.randint(0, 60, 36), np.ones(36))
    a, b = sc.copy(), sc.copy()
    m.downvote_seen_items(a, seen)
    orc.downvote_seen_items(b, seen)
    assert np.array_equal(a, b)
    assert np.array_equal(m.get_topk_elements(a, 7), orc.get_topk_elements(b, 7))
    assert np.array_equal(m.topsort(sc[3], 5), orc.topsort(sc[3], 5))
    one = sc[0].copy()
    m.downvote_seen_items(one, (np.zeros(3, dtype=int), np.array([2, 5, 7])))      # single-user form
    assert set(np.argsort(one)[:3]) == {2, 5, 7}
    with pytest.raises(ValueError):
        m.get_topk_elements(sc, 61)
    notie = g['boundary_gap'] > 0
    for row in np.flatnonzero(notie)[:5]:
        top, seen_items = m.show_recommendations(int(row))
        assert np.array_equal(top, g['recs'][row])
        assert set(seen_items) == set(g['test_item'][g['test_user'] == row])
    # an ad-hoc user: the test set is swapped in and restored, the cached lists stay untouched
    # (a plain ArrayData: the golden wrapper above answers test queries from the fixture, whatever the test set is)
    from polara_amd.data import ArrayData
    idx = g['train_idx']
    data = ArrayData((idx[:, 0], idx[:, 1], g['train_val']), n_users=int(g['train_shape'][0]), n_items=int(g['train_shape'][1]),
                     test=(g['test_user'], g['test_item'], g['test_fdbk']))
    m = SVDModel(data, ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    m.build()
    cached = m.recommendations
    before = data.test
    top, seen_items = m.show_recommendations([3, 9, 27], topk=4)
    assert data.test is before and m.recommendations is cached and m.topk == int(g['topk'])
    assert len(top) == 4 and set(seen_items) == {3, 9, 27} and not set(top) & {3, 9, 27}
    V = m.factors[data.fields.itemid]
    fmax = float(np.max(data.training.feedback))
    prof = np.zeros(V.shape[0]); prof[[3, 9, 27]] = fmax
    s = (prof @ V) @ V.T
    s[[3, 9, 27]] = -np.inf
    assert np.array_equal(top, np.argsort(-s, kind='stable')[:4])
    top_w, _ = m.show_recommendations({3: 1.0, 9: 5.0}, topk=4)
    prof = np.zeros(V.shape[0]); prof[3], prof[9] = 1.0, 5.0
    s = (prof @ V) @ V.T
    s[[3, 9]] = -np.inf
    assert np.array_equal(top_w, np.argsort(-s, kind='stable')[:4])
    with pytest.raises(ValueError):
        m.show_recommendations('user')


def test_block_lanczos_matches_the_subspace_iteration_and_arpack(monkeypatch):
    """solver._block_lanczos (Rayleigh-Ritz over the whole Krylov space, nested solve of the projected problem, monitors
    on a worker thread) against the filtered subspace iteration and against the reference's own call (scipy svds = ARPACK,
    tol 0): singular values to 1e-10, projectors to 1e-8, fewer Gramian steps, a VERIFIED residual below the tolerance;
    the same factors with the monitors switched off (monitor_lag=0: every look on the calling thread)."""
    from scipy.sparse.linalg import svds
    from polara_amd.synth import planted_csr
    m = planted_csr(5000, 2000, 50, 16, levels=5, seed=7, min_items=6, max_items=300)
    ops = NumpyOps()
    A = ops.csr(m['indptr'].numpy(), m['indices'].numpy(), m['values'].numpy().astype(np.float64), m['shape'])
    k = 16
    out = {}
    for meth in ('subspace', 'lanczos'):
        _, s, V, st = svd_topk(ops, A, k, method=meth)
        out[meth] = (s.numpy(), V.numpy(), st)
        assert st['converged'] and st['final_rel_residual'] <= 1e-12
    sl, Vl, stl = out['lanczos']
    ss, Vs, sts = out['subspace']
    assert stl['method'] == 'lanczos' and 'lanczos_fallback' not in stl and stl['verified_rel_residual'] <= 1e-12
    assert stl['gramian_steps'] < sts['gramian_steps'] and stl['gramian_steps'] == stl['lanczos_steps'] + 1
    assert np.allclose(sl, ss, rtol=1e-10) and np.abs(Vl @ Vl.T - Vs @ Vs.T).max() < 1e-8
    np.random.seed(0)
    _, s_ref, vt = svds(A.m, k=k, tol=0)
    assert np.allclose(np.sort(s_ref)[::-1], sl, rtol=1e-10) and np.abs(vt.T @ vt - Vl @ Vl.T).max() < 1e-8
    _, s0, V0, st0 = svd_topk(ops, A, k, method='lanczos', monitor_lag=0)
    assert np.allclose(s0.numpy(), sl, rtol=1e-12) and np.abs(V0.numpy() @ V0.numpy().T - Vl @ Vl.T).max() < 1e-9
    # 'auto' keeps small matrices on the subspace iteration (the projected eigenproblems cost more than they save there)
    _, _, _, sta = svd_topk(ops, A, k)
    assert sta['method'] == 'subspace'


def test_narrow_krylov_blocks_give_the_same_factors():
    """Round 6: the width of a Krylov block is decoupled from the nested width l = k + guard vectors.  A narrower block takes
    more steps (about (l / b)^(0.33 + 0.035 log2(l / b)) times) and gathers fewer columns in total; the factors are those of the full-width build,
    the verified residual stays below the tolerance, blocks narrower than k included (the Ritz vectors come from the whole
    Krylov space)."""
    from polara_amd.synth import planted_csr
    m = planted_csr(5000, 2000, 50, 16, levels=5, seed=7, min_items=6, max_items=300)
    ops = NumpyOps()
    A = ops.csr(m['indptr'].numpy(), m['indices'].numpy(), m['values'].numpy().astype(np.float64), m['shape'])
    k = 16
    _, s0, V0, st0 = svd_topk(ops, A, k, method='lanczos', krylov_block=32, monitor_lag=3)
    assert st0['krylov_block'] == st0['block'] == 32
    cols0 = st0['lanczos_steps'] * 32
    for kb, lag in ((8, 3), (12, 0), (16, 3)):
        _, s, V, st = svd_topk(ops, A, k, method='lanczos', krylov_block=kb, monitor_lag=lag)
        assert st['method'] == 'lanczos' and 'lanczos_fallback' not in st and st['krylov_block'] == kb and st['block'] == 32
        assert st['converged'] and st['verified_rel_residual'] <= 1e-12
        assert st['lanczos_steps'] * kb < cols0      # fewer gathered columns in total (in more, narrower steps)
        assert np.allclose(s.numpy(), s0.numpy(), rtol=1e-11) and np.abs(V.numpy() @ V.numpy().T - V0.numpy() @ V0.numpy().T).max() < 1e-9
    with pytest.raises(ValueError):
        svd_topk(ops, A, k, method='krylov')


def test_choice_of_the_krylov_block_width_and_of_the_method():
    """solver.choose_krylov_block / choose_method (the same rule in csrc/driver.hip): never wider than the nested width, narrow
    where the sparse products dominate a step, and the subspace iteration where a build is all fixed costs."""
    from polara_amd.solver import choose_krylov_block, choose_method, KRYLOV_WIDTHS
    for nnz, n_items, l in ((2e7, 26744, 64), (2e7, 26744, 128), (1e8, 100000, 64), (5e7, 500000, 256), (1e6, 3706, 24), (500, 40, 8)):
        for world in (1, 2, 8):
            b = choose_krylov_block(nnz, n_items, l, world)
            assert b <= l and (b in KRYLOV_WIDTHS or b == l)
    assert choose_krylov_block(1e8, 100000, 64) == 16          # S-1M: 169 -> 80 ms measured
    assert choose_krylov_block(2e7, 26744, 64) in (16, 32)
    assert choose_method(2e7, 26744, 64) == 'lanczos' and choose_method(1e8, 100000, 64) == 'lanczos'
    assert choose_method(1e6, 3706, 24) == 'subspace' and choose_method(6e4, 800, 24) == 'subspace'
    assert choose_method(float('inf'), 1000, 24) == 'lanczos'  # an operator that does not say how many entries it holds


def test_block_lanczos_hands_rank_deficient_matrices_to_the_subspace_iteration():
    """Matrices the Krylov recurrence cannot finish on — exact rank below the block width (the residual block loses rank
    at once), an identity-like Gramian, fewer than four blocks of room — go to the filtered subspace iteration, whose
    rebuild path is made for them; the factors are those of a dense SVD either way."""
    rng = np.random.default_rng(5)
    ops = NumpyOps()
    cases = {'rank6_ask10': (sps.csr_matrix(rng.standard_normal((400, 6)) @ rng.standard_normal((6, 300))), 10),
             'orthogonal_rows': (sps.csr_matrix(np.eye(260)[:, :200] * 3.0), 5),
             'narrow': (sps.random(300, 60, density=0.2, random_state=3, format='csr'), 12)}
    for name, (M, k) in cases.items():
        M = sps.csr_matrix(M)
        M.sort_indices()
        A = ops.csr(M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64), M.shape)
        _, s, V, st = svd_topk(ops, A, k, method='lanczos')
        sd = np.linalg.svd(M.toarray(), compute_uv=False)[:k]
        assert st['converged'], (name, st)
        if name != 'orthogonal_rows':      # (a multiple of the identity: T_1 already holds the answer, the first look accepts it)
            assert 'lanczos_fallback' in st and st['method'].startswith('subspace'), (name, st)
        assert np.allclose(s.numpy(), sd, rtol=1e-9, atol=1e-9 * sd[0]), name


def test_block_lanczos_stops_at_the_widest_gram_operand_and_falls_back(monkeypatch):
    """the Krylov basis is an operand of pk_gram_f64 (at most 4096 columns).  A recurrence that would outgrow
    it (rank 100: block 128, 32 blocks) must end as a breakdown — the subspace iteration takes over — not as an error
    out of the kernel launcher.  The limit is lowered here so that the planted matrix hits it after four blocks."""
    from polara_amd import solver
    from polara_amd.synth import planted_csr
    m = planted_csr(3000, 1200, 40, 12, levels=5, seed=11, min_items=6, max_items=200)
    ops = NumpyOps()
    A = ops.csr(m['indptr'].numpy(), m['indices'].numpy(), m['values'].numpy().astype(np.float64), m['shape'])
    k = 12
    _, s_ref, _, st_ref = svd_topk(ops, A, k, method='subspace')
    b = st_ref['block']
    monkeypatch.setattr(solver, 'MAX_KRYLOV_COLS', 4 * b)          # room for exactly four blocks: too few to converge
    _, s, V, st = svd_topk(ops, A, k, method='lanczos')
    assert st['converged'] and 'lanczos_fallback' in st and 'not converged in 4 blocks' in st['lanczos_fallback'], st
    assert st.get('krylov_dim', 0) <= 4 * b
    assert np.allclose(s.numpy(), s_ref.numpy(), rtol=1e-10)
    monkeypatch.setattr(solver, 'MAX_KRYLOV_COLS', 3 * b)          # fewer than four blocks: no Krylov space at all
    _, s3, _, st3 = svd_topk(ops, A, k, method='lanczos')
    assert 'at most 3 blocks' in st3['lanczos_fallback'] and np.allclose(s3.numpy(), s_ref.numpy(), rtol=1e-10)


def test_unconverged_build_raises_like_arpack():
    """a build that stops at max_outer without converging must not mark the model ready silently; the
    reference's svds raises ArpackNoConvergence (models.py:844)."""
    import warnings
    from polara_amd.solver import NoConvergence
    g = load_golden('svd_warm')
    m = SVDModel(GoldenData(g), ops=NumpyOps())
    m.verbose = False
    m.rank = int(g['rank'])
    m.svd_max_outer = 1
    with pytest.raises(NoConvergence) as err:
        m.build()
    assert not m._is_ready and err.value.stats['converged'] is False and err.value.V.shape[1] == m.rank
    assert err.value.stats['final_rel_residual'] > m.svd_tol
    m.svd_on_no_convergence = 'warn'
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m.build()
    assert m._is_ready and any('did not converge' in str(x.message) for x in w)
    m.svd_max_outer = 200
    m.build()
    assert m.build_stats['converged'] and m.build_stats['final_rel_residual'] <= m.svd_tol


def test_bench_compact_line_is_small_and_parses():
    """the driver could not parse round 2's 25 KB stdout line.  bench.compact_line is a pure function of the
    measurement record: on a canned record with every block present (and absurdly long free-text fields) the line stays
    under 3 000 bytes, round-trips through json and carries the fields the driver and the judge read."""
    import json
    import bench
    long_text = 'x' * 5000
    head = {
        'value': 158712345.678, 'ms_per_step': 0.87261234, 'latency_ms_per_pass': 1.1, 'workload': bench.WORKLOAD_TEXT['ml20m'] +
        ', PureSVD rank=50, top-10, all users scored', 'n_users': 138493, 'n_items': 26744, 'nnz': 20000263, 'rank': 50,
        'topk': 10, 'prune': True, 'launch': 'python, kernel by kernel ' + long_text, 'launch_short': 'python',
        'launches_per_pass': 14, 'build_s': 0.0651234,
        'build': {'solver_s': 0.052, 'gramian_steps': 37, 'converged': True, 'spmm_ms': 40.1, 'note': long_text},
        'score': {'swept_fraction': 0.0946, 'kernel_ms': {'a': 1.0}},
        'roofline': {'kernel': 'score_candidates_kernel', 'bound': 'mfma', 'dtype': 'bf16 (split product)', 'achieved': 363.2,
                     'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.1452, 'avg_ms': 0.1853, 'launches_per_pass': 2,
                     'swept_fraction': 0.0946, 'traffic': 1.2e8, 'note': long_text},
        'roofline_build': {'kernel': 'spmm_csr_groups_kernel', 'bound': 'hbm', 'achieved': 401.0, 'peak': 8000.0, 'unit': 'GB/s',
                           'frac': 0.05, 'launches': 330, 'total_ms': 40.0, 'traffic': 1.64e9,
                           'algorithmic_bytes_per_product': 2.16e8, 'gather_note': long_text},
        'roofline_foldin': {'kernel': 'fold_q20_kernel', 'bound': 'hbm', 'achieved': 1102.0, 'peak': 8000.0, 'unit': 'GB/s',
                            'frac': 0.1378, 'avg_ms': 0.2014, 'traffic': 3.795e8, 'refolded_users': 1267, 'refold_ms': 0.052,
                            'algorithmic_bytes': 1.84e8, 'image_row_bytes': 128, 'gather_GBps': 12700.0, 'note': long_text},
        'cold': {'ops_create_s': 0.2719, 'warm_up_s': 0.1787, 'build_cold_s': 0.04363, 'solver_cold_s': 0.03505,
                 'first_pass_ms': 1.364, 'hw_queues': 8, 'hw_queues_in_time': True},
        'cpu_baseline': {'value': 9923.0, 'unit': 'users/s', 'cores': 128, 'kind': 'port', 'sample': long_text,
                         'gpu_vs_cpu_identical_rows': 1.0, 'build_s': 21.3, 'build_whole_matrix': True,
                         'speedup_scoring': 15994.0, 'speedup_build': 327.0, 'speedup_build_plus_score': 534.0},
    }
    line = bench.compact_line(head, 1, 20, 5, {'flat_norm': 44.9e6, 'pop25_norm': 80e6, 'no_prune': 60.7e6})
    assert len(line) < 3000 and '\n' not in line
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'build_s', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['vs_baseline'] is None
    assert d['config']['workload'].startswith('ML-20M') and d['config']['swept_fraction'] == 0.0946
    assert set(d['config']['adversarial_users_per_s']) == {'flat_norm', 'pop25_norm', 'no_prune'}
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline'])
    # round 5: the fold-in's own roofline and what a process pays once ride in the line
    assert {'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_ms', 'traffic', 'refolded_users', 'refold_ms'} <= set(d['roofline_foldin'])
    assert abs(d['roofline_foldin']['frac'] - d['roofline_foldin']['achieved'] / d['roofline_foldin']['peak']) < 1e-3
    assert d['cold'] == {'ops_create_s': 0.2719, 'warm_up_s': 0.1787, 'build_cold_s': 0.04363, 'solver_cold_s': 0.03505,
                         'first_pass_ms': 1.364, 'hw_queues': 8, 'hw_queues_in_time': True}
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(d['cpu_baseline']) and len(d['cpu_baseline']['sample']) <= 200
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-3
    # no nested free text anywhere
    assert long_text[:300] not in line
    # a multi-GPU line (no cpu_baseline, no adversarial rows) has the same shape
    head2 = {k: v for k, v in head.items() if k != 'cpu_baseline'}
    head2['dist'] = {'world': 8, 'backend': 'nccl', 'rccl': '2.26.6', 'device': 'AMD Instinct MI355X ' + long_text,
                     'users_per_rank': [17312] * 8, 'nnz_per_rank': [2500033] * 8, 'device_of_rank': list(range(8)),
                     'build_collectives': {'all_gather': 37, 'reduce_scatter': 37, 'all_reduce': 90, 'MB': 1013.123456},
                     'scoring_collectives': 0}
    head2['rank100_top20'] = {'users_per_s': 82.6e6, 'ms_per_step': 1.68, 'build_s': 0.135, 'gramian_steps': 52}
    line2 = bench.compact_line(head2, 8, 20, 5, None, scale=1.0)
    assert len(line2) < 3000 and long_text[:300] not in line2
    d2 = json.loads(line2)
    assert d2['n_gpus'] == 8 and 'cpu_baseline' not in d2 and 'users sharded over 8' in d2['config']['parallelism']
    assert d2['scaling'] == 'strong' and d2['dist']['world'] == 8 and d2['dist']['backend'] == 'nccl' and d2['dist']['rccl']
    assert len(d2['dist']['users_per_rank']) == 8 and d2['dist']['build_collectives']['reduce_scatter'] == 37
    assert d2['dist']['scoring_collectives'] == 0 and d2['config']['configs2_rank100_top20']['users_per_s'] == 82.6e6


def test_bench_refuses_counters_of_a_profile_taken_on_other_kernel_sources(tmp_path, monkeypatch):
    """`roofline.traffic` comes from committed rocprofv3 --pmc summaries.  A summary carries the hashes of the
    kernel sources of the tree it was taken in (tools/summarize_rocprof.py); bench.py compares them with the sources here
    and marks the profile stale — and leaves the counters out of the record — when score.hip / spmm.hip differ or the
    summary has no hashes (rounds 2-3)."""
    import shutil
    import bench
    sys_tools = os.path.join(bench.ROOT, 'tools')
    import sys
    sys.path.insert(0, sys_tools)
    from summarize_rocprof import kernel_source_hashes
    now = kernel_source_hashes(bench.ROOT)
    assert set(now) >= {'score.hip', 'spmm.hip'}
    root = tmp_path / 'repo'
    (root / 'profiles').mkdir(parents=True)
    (root / 'tools').mkdir()
    shutil.copy(os.path.join(sys_tools, 'summarize_rocprof.py'), root / 'tools')
    shutil.copytree(os.path.join(bench.ROOT, 'polara_amd', 'csrc'), root / 'polara_amd' / 'csrc', ignore=shutil.ignore_patterns('_obj'))
    row = '%-72s %-22s %10d %18.1f %18.1f\n'

    def write(hashes):
        head = '# commit abc1234\n' + ''.join('# sha256 %s %s\n' % kv for kv in hashes.items())
        for kind, per in (('fetch', 400000.0), ('write', 50000.0)):
            with open(root / 'profiles' / ('r04_ml20m_pmc_%s_size.txt' % kind), 'w') as f:
                f.write(head)
                f.write(row % ('void score_candidates_kernel<4, 16, false, true, false>', kind.upper() + '_SIZE', 4, 4 * per, per))
                f.write(row % ('void spmm_csr_groups_kernel<float, 4, double, false, true>', kind.upper() + '_SIZE', 60, 60 * per, per))
    monkeypatch.setattr(bench, 'ROOT', str(root))
    write(now)
    t = bench.pmc_traffic('ml20m')
    assert t['round'] == 'r04' and t['commit'] == 'abc1234' and t['stale_score'] is False and t['stale_spmm'] is False and t['score'] > 0
    write(dict(now, **{'score.hip': '0' * 16}))
    t = bench.pmc_traffic('ml20m')
    assert t['stale_score'] is True and t['stale_spmm'] is False
    write({})
    t = bench.pmc_traffic('ml20m')
    assert t['stale_score'] is True and t['stale_spmm'] is True


def test_local_movielens_files_become_the_bench_matrix(tmp_path):
    """datasets.load_movielens (the reference's loader, datasets/movielens.py:11-80, minus the download): old `::` format
    without a header, new comma format with one, plain file or zip archive; users / items renumbered in sorted id
    order, duplicate pairs summed, canonical CSR — what bench.py runs on when data/ml-20m* or data/ml-1m* exists."""
    import zipfile
    from polara_amd.datasets import load_movielens, find_movielens
    rows = [(7, 300, 4.0, 1), (2, 100, 5.0, 2), (7, 100, 0.5, 3), (9, 200, 3.5, 4), (2, 300, 1.0, 5), (2, 100, 1.0, 6)]
    want = sps.coo_matrix(([r[2] for r in rows], ([{2: 0, 7: 1, 9: 2}[r[0]] for r in rows], [{100: 0, 200: 1, 300: 2}[r[1]] for r in rows])),
                          shape=(3, 3)).tocsr()
    (tmp_path / 'data' / 'ml-1m').mkdir(parents=True)
    dat = tmp_path / 'data' / 'ml-1m' / 'ratings.dat'
    dat.write_text(''.join('%d::%d::%g::%d\n' % r for r in rows))
    (tmp_path / 'data' / 'ml-20m').mkdir()
    csv = tmp_path / 'data' / 'ml-20m' / 'ratings.csv'
    csv.write_text('userId,movieId,rating,timestamp\n' + ''.join('%d,%d,%g,%d\n' % r for r in rows))
    z = tmp_path / 'ml-20m.zip'
    with zipfile.ZipFile(z, 'w') as zf:
        zf.write(csv, 'ml-20m/ratings.csv')
    assert find_movielens('ml1m', str(tmp_path)) == str(dat) and find_movielens('ml20m', str(tmp_path)) == str(csv)
    assert find_movielens('s1m', str(tmp_path)) is None
    for path in (dat, csv, z):
        c = load_movielens(str(path))
        got = sps.csr_matrix((c['values'], c['indices'], c['indptr']), shape=c['shape'])
        assert c['shape'] == (3, 3) and (got != want).nnz == 0 and got.has_sorted_indices
        assert list(c['users']) == [2, 7, 9] and list(c['items']) == [100, 200, 300] and c['indices'].dtype == np.int32


def test_bench_gpus_flag_launches_ranks_or_fails_loudly(monkeypatch):
    """`--gpus N` was parsed and never read.  Started bare with N > 1 bench.py re-executes itself under
    torch.distributed.run with N ranks; on a box with fewer GPUs it exits with a message instead of measuring one GPU;
    under a launcher whose WORLD_SIZE disagrees with the flag it refuses."""
    import argparse
    import subprocess
    import bench
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('PK_BENCH_DEBUG_BACKEND', raising=False)
    assert bench.ensure_world(argparse.Namespace(gpus=1)) is None
    with pytest.raises(SystemExit) as e:          # this container has no GPU at all
        bench.ensure_world(argparse.Namespace(gpus=2))
    assert 'needs 2 visible GPUs' in str(e.value)
    monkeypatch.setenv('WORLD_SIZE', '4')
    with pytest.raises(SystemExit) as e:
        bench.ensure_world(argparse.Namespace(gpus=8))
    assert 'must agree' in str(e.value)
    assert bench.ensure_world(argparse.Namespace(gpus=4)) is None
    # the re-exec itself: the command line is torchrun's, every flag passed through, the child's exit code returned
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setenv('PK_BENCH_DEBUG_BACKEND', 'gloo')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(bench.sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '3'])
    with pytest.raises(SystemExit) as e:
        bench.ensure_world(argparse.Namespace(gpus=2))
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '2', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_training_rows_shortcut_builds_the_protocol_test_matrix():
    """models._training_rows_test_csr (holdout names every user, no explicit test set): the test CSR from the training
    columns as they lie equals the one from the protocol's sorted triplets (`_get_test_data`, models.py:227-257) — same
    lists — and the shortcut steps aside when a user has no interactions (the protocol renumbers rows then)."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import SVDModel, RecommenderModel

    class ProtocolData(ArrayData):                    # overriding test_to_coo switches the shortcut off
        def test_to_coo(self, *a, **k):
            return ArrayData.test_to_coo(self, *a, **k)

    rs = np.random.RandomState(11)
    n_users, n_items, n = 400, 120, 9000
    u = np.r_[np.arange(n_users), rs.randint(0, n_users, n)]          # every user at least once, unsorted
    i = rs.randint(0, n_items, len(u))
    f = rs.choice([0.0, 1.0, 2.0, 3.5, 5.0], len(u))                  # explicit zeros: not scored, still seen
    perm = rs.permutation(len(u))
    u, i, f = u[perm], i[perm], f[perm]
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    recs, used = [], []
    for cls in (ArrayData, ProtocolData):
        m = SVDModel(cls((u, i, f), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False), ops=NumpyOps())
        m.verbose = False
        m.rank, m.topk = 8, 7
        m.build()
        used.append(m._training_rows_test_csr() is not None)
        recs.append(m.get_recommendations())
    assert used == [True, False]
    assert np.array_equal(recs[0], recs[1])
    # a user without interactions: the shortcut declines, the protocol path runs
    keep = u != 5
    m = SVDModel(ArrayData((u[keep], i[keep], f[keep]), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False), ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk = 8, 7
    m.build()
    assert m._training_rows_test_csr() is None
    assert m.get_recommendations().shape[1] == 7


@pytest.mark.parametrize('gaps', [False, True])
def test_explicit_test_set_shortcut_builds_the_protocol_test_matrix(gaps):
    """The other branch of models._training_rows_test_csr: an explicit (warm-start) test set goes to the device as it
    lies, sortedness / gap checks and the gap-free renumbering of models.py:244-255 run there (scoring.renumbered_test_rows);
    same lists as through the protocol's host passes, with and without gaps in the test users' ids, and the same
    complaint about an unsorted set."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import SVDModel
    from polara_amd import scoring

    class ProtocolData(ArrayData):
        def test_to_coo(self, *a, **k):
            return ArrayData.test_to_coo(self, *a, **k)

    rs = np.random.RandomState(21)
    n_users, n_items, n = 300, 100, 7000
    u, i = rs.randint(0, n_users, n), rs.randint(0, n_items, n)
    f = rs.choice([0.0, 1.0, 2.0, 4.5], n)
    test_users = np.arange(0, 120, 2 if gaps else 1) + (7 if gaps else 0)        # ids 7, 9, 11, ... or 0, 1, 2, ...
    tu = np.repeat(test_users, 6)
    ti = rs.randint(0, n_items, len(tu))
    tf = rs.choice([0.0, 1.0, 3.0], len(tu))
    hold = (test_users, rs.randint(0, n_items, len(test_users)), np.ones(len(test_users)))
    recs, used = [], []
    for cls in (ArrayData, ProtocolData):
        d = cls((u, i, f), n_users=n_users, n_items=n_items, test=(tu, ti, tf), holdout=hold, warm_start=True)
        m = SVDModel(d, ops=NumpyOps())
        m.verbose = False
        m.rank, m.topk = 6, 5
        m.build()
        used.append(m._training_rows_test_csr() is not None)
        recs.append(m.get_recommendations())
    assert used == [True, False]
    assert recs[0].shape == (len(test_users), 5) and np.array_equal(recs[0], recs[1])
    with pytest.raises(AssertionError):
        scoring.renumbered_test_rows(NumpyOps(), np.array([0, 2, 1]))
    assert scoring.renumbered_test_rows(NumpyOps(), np.array([5])).tolist() == [0]
    assert scoring.renumbered_test_rows(NumpyOps(), np.array([3, 3, 8, 8, 9])).tolist() == [0, 0, 1, 1, 2]


def test_the_cost_models_constants_are_one_table_in_each_language():
    """the C++ statement of the solver (csrc/driver.hip, namespace model) and the Python one
    (polara_amd/machine_model.py) must price a step alike — every rank of a sharded build derives its sequence of collectives
    (two-panel exchange or not, method, block width) from these numbers.  The C++ table is read from the source."""
    import re
    from polara_amd.machine_model import value
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'polara_amd', 'csrc', 'driver.hip')).read()
    block = src.split('namespace model {')[1].split('}  // namespace model')[0]
    table = {name: float(val) for val, name in re.findall(r'constexpr double k\w+ = ([0-9.e+-]+);\s*//\s*(\w+)', block)}
    assert set(table) == {'dense_f64_flops', 'lanczos_step_fixed_s', 'nested_solve_s', 'xgmi_bus_Bps', 'collective_step_s'}
    for name, val in table.items():
        assert val == value(name), (name, val, value(name))
    assert src.count('100e9') == 1 and src.count('20e12') == 1          # no second copy of a constant outside the table
