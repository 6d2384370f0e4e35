"""-m gpu: the model classes on the HIP backend against (a) the reference's golden vectors,
(b) the oracle on ML-1M-shaped seeded inputs (sizes the oracle finishes in seconds), and
(c) size-independent properties on larger inputs.

Tolerances (BASELINE.json north_star): singular values and scores within 1e-4 relative — we hold
1e-9; top-k index lists identical on every row whose reference result is well defined (rows with
an exact k-th/(k+1)-th tie are implementation-defined in the reference)."""
import numpy as np
import pytest
import scipy.sparse as sps

from conftest import check_coffee_extras, load_golden, GoldenData
from oracle import polara_oracle as orc
from polara_amd.data import ArrayData
from polara_amd.models import SVDModel, CoffeeModel, ScaledSVD
from polara_amd.synth import make_workload, planted_csr, csr_to_numpy, csr_to_coo_triplets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['svd_warm', 'svd_known', 'svd_fewunseen', 'svd_nofilter'])
def test_svd_model_vs_reference_golden(hip_ops, name):
    g = load_golden(name)
    m = SVDModel(GoldenData(g), ops=hip_ops)
    m.verbose = False
    m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
    m.build()
    assert m.build_stats['converged']
    assert np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9, atol=0)
    V = m.factors[m.data.fields.itemid]
    assert V.flags.f_contiguous and np.abs(V @ V.T - g['V'] @ g['V'].T).max() < 1e-8
    recs = m.recommendations
    notie = g['boundary_gap'] > 0
    assert recs.dtype == np.int64 and recs.shape == g['recs'].shape
    assert np.array_equal(recs[notie], g['recs'][notie]), (recs[notie] != g['recs'][notie]).any(axis=1).sum()
    # dense score rows (slice_recommendations surface) within 1e-9
    td = (g['test_user'], g['test_item'], g['test_fdbk'])
    for u, s in zip(g['probe_users'], g['probe_scores']):
        sc, _ = m.slice_recommendations(td, tuple(int(x) for x in g['test_shape']), int(u), int(u) + 1)
        assert np.allclose(sc[0], s, rtol=1e-9, atol=1e-10)
    if name == 'svd_warm':
        m.rank = 5
        assert m._is_ready and np.array_equal(m.recommendations, g['recs_rank5'])


def test_scaled_svd_vs_reference_golden(hip_ops):
    """ScaledSVD: non-representable fp64 CSR values exercise the f64 value stream."""
    g = load_golden('svd_scaled')
    m = ScaledSVD(GoldenData(g), ops=hip_ops)
    m.verbose = False
    m.col_scaling, m.row_scaling = float(g['col_scaling']), float(g['row_scaling'])
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    m.build()
    assert np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9)
    V = m.factors[m.data.fields.itemid]
    assert np.abs(V @ V.T - g['V'] @ g['V'].T).max() < 1e-8
    notie = g['boundary_gap'] > 0
    assert np.array_equal(m.recommendations[notie], g['recs'][notie])


@pytest.mark.parametrize('name', ['coffee_small', 'coffee_warm'])
def test_coffee_model_vs_reference_golden(hip_ops, name):
    g = load_golden(name)
    m = CoffeeModel(GoldenData(g), ops=hip_ops)
    m.verbose = False
    m.mlrank, m.topk, m.seed = tuple(int(x) for x in g['mlrank']), int(g['topk']), int(g['seed'])
    m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
    m.build()
    assert len(m.core_norm_trace) == len(g['core_norm_trace'])
    assert np.allclose(m.core_norm_trace, g['core_norm_trace'], rtol=1e-9)
    f = m.data.fields
    for key, ref in ((f.userid, g['u0']), (f.itemid, g['u1']), (f.feedback, g['u2'])):
        a = m.factors[key]
        assert np.abs(a @ a.T - ref @ ref.T).max() < 1e-8
    assert np.isclose(np.linalg.norm(m.factors['core']), np.linalg.norm(g['core']), rtol=1e-9)
    notie = g['boundary_gap'] > 0
    assert np.array_equal(m.recommendations[notie], g['recs'][notie])
    check_coffee_extras(m, g)          # unfolded slices, holdout slice, predict_feedback (pk_tucker_predict_f64)


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


def test_coffee_full_feedback_mode_rank(hip_ops):
    g = load_golden('coffee_small')
    m = CoffeeModel(GoldenData(g), ops=hip_ops)
    m.verbose = False
    n_fdbk = int(m.data.to_coo(tensor_mode=True)[2][2])
    m.mlrank, m.topk, m.seed = (int(g['mlrank'][0]), int(g['mlrank'][1]), n_fdbk), int(g['topk']), int(g['seed'])
    m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
    m.build()
    _check_full_feedback_mode(m, g)


def _oracle_side(c, rank, topk, test_rows):
    A = sps.csr_matrix((c['values'].astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
    np.random.seed(0)
    _, sigma, V = orc.svd_build(A, rank)
    sub = A[test_rows]
    coo = sub.tocoo()
    order = np.lexsort((coo.col, coo.row))
    td = (coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order])
    recs, scores = orc.svd_recommendations(V, td, (len(test_rows), c['shape'][1]), topk, True, return_scores=True)
    full, sd = orc.svd_slice_recommendations(V, td, (len(test_rows), c['shape'][1]), 0, len(test_rows))
    orc.downvote_seen_items(full, sd)
    return sigma, V, recs, scores, orc.boundary_gap(full, topk)


def test_ml1m_shaped_build_and_recs_vs_oracle(hip_ops):
    """BASELINE config 0 shape (6040 x 3706, ~1M nnz, rank 10, top-10): all users scored."""
    csr, cfg = make_workload('ml1m')
    c = csr_to_numpy(csr)
    u, i, v = csr_to_coo_triplets(csr)
    n_users = c['shape'][0]
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))   # all users are test users
    d = ArrayData((u, i, v), n_users=n_users, n_items=c['shape'][1], holdout=hold, warm_start=False)
    m = SVDModel(d, ops=hip_ops)
    m.verbose = False
    m.rank, m.topk = cfg['rank'], cfg['topk']
    m.build()
    recs = m.recommendations
    sigma, V, o_recs, o_scores, gap = _oracle_side(c, cfg['rank'], cfg['topk'], np.arange(n_users))
    assert np.abs(m.factors['singular_values'] / sigma - 1).max() < 1e-9
    Vd = m.factors[d.fields.itemid]
    assert np.abs(Vd @ Vd.T - V @ V.T).max() < 1e-8
    ok = gap > 1e-9
    same = (recs[ok] == o_recs[ok]).all(axis=1)
    assert same.all(), ('rows differing', int((~same).sum()), 'of', int(ok.sum()))
    assert ok.mean() > 0.99


def test_rank50_top20_sample_vs_oracle_and_properties(hip_ops):
    """A 30k x 12k planted matrix (rank 50 / top-20): oracle parity on a 1500-user sample, plus
    properties over all users: descending exact scores, no seen item, idempotence, linearity of the
    fold-in in the profile weights."""
    from polara_amd import scoring
    c = csr_to_numpy(planted_csr(30000, 12000, 60, 50, levels=10, seed=21, min_items=15, max_items=2500))
    rank, topk = 50, 20
    A = hip_ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    from polara_amd.solver import svd_topk
    _, sigma, V, st = svd_topk(hip_ops, A, rank)
    assert st['converged']
    F = scoring.FactorImage(hip_ops, V)
    recs, sc = scoring.recommend(hip_ops, F, A, topk, True, return_scores=True)
    recs, sc = hip_ops.to_host(recs), hip_ops.to_host(sc)
    rows = np.linspace(0, c['shape'][0] - 1, 1500).astype(np.int64)
    o_sigma, o_V, o_recs, o_scores, gap = _oracle_side(c, rank, topk, rows)
    assert np.abs(hip_ops.to_host(sigma) / o_sigma - 1).max() < 1e-9
    ok = gap > 1e-9
    assert (recs[rows][ok] == o_recs[ok]).all(), int((recs[rows][ok] != o_recs[ok]).any(axis=1).sum())
    assert np.allclose(sc[rows][ok], o_scores[ok], rtol=1e-7, atol=1e-9)
    # properties on ALL users
    assert (np.diff(sc, axis=1) <= 0).all()
    seen = sps.csr_matrix((np.ones_like(c['values']), c['indices'], c['indptr']), shape=c['shape'])
    hit = seen[np.repeat(np.arange(c['shape'][0]), topk), recs.ravel()]
    assert hit.sum() == 0
    recs2 = hip_ops.to_host(scoring.recommend(hip_ops, F, A, topk, True))
    assert np.array_equal(recs, recs2)
    A2 = hip_ops.csr(c['indptr'], c['indices'], 3.0 * c['values'], c['shape'])   # scores scale, order does not
    recs3, sc3 = scoring.recommend(hip_ops, F, A2, topk, True, return_scores=True)
    assert np.array_equal(recs, hip_ops.to_host(recs3))
    assert np.allclose(hip_ops.to_host(sc3), 3.0 * sc, rtol=1e-12)


def test_build_returns_user_factors(hip_ops):
    g = load_golden('svd_warm')
    m = SVDModel(GoldenData(g), ops=hip_ops)
    m.verbose = False
    m.rank = int(g['rank'])
    m.build(return_factors=True)
    U = m.factors[m.data.fields.userid]
    V = m.factors[m.data.fields.itemid]
    s = m.factors['singular_values']
    A = orc.get_training_matrix(g['train_idx'], g['train_val'], tuple(g['train_shape']), dtype=np.float64)
    assert np.abs(U.T @ U - np.eye(len(s))).max() < 1e-9
    assert np.abs(A @ V - U * s).max() < 1e-8 * s[0]


def test_s1m_full_size_properties(hip_ops):
    """BASELINE.json configs[1] at FULL size (1M users x 100K items, 1e8 nnz, rank 50, top-10), as bench.py runs
    it: size-independent properties of the whole result + the CPU oracle on a 1 000-user sample."""
    import torch
    from polara_amd import scoring
    from polara_amd.solver import svd_topk
    from polara_amd.csr import popularity_order
    ops = hip_ops
    csr, cfg = make_workload('s1m', device=str(ops.device))
    c = csr_to_numpy(csr)
    del csr
    n_users, n_items = c['shape']
    rank, topk = cfg['rank'], cfg['topk']
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, inv_order = popularity_order(c['indices'], n_items)
    A = ops.csr_relabel_cols(A, rank_of)
    _, sigma, V, st = svd_topk(ops, A, rank)
    assert st['converged'] and st['final_rel_residual'] < 1e-12
    sig = ops.to_host(sigma)
    assert np.all(np.diff(sig) < 0) and sig[-1] > 0
    VtV = ops.to_host(ops.gram(V))
    assert np.abs(VtV - np.eye(rank)).max() < 1e-12                      # orthonormal item factors
    F = scoring.FactorImage(ops, V)
    stats = {}
    recs, sc = scoring.recommend(ops, F, A, topk, True, return_scores=True, stats=stats)
    assert recs.shape == (n_users, topk) and int(recs.min()) >= 0 and int(recs.max()) < n_items
    assert stats['flagged_users'] == 0 and stats['tiles_scored'] < 0.2 * stats['tiles_total']
    assert bool((sc[:, 1:] <= sc[:, :-1]).all())                         # descending scores
    assert bool((torch.sort(recs, dim=1).values.diff(dim=1) > 0).all())  # no duplicates in a row
    # no seen item is recommended: (user, item) keys of the CSR are ascending -> binary search
    rows = torch.repeat_interleave(torch.arange(n_users, device=recs.device), A.indptr[1:] - A.indptr[:-1])
    keys = rows * n_items + A.indices.long()
    q = (torch.arange(n_users, device=recs.device)[:, None] * n_items + recs).flatten()
    pos = torch.searchsorted(keys, q).clamp_max(keys.numel() - 1)
    assert not bool((keys[pos] == q).any())
    del rows, keys, q, pos
    # idempotence, and the pruned sweep against the full sweep on a 64K-user slice
    recs2 = scoring.recommend(ops, F, A, topk, True)
    assert bool((recs2 == recs).all())
    T = ops.csr_rows(A, 0, 65536)
    r_full = scoring.recommend(ops, F, T, topk, True, prune=False)
    assert bool((r_full == recs[:65536]).all())
    # scores are E V^T at the recommended items (fp64), checked on a slice
    E = ops.spmm(T, V)[:4096]
    want = torch.gather(E @ V.T, 1, recs[:4096])
    assert torch.allclose(want, sc[:4096], rtol=1e-12, atol=1e-12)
    # CPU oracle (reference path restated) on the first 1000 users, external item ids
    n_chk = 1000
    p1 = int(c['indptr'][n_chk])
    test_data = (np.repeat(np.arange(n_chk), np.diff(c['indptr'][:n_chk + 1])), c['indices'][:p1].astype(np.int64),
                 c['values'][:p1].astype(np.float64))
    V_ext = np.ascontiguousarray(ops.to_host(V)[rank_of])
    ref = orc.svd_recommendations(V_ext, test_data, (n_chk, n_items), topk, filter_seen=True)
    got = inv_order[ops.to_host(recs[:n_chk])]
    assert np.array_equal(got, ref)


def test_build_with_linear_operator_on_device_solver(hip_ops):
    """build(operator=...) (models.py:835-844): host LinearOperator products, device block solver."""
    from scipy.sparse.linalg import aslinearoperator, svds
    g = load_golden('svd_known')
    m = SVDModel(GoldenData(g), ops=hip_ops)
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    A = m.get_training_matrix(dtype=np.float64)
    d = 1.0 / np.sqrt(1.0 + np.asarray(A.getnnz(axis=0)).ravel())
    op = aslinearoperator(A @ sps.diags(d))
    m.build(operator=op, return_factors=True)
    u, s, vt = svds(op, k=m.rank)
    order = np.argsort(-s)
    assert np.allclose(m.factors['singular_values'], s[order], rtol=1e-9)
    V, U = m.factors[m.data.fields.itemid], m.factors[m.data.fields.userid]
    assert np.abs(V @ V.T - vt.T @ vt).max() < 1e-8 and np.abs(U @ U.T - u @ u.T).max() < 1e-8
    td, shp = (g['test_user'], g['test_item'], g['test_fdbk']), tuple(int(x) for x in g['test_shape'])
    Vref = np.ascontiguousarray(vt.T[:, order])
    want = orc.svd_recommendations(Vref, td, shp, m.topk, True)
    scores, slice_data = orc.svd_slice_recommendations(Vref, td, shp, 0, shp[0])
    orc.downvote_seen_items(scores, slice_data)
    top = -np.sort(-scores, axis=1)[:, :m.topk + 1]
    clear = (np.diff(-top, axis=1) > 1e-9 * np.abs(top[:, :1])).all(axis=1)
    assert clear.mean() > 0.5 and np.array_equal(m.recommendations[clear], want[clear])


def test_presharded_dataset_on_device(hip_ops, tmp_path):
    """On-disk CSR shards -> mapped arrays -> DeviceCSR (no COO): same factors and lists as the triplet route,
    with and without stored values, and against the oracle."""
    from polara_amd import shards
    from polara_amd.data import ShardedArrayData
    c = csr_to_numpy(planted_csr(3000, 700, mean_items=30, rank=8, seed=31, min_items=1, max_items=120))
    u = np.repeat(np.arange(c['shape'][0]), np.diff(c['indptr']))
    for vdt in (np.float32, None):
        path = str(tmp_path / ('ds_%s' % (vdt.__name__ if vdt else 'ones')))
        vals = c['values'] if vdt else np.ones(len(c['indices']), dtype=np.float32)
        shards.write_csr_shards(path, c['indptr'], c['indices'], c['values'] if vdt else None, c['shape'][1], 4, value_dtype=vdt)
        sd = ShardedArrayData.from_shards(path)
        ad = ArrayData((u, c['indices'], vals), n_users=c['shape'][0], n_items=c['shape'][1], test=(u, c['indices'], vals))
        res = []
        for data in (sd, ad):
            m = SVDModel(data, ops=hip_ops)
            m.verbose = False
            m.rank, m.topk = 12, 10
            m.build()
            res.append((m.factors['singular_values'], m.factors[data.fields.itemid], m.get_recommendations()))
        assert np.allclose(res[0][0], res[1][0], rtol=1e-12)
        assert np.abs(res[0][1] @ res[0][1].T - res[1][1] @ res[1][1].T).max() < 1e-9
        assert np.array_equal(res[0][2], res[1][2])
        A = sps.csr_matrix((vals.astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
        _, s_ref, Vt = orc.svd_build(A, 12)
        assert np.allclose(res[0][0], s_ref, rtol=1e-9)


def test_build_with_device_resident_operator_product(hip_ops):
    """build(operator=SparseProduct(L_K^T, A, L_S)) / operator=<sparse matrix>: factors on the device, the chain
    of SpMMs as the operator (the device form of hybrid/models.py:357-381), against svds of the product."""
    from scipy.sparse.linalg import svds
    from polara_amd.operator import SparseProduct
    c = csr_to_numpy(planted_csr(2500, 600, mean_items=25, rank=8, seed=41, min_items=1, max_items=100))
    u = np.repeat(np.arange(c['shape'][0]), np.diff(c['indptr']))
    data = ArrayData((u, c['indices'], c['values']), n_users=c['shape'][0], n_items=c['shape'][1],
                     test=(u, c['indices'], c['values']))
    A = sps.csr_matrix((c['values'].astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
    rng = np.random.RandomState(7)
    Ls = (sps.eye(600) + 0.3 * sps.tril(sps.random(600, 600, 0.02, random_state=rng), -1)).tocsr()
    Lk = (sps.eye(2500) + 0.3 * sps.tril(sps.random(2500, 2500, 0.004, random_state=rng), -1)).tocsr()
    full = (Lk.T @ A @ Ls).tocsr()
    uu, s, vt = svds(full, k=10)
    order = np.argsort(-s)
    recs = []
    for op in (SparseProduct(Lk.T, A, Ls), full):
        m = SVDModel(data, ops=hip_ops)
        m.verbose = False
        m.rank, m.topk = 10, 10
        m.build(operator=op, return_factors=True)
        assert m.build_stats['converged']
        assert np.allclose(m.factors['singular_values'], s[order], rtol=1e-9)
        V, U = m.factors[data.fields.itemid], m.factors[data.fields.userid]
        assert np.abs(V @ V.T - vt.T @ vt).max() < 1e-8 and np.abs(U @ U.T - uu @ uu.T).max() < 1e-8
        recs.append(m.get_recommendations())
    # same lists from both forms wherever the scores leave no doubt (the two operators differ in rounding only)
    Vref = np.ascontiguousarray(vt.T[:, order])
    scores = (A @ Vref) @ Vref.T
    scores[u, c['indices']] = -np.inf
    top = -np.sort(-scores, axis=1)[:, :11]
    clear = (np.diff(-top, axis=1) > 1e-9 * np.abs(top[:, :1])).all(axis=1)
    assert clear.mean() > 0.9 and np.array_equal(recs[0][clear], recs[1][clear])
    want = np.argsort(-scores, axis=1, kind='stable')[:, :10]
    assert np.array_equal(recs[0][clear], want[clear])


def test_captured_pass_replays_the_same_lists(hip_ops):
    """scoring.CapturedPass: the pass captured in a hipGraph returns, replay after replay, what the launched pass
    returns — including users that need the exact-row re-do (fewer than k unseen items), whose list never leaves the
    device."""
    import torch
    from polara_amd import scoring
    from polara_amd.solver import svd_topk
    ops = hip_ops
    for (n_users, n_items, mean, max_items, rank, topk) in ((6000, 900, 40, 300, 12, 10), (500, 40, 30, 39, 6, 10)):
        c = csr_to_numpy(planted_csr(n_users, n_items, mean, rank, seed=77, min_items=5, max_items=max_items))
        A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
        _, _, V, st = svd_topk(ops, A, rank)
        F = scoring.FactorImage(ops, V)
        stats = {}
        want = scoring.recommend(ops, F, A, topk, True, stats=stats)
        if n_items == 40:
            assert stats['flagged_users'] > 0            # the exact-row path is part of the captured pass
        cap = scoring.CapturedPass(ops, F, A, topk, True)
        for _ in range(3):
            got = cap.replay().clone()
            torch.cuda.synchronize()
            assert torch.equal(got, want)
        # the launched pass still works next to the graph (separate scratch buffers)
        assert torch.equal(scoring.recommend(ops, F, A, topk, True), want)
        assert torch.equal(cap.replay(), want)


def test_passes_from_several_host_threads_share_one_ops_object(hip_ops):
    """The reference parallelises its chunk loop with a thread pool (models.py:374-382): host threads driving ONE ops
    object on one stream must not interleave their passes (shared per-stream scratch state) — every thread gets the
    lists of the serial run, for different test matrices and list lengths at once."""
    import threading
    import torch
    from polara_amd import scoring
    from polara_amd.solver import svd_topk
    ops = hip_ops
    c = csr_to_numpy(planted_csr(9000, 700, 40, 10, seed=21, min_items=5, max_items=300))
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    _, _, V, st = svd_topk(ops, A, 10)
    F = scoring.FactorImage(ops, V)
    parts = [ops.csr_rows(A, lo, hi) for lo, hi in ((0, 9000), (0, 4500), (4500, 9000), (100, 8300))]
    topks = [10, 5, 20, 10]
    want = [scoring.recommend(ops, F, T, k, True).clone() for T, k in zip(parts, topks)]
    torch.cuda.synchronize()
    got = [[None] * 6 for _ in parts]
    errors = []

    def work(j):
        try:
            torch.cuda.set_device(ops.device)
            for r in range(6):
                got[j][r] = scoring.recommend(ops, F, parts[j], topks[j], True)
        except Exception as exc:      # surfaces in the main thread
            errors.append(exc)
    threads = [threading.Thread(target=work, args=(j,)) for j in range(len(parts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for j in range(len(parts)):
        for r in range(6):
            assert torch.equal(got[j][r], want[j]), (j, r)


def test_head_batch_and_device_id_renaming_change_nothing(hip_ops):
    """(1) `head_users`: the heaviest users of an activity-ordered pass as a batch of their own (item splits, second
    stream) return the lists of the single pass; (2) `HipOps.ids_to_host`: the renaming to external ids on the device
    (pk_map_ids_i64) + pinned transfer equals the NumPy renaming, -1 padding included."""
    import torch
    from polara_amd import scoring
    from polara_amd.solver import svd_topk
    ops = hip_ops
    c = csr_to_numpy(planted_csr(40000, 1500, 40, 12, seed=31, min_items=5, max_items=600))
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    _, _, V, st = svd_topk(ops, A, 12)
    F = scoring.FactorImage(ops, V)
    want = scoring.recommend(ops, F, A, 10, True, head_users=0)
    for head in (1000, 1024, 4096, 9984):     # 1000: rounded down to a 128-user boundary (the dense seen masks are per 32-user group)
        assert torch.equal(scoring.recommend(ops, F, A, 10, True, head_users=head), want), head
    # a head larger than a quarter of the users is ignored
    assert torch.equal(scoring.recommend(ops, F, A, 10, True, head_users=20000), want)
    rng = np.random.RandomState(0)
    table = rng.permutation(1500).astype(np.int64) * 7 + 3
    recs = want.clone()
    recs[::97, -2:] = -1
    host = ops.ids_to_host(recs, table)
    r = recs.cpu().numpy()
    assert host.dtype == np.int64 and np.array_equal(host, np.where(r >= 0, table[np.maximum(r, 0)], -1))
    assert np.array_equal(ops.ids_to_host(recs), r)
    table2 = table[::-1].copy()                                    # another table: the cached device copy is replaced
    assert np.array_equal(ops.ids_to_host(recs, table2), np.where(r >= 0, table2[np.maximum(r, 0)], -1))


def test_scaled_svd_device_scaling_equals_the_host_formula(hip_ops):
    """ScaledMatrixMixin on the device (pk_csr_scale_f64 after the device ingest) against its own host restatement
    (`_scale_values` on the canonical CSR): the same singular values and lists."""
    from polara_amd.models import ScaledSVD
    u, i, v = csr_to_coo_triplets(planted_csr(3000, 500, 30, 8, seed=13, min_items=5, max_items=200))
    d = ArrayData((u, i, v), n_users=3000, n_items=500, test=(u, i, v))
    m = ScaledSVD(d, ops=hip_ops)
    m.verbose = False
    m.rank, m.topk, m.col_scaling, m.row_scaling = 8, 10, 0.3, 0.8
    m.build()
    sig_dev, recs_dev = m.factors['singular_values'].copy(), m.get_recommendations().copy()
    S = m.get_training_matrix().astype(np.float64)                 # the host restatement of the scaled matrix
    s_ref = np.linalg.svd(S.toarray(), compute_uv=False)[:8]
    assert np.allclose(sig_dev, s_ref, rtol=1e-10)
    V = np.ascontiguousarray(m.factors[d.fields.itemid])
    c = csr_to_numpy(planted_csr(3000, 500, 30, 8, seed=13, min_items=5, max_items=200))
    from test_gpu_configs import _oracle_lists
    ref, clear = _oracle_lists(c, V, np.arange(3000), 10)
    assert clear.mean() > 0.9 and np.array_equal(recs_dev[clear], ref[clear])


def _planted_device_matrix(ops, n_users=30000, n_items=4000, seed=11):
    m = planted_csr(n_users, n_items, 60, 40, levels=5, seed=seed, min_items=8, max_items=400)
    c = csr_to_numpy(m)
    return ops.csr(c['indptr'], c['indices'], c['values'], c['shape']), c


def test_library_recurrence_equals_the_composition_and_arpack(hip_ops, monkeypatch):
    """Round 6: on one GPU the steps of the block Lanczos recurrence run inside the library (pk_lanczos_steps on a NON-owning
    pk_mat over the layer's CSR arrays: csrc/driver.hip::lanczos_step, the function the coarse build runs) — against the
    Python composition of the same step from single kernels (what sharded builds, host operators and the CPU double take)
    and against the reference's own call (scipy svds = ARPACK, tol 0, models.py:844): singular values to 1e-11 / 1e-9,
    projectors to 1e-9 / 1e-8, the same number of steps; narrow Krylov blocks (16 columns under a nested width of 40)."""
    from scipy.sparse.linalg import svds
    from polara_amd.ops import HipOps
    from polara_amd.solver import svd_topk
    A, c = _planted_device_matrix(hip_ops)
    k = 24
    _, s_lib, V_lib, st_lib = svd_topk(hip_ops, A, k, method='lanczos', krylov_block=16, monitor_lag=0, first_look=6)
    assert st_lib['recurrence'] == 'library' and st_lib['krylov_block'] == 16 and st_lib['block'] > 16
    assert st_lib['converged'] and st_lib['verified_rel_residual'] <= 1e-12
    monkeypatch.delattr(HipOps, 'lanczos_recurrence')
    _, s_cmp, V_cmp, st_cmp = svd_topk(hip_ops, A, k, method='lanczos', krylov_block=16, monitor_lag=0, first_look=6)
    assert st_cmp['recurrence'] == 'composition' and st_cmp['lanczos_steps'] == st_lib['lanczos_steps']
    s_lib, s_cmp = hip_ops.to_host(s_lib), hip_ops.to_host(s_cmp)
    V_lib, V_cmp = hip_ops.to_host(V_lib), hip_ops.to_host(V_cmp)
    assert np.allclose(s_lib, s_cmp, rtol=1e-11) and np.abs(V_lib @ V_lib[:300].T - V_cmp @ V_cmp[:300].T).max() < 1e-9
    M = sps.csr_matrix((c['values'].astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
    np.random.seed(0)
    _, s_ref, vt = svds(M, k=k, tol=0)
    assert np.allclose(np.sort(s_ref)[::-1], s_lib, rtol=1e-9) and np.abs(vt.T @ vt[:, :300] - V_lib @ V_lib[:300].T).max() < 1e-8


def test_rounded_late_products_give_the_factors_of_the_exact_build(hip_ops):
    """svd_topk(products='relaxed'): once a look says the pairs are within 1e-7 of convergence, both sparse products of a
    step gather fp32 images of their dense blocks and only the BAND of the block column of T is kept (the mirror image of the
    rounding noise would meet the O(1) early coefficients: what stalled round 4's rounded products at 5e-12) — the pairs are
    accepted on a TRUE fp64 residual below 1e-12 like those of the exact build, and the factors agree to 1e-11 / 1e-9."""
    from polara_amd.solver import svd_topk
    A, _ = _planted_device_matrix(hip_ops, seed=12)
    k = 24
    kw = dict(method='lanczos', krylov_block=16, monitor_lag=1, first_look=4)      # a look per step: the gate opens while steps remain
    _, s0, V0, st0 = svd_topk(hip_ops, A, k, **kw)
    _, s1, V1, st1 = svd_topk(hip_ops, A, k, products='relaxed', **kw)
    assert st0.get('products_rounded_from') is None and st1.get('products_rounded_from') is not None
    assert st1['products_rounded_from'] < st1['lanczos_steps'], st1          # rounded steps did run
    assert st1['converged'] and st1['verified_rel_residual'] <= 1e-12 and 'exchange_relaxed_failed_at' not in st1
    s0, s1, V0, V1 = (hip_ops.to_host(t) for t in (s0, s1, V0, V1))
    assert np.allclose(s0, s1, rtol=1e-11) and np.abs(V0 @ V0[:300].T - V1 @ V1[:300].T).max() < 1e-9


def test_factor_image_without_torch_kernels_is_the_image_of_round_5(hip_ops):
    """ops.v32_image (pk_v32_image_f32: the fp32 image of the item factors, the row-norm bounds in column K, the range check's
    two numbers in one launch) against the fills / casts / index writes it replaced, and the serving order on the device
    (pk_row_norm_order_f64) inside a FactorImage round trip; non-finite factors are refused as before."""
    import torch
    from polara_amd import scoring
    rng = np.random.RandomState(5)
    V = rng.standard_normal((3001, 50)) * np.exp(-2.0 * rng.rand(3001))[:, None]
    Vd = hip_ops.to_device(V)
    F = scoring.FactorImage(hip_ops, Vd)
    want = torch.zeros(3001, F.V32x.stride(0), dtype=torch.float32, device=Vd.device)
    want[:, :50] = Vd.to(torch.float32)
    want[:, 50] = F.vnorm
    assert torch.equal(F.V32x, want[:, :F.Kx])
    true_max = float(np.linalg.norm(V, axis=1).max())
    assert true_max <= F.vmax <= true_max * (1 + 1e-6)
    bad = V.copy(); bad[17, 3] = np.inf
    with pytest.raises(ValueError):
        scoring.FactorImage(hip_ops, hip_ops.to_device(bad))
    bad[17, 3] = np.nan
    with pytest.raises(ValueError):
        scoring.FactorImage(hip_ops, hip_ops.to_device(bad))


def test_library_recurrence_hands_hard_matrices_to_the_subspace_iteration(hip_ops):
    """The breakdown paths of the block Lanczos build with the steps inside the library: an exactly rank-deficient matrix (the
    residual block loses rank: Cholesky verdicts / orthonormality flags on the device, read at the first look), a matrix with
    fewer items than four Krylov blocks, and clustered singular values (tripled columns) — every one ends with the factors of
    the dense SVD, through the fall-back where the recurrence cannot finish (`stats['lanczos_fallback']`), never with an error
    out of a launcher."""
    from polara_amd.solver import svd_topk
    rng = np.random.RandomState(3)
    cases = []
    B = rng.standard_normal((400, 5)) @ rng.standard_normal((5, 90))                      # rank 5 exactly, 9 vectors asked for
    cases.append(('rank5_ask9', sps.csr_matrix(B), 9, True))
    cases.append(('tiny_40_items', sps.random(300, 40, density=0.3, random_state=1, format='csr'), 6, False))     # five blocks of 8 fill the whole space: the recurrence may finish
    M = sps.random(3000, 120, density=0.08, random_state=2, format='csr')
    cases.append(('tripled_columns', sps.hstack([M, M, M]).tocsr(), 20, False))
    for name, M, k, must_fall_back in cases:
        M = M.astype(np.float64)
        A = hip_ops.csr(M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, M.shape)
        _, s, V, st = svd_topk(hip_ops, A, k, method='lanczos', krylov_block=8)
        s_ref = np.linalg.svd(M.toarray(), compute_uv=False)[:k]
        s = hip_ops.to_host(s)
        nz = s_ref > 1e-8 * s_ref[0]
        assert st['converged'], (name, st)
        assert np.allclose(s[nz], s_ref[nz], rtol=1e-9), (name, np.abs(s[nz] / s_ref[nz] - 1).max())
        Vh = hip_ops.to_host(V)
        assert np.abs(Vh.T @ Vh - np.eye(k)).max() < 1e-8, name
        if must_fall_back:
            assert 'lanczos_fallback' in st and st['method'].startswith('subspace'), (name, st.get('method'))


@pytest.mark.gpu
def test_recorded_pass_replays_the_pass_call_by_call():
    """scoring.RecordedPass: the library calls of one pass recorded once and issued again give, replay after replay and on
    two streams in turn, the lists of the launched pass — pruned and unpruned, a user set small enough for item splits, and
    a second recording on another stream does not disturb the first."""
    import torch
    from polara_amd import scoring
    from polara_amd.ops import HipOps
    from polara_amd.solver import svd_topk
    ops = HipOps('cuda:0')
    for (n_users, n_items, per, topk, prune) in ((6000, 3000, 40, 10, True), (900, 5000, 25, 20, True), (3000, 2000, 30, 10, False)):
        csr = planted_csr(n_users, n_items, per, rank=16, seed=n_users)
        A = ops.csr(np.asarray(csr['indptr']), np.asarray(csr['indices']), np.asarray(csr['values']), csr['shape'])
        _, sigma, V, _ = svd_topk(ops, A, 12, seed=1)
        order, rank, Vs = ops.norm_order(V)
        T = ops.csr_relabel_cols(A, rank, sort=True)
        F = scoring.FactorImage(ops, Vs)
        want = scoring.recommend(ops, F, T, topk, True, prune=prune).clone()
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        rp0 = scoring.RecordedPass(ops, F, T, topk, True, prune=prune)
        side.wait_stream(main)
        pinned = torch.empty(tuple(want.shape), dtype=torch.int64).pin_memory()
        with torch.cuda.stream(side):
            rp1 = scoring.RecordedPass(ops, F, T, topk, True, prune=prune, host_out=pinned)     # ... handing its lists to the host itself
        assert rp0.stream != rp1.stream and len(rp1.calls) == len(rp0.calls) + 1 >= 5 and rp1.calls[-1][0] == 'pk_copy_to_host_async'
        assert all(name.startswith('pk_') for name, _, _ in rp0.calls)
        for i in range(6):
            rp = (rp0, rp1)[i & 1]
            out = rp.replay()
            torch.cuda.synchronize()
            if rp is rp1:
                assert out is pinned and torch.equal(pinned, want.cpu()), (n_users, i)      # handed over by the recording's last call
                pinned.zero_()
            else:
                assert torch.equal(out, want), (n_users, i)
        # a launched pass in between uses the same per-stream scratch and leaves the recording intact
        assert torch.equal(scoring.recommend(ops, F, T, topk, True, prune=prune), want)
        assert torch.equal(rp0.replay(), want)
        with pytest.raises(ValueError):
            scoring.recommend(ops, F, T, topk, True, prune=prune, out=torch.empty(tuple(want.shape), dtype=torch.int64))   # a host destination must be pinned
        pinned.zero_()
        rp2 = scoring.RecordedPass(ops, F, T, topk, True, prune=prune, host_out=pinned, hand_over='mapped')    # the last KERNEL writes the host array
        assert len(rp2.calls) <= len(rp0.calls) + 1 and rp2.calls[-1][0] == 'pk_scatter_rows_i64' and rp2.replay() is pinned
        torch.cuda.synchronize()
        assert torch.equal(pinned, want.cpu())
        dev_out = torch.empty_like(want)
        assert scoring.recommend(ops, F, T, topk, True, prune=prune, out=dev_out) is dev_out and torch.equal(dev_out, want)
