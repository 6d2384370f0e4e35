"""-m gpu: every HIP kernel, called through the C ABI (polara_amd.ops.HipOps -> libpolarahip.so),
against NumPy/SciPy on the same seeded inputs.  Integer/index results must be bit-exact; fp64
kernels are held to 1e-12 relative (summation order differs from BLAS), the fp32 MFMA candidate
pass to the certified error bound it reports itself."""
import numpy as np
import pytest
import scipy.sparse as sps
import torch

from numpy_ops import NumpyOps
from oracle import polara_oracle as orc

pytestmark = pytest.mark.gpu


def rand_csr(rng, n_rows, n_cols, mean, long_rows=(), empty_rows=(), dtype=np.float32, levels=5):
    counts = rng.poisson(mean, n_rows).clip(0, n_cols)
    for r, c in long_rows:
        counts[r] = min(c, n_cols)
    for r in empty_rows:
        counts[r] = 0
    indptr = np.r_[0, np.cumsum(counts)].astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, c, replace=False)) for c in counts] or [[]]).astype(np.int32)
    if dtype == np.float32:
        values = rng.randint(1, levels + 1, indptr[-1]).astype(np.float32)
    else:
        values = rng.randn(indptr[-1])
    return indptr, indices, values


def test_library_reports_gfx950(hip_ops):
    from polara_amd import _lib
    info = _lib.device_info(0)
    assert info['n_devices'] >= 1 and 'gfx950' in info['arch'], info


@pytest.mark.parametrize('nc', [1, 10, 50, 64, 72, 130, 256])
@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
def test_spmm_matches_scipy(hip_ops, nc, vdtype):
    rng = np.random.RandomState(nc)
    n_rows, n_cols = 3000, 1500
    indptr, indices, values = rand_csr(rng, n_rows, n_cols, 25, long_rows=[(5, 1400), (17, 1100), (2999, 1300)],
                                       empty_rows=[0, 7, 2998], dtype=vdtype)
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols), split=256)
    assert A.n_long >= 3
    X = rng.randn(n_cols, nc)
    out = hip_ops.to_host(hip_ops.spmm(A, hip_ops.to_device(X)))
    ref = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols)) @ X
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-13, err
    assert (out[[0, 7, 2998]] == 0).all()
    # transpose product through the CSC plan
    Y = rng.randn(n_rows, min(nc, 64))
    outT = hip_ops.to_host(hip_ops.spmm(A.T, hip_ops.to_device(Y)))
    refT = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols)).T @ Y
    assert np.abs(outT - refT).max() / np.abs(refT).max() < 1e-13
    # strided X (leading dimension > nc) is honoured
    Xw = hip_ops.to_device(rng.randn(n_cols, nc + 5))
    out2 = hip_ops.to_host(hip_ops.spmm(A, Xw[:, :nc]))
    ref2 = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols)) @ hip_ops.to_host(Xw)[:, :nc]
    assert np.abs(out2 - ref2).max() / np.abs(ref2).max() < 1e-13


@pytest.mark.parametrize('nc', [4, 52, 64, 104, 204])
def test_spmm_fp32_dense_block(hip_ops, nc):
    """pk_spmm_csr_x with an fp32 dense block (the approximate fold-in): fp64 accumulation of fl32(X)."""
    import torch
    rng = np.random.RandomState(nc)
    n_rows, n_cols = 700, 900
    indptr, indices, values = rand_csr(rng, n_rows, n_cols, 30, long_rows=[(3, 850)], empty_rows=[0, 11], dtype=np.float32)
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols))
    X32 = rng.randn(n_cols, nc).astype(np.float32)
    got = hip_ops.to_host(hip_ops.spmm(A, hip_ops.to_device(X32)))
    ref = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols)) @ X32.astype(np.float64)
    assert got.dtype == np.float64 and np.allclose(got, ref, rtol=1e-13, atol=1e-13)
    # a wider, strided output (the fold-in writes K of Kx columns of a row)
    out = torch.zeros(n_rows, nc + 4, dtype=torch.float64, device=hip_ops.device)
    hip_ops.spmm(A, hip_ops.to_device(X32), out=out[:, :nc])
    assert np.allclose(hip_ops.to_host(out)[:, :nc], ref, rtol=1e-13, atol=1e-13) and float(out[:, nc:].abs().sum()) == 0.0


def test_device_coo_to_csr_matches_scipy(hip_ops):
    rng = np.random.RandomState(4)
    n_rows, n_cols, nnz = 5000, 3000, 400000
    r, c = rng.randint(0, n_rows, nnz), rng.randint(0, n_cols, nnz)
    v = rng.randint(1, 6, nnz).astype(np.float64)
    v[::11] = 0.0                                   # explicit zeros survive (they still mean "seen")
    A = hip_ops.csr_from_coo(r, c, v, (n_rows, n_cols))
    ref = sps.coo_matrix((v, (r, c)), shape=(n_rows, n_cols)).tocsr()   # sums duplicates, sorts
    key = r.astype(np.int64) * n_cols + c
    assert A.nnz == len(np.unique(key)) >= ref.nnz
    ours = sps.csr_matrix((hip_ops.to_host(A.values).astype(np.float64), hip_ops.to_host(A.indices),
                           hip_ops.to_host(A.indptr)), shape=(n_rows, n_cols))
    chk = ours.copy()
    chk.sort_indices()
    assert np.array_equal(chk.indices, ours.indices)      # canonical: column-sorted within rows
    assert abs(ours - ref).max() == 0
    sub = hip_ops.csr_rows(A, 100, 1100)
    X = rng.randn(n_cols, 8)
    assert np.allclose(hip_ops.to_host(hip_ops.spmm(sub, hip_ops.to_device(X))), ref[100:1100] @ X, rtol=1e-13)
    with pytest.raises(ValueError):
        hip_ops.csr_from_coo([0, n_rows], [0, 1], [1.0, 1.0], (n_rows, n_cols))


def test_spmm_is_deterministic(hip_ops):
    rng = np.random.RandomState(3)
    indptr, indices, values = rand_csr(rng, 2000, 900, 40, long_rows=[(3, 880)])
    A = hip_ops.csr(indptr, indices, values, (2000, 900), split=128)
    X = hip_ops.to_device(rng.randn(900, 64))
    a = hip_ops.to_host(hip_ops.spmm(A, X))
    b = hip_ops.to_host(hip_ops.spmm(A, X))
    assert np.array_equal(a, b)


@pytest.mark.parametrize('shape', [(5000, 64, 64), (777, 10, 72), (40000, 130, 130), (33, 5, 3), (100000, 24, 24)])
def test_gram_and_tsmm(hip_ops, shape):
    n, la, lb = shape
    rng = np.random.RandomState(n)
    A, B = rng.randn(n, la), rng.randn(n, lb)
    G = hip_ops.to_host(hip_ops.gram(hip_ops.to_device(A), hip_ops.to_device(B)))
    ref = A.T @ B
    assert np.abs(G - ref).max() / np.abs(ref).max() < 1e-13
    Gs = hip_ops.to_host(hip_ops.gram(hip_ops.to_device(A)))
    assert np.abs(Gs - A.T @ A).max() / np.abs(A.T @ A).max() < 1e-13
    Cm = rng.randn(la, lb)
    out = hip_ops.to_host(hip_ops.tsmm(hip_ops.to_device(A), hip_ops.to_device(Cm)))
    assert np.abs(out - A @ Cm).max() / np.abs(A @ Cm).max() < 1e-13


@pytest.mark.parametrize('shape', [(26744, 896, 64), (5001, 130, 50), (640, 640, 64), (900, 64, 64), (333, 7, 9), (70000, 256, 128)])
def test_wide_gram_strided_operands_and_fused_projection(hip_ops, shape):
    """The products of the block Lanczos build: the whole Krylov basis (a column slice of a wider buffer: leading
    dimension > width) against one block, the fused Z - X C (pk_tsmm_sub_f64, in place), and the small square products of
    the projected problem."""
    n, la, lb = shape
    rng = np.random.RandomState(la + lb)
    buf = hip_ops.to_device(rng.randn(n, la + 64))
    A = buf[:, :la]                                   # strided view
    B = hip_ops.to_device(rng.randn(n, lb))
    An, Bn = hip_ops.to_host(A), hip_ops.to_host(B)
    G = hip_ops.gram(A, B)
    ref = An.T @ Bn
    assert np.abs(hip_ops.to_host(G) - ref).max() / np.abs(ref).max() < 1e-13
    assert np.array_equal(hip_ops.to_host(hip_ops.gram(A, B)), hip_ops.to_host(G))        # deterministic
    Cm = hip_ops.to_device(rng.randn(la, lb))
    want = Bn - An @ hip_ops.to_host(Cm)
    out = hip_ops.tsmm_sub(B, A, Cm)
    assert np.abs(hip_ops.to_host(out) - want).max() / np.abs(want).max() < 1e-13
    Z = B.clone()
    hip_ops.tsmm_sub(Z, A, Cm, out=Z)                 # in place
    assert np.array_equal(hip_ops.to_host(Z), hip_ops.to_host(out))
    assert np.abs(hip_ops.to_host(hip_ops.tsmm(A, Cm)) - An @ hip_ops.to_host(Cm)).max() / np.abs(An @ hip_ops.to_host(Cm)).max() < 1e-13


@pytest.mark.parametrize('case', [(500, 20, 32, 'graded'), (896, 50, 64, 'planted'), (130, 10, 24, 'flat'), (64, 8, 16, 'graded'),
                                  (300, 12, 24, 'rank40')])
def test_sym_eig_topk_against_lapack(hip_ops, case):
    """pk_sym_eig_topk_f64 (the projected problem of the block Lanczos build, solved from C++): the k leading pairs of a
    dense PSD matrix against numpy's eigh — eigenvalues to 1e-12 of the largest, residuals below the tolerance asked for,
    orthonormal vectors; a warm start from the pairs of the leading principal submatrix; same bits on a second call."""
    n, k, l, kind = case
    rng = np.random.RandomState(n + k)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    if kind == 'graded':
        lam = np.exp(-np.arange(n) / 12.0) * 1e6
    elif kind == 'planted':
        lam = np.r_[1e6 / (1 + np.arange(100)), 3e3 * rng.rand(n - 100)]
    elif kind == 'flat':
        lam = 1.0 + rng.rand(n)
    else:
        lam = np.r_[np.linspace(5, 1, 40), np.zeros(n - 40)]
    T = (Q * lam) @ Q.T
    T = 0.5 * (T + T.T)
    w = np.sort(np.linalg.eigvalsh(T))[::-1]
    Td = hip_ops.to_device(T)
    tol = 1e-13
    basis, lam_all, res, n_lock, conv = hip_ops.sym_eig_topk(Td, k, hip_ops.to_device(np.eye(n, l)), tol)
    assert conv and basis.shape == (n, l) and len(lam_all) == l
    Y = hip_ops.to_host(basis)[:, :k]
    assert np.abs(lam_all[:k] - w[:k]).max() <= 1e-12 * w[0]
    assert np.abs(Y.T @ Y - np.eye(k)).max() < 1e-10
    assert np.linalg.norm(T @ Y - Y * lam_all[:k], axis=0).max() <= 4 * tol * w[0]
    b2, l2, _, _, _ = hip_ops.sym_eig_topk(Td, k, hip_ops.to_device(np.eye(n, l)), tol)
    assert np.array_equal(hip_ops.to_host(b2), hip_ops.to_host(basis)) and np.array_equal(l2, lam_all)
    # warm start: the pairs of the leading principal submatrix, padded with zero rows by the callee
    m = n - max(8, n // 8)
    bs, ls, _, _, cs = hip_ops.sym_eig_topk(hip_ops.to_device(np.ascontiguousarray(T[:m, :m])), k, hip_ops.to_device(np.eye(m, l)), tol)
    st = {}
    bw, lw, _, _, cw = hip_ops.sym_eig_topk(Td, k, bs, tol, stats=st)
    assert cs and cw and np.abs(lw[:k] - w[:k]).max() <= 1e-12 * w[0] and st['outer'] >= 1 and st['steps'] >= 1


@pytest.mark.parametrize('n', [1, 2, 5, 24, 63, 64, 72, 128, 136, 137, 150, 200, 256, 301, 520])
def test_eigh_psd_jacobi(hip_ops, n):
    rng = np.random.RandomState(n)
    M = rng.randn(n + 3, n) * np.logspace(0, -6, n)[None, :]   # badly scaled Gram matrix
    S = M.T @ M
    lam, C = hip_ops.eigh_psd(hip_ops.to_device(S))
    lam, C = hip_ops.to_host(lam), hip_ops.to_host(C)
    ref = np.linalg.eigvalsh(S)[::-1]
    assert (np.diff(lam) <= 1e-300 + 1e-14 * lam[0]).all()
    assert np.abs(lam - ref).max() <= 1e-12 * ref[0]
    assert np.abs(C.T @ C - np.eye(n)).max() < 1e-12
    assert np.abs(S @ C - C * lam[None, :]).max() <= 1e-11 * ref[0]
    info = hip_ops.to_host(hip_ops._info)
    assert info[1] == 1, 'jacobi did not converge: %s' % info
    # small eigenvalues keep RELATIVE accuracy (what the whitening step relies on)
    keep = ref > 1e-10 * ref[0]
    assert np.abs(lam[keep] / ref[keep] - 1).max() < 1e-6


@pytest.mark.parametrize('n', [8, 70, 130, 150])
def test_eigh_exactly_singular_inputs(hip_ops, n):
    """Rows that vanish exactly are completed to an orthonormal basis (no rotation matrix is kept)."""
    Z = np.zeros((n, n))
    lam, C = hip_ops.eigh_psd(hip_ops.to_device(Z))
    C = hip_ops.to_host(C)
    assert (hip_ops.to_host(lam) == 0).all() and np.abs(C.T @ C - np.eye(n)).max() < 1e-12
    S = np.zeros((n, n))
    S[:3, :3] = np.array([[4., 1, 0], [1, 3, 0], [0, 0, 0]])     # exact zero rows/cols + a zero eigenvalue
    lam, C = hip_ops.eigh_psd(hip_ops.to_device(S))
    lam, C = hip_ops.to_host(lam), hip_ops.to_host(C)
    assert np.allclose(lam[:2], np.linalg.eigvalsh(S)[::-1][:2]) and (lam[2:] == 0).all()
    assert np.abs(C.T @ C - np.eye(n)).max() < 1e-12 and np.abs(S @ C - C * lam).max() < 1e-12
    if n > 136:
        # a dense rank-one input leaves n - 1 rows of pure rounding noise that every rotation against the one real
        # row refreshes: the LDS kernel happens to zero them exactly, the block variant need not — the solver never
        # relies on it (its rank-deficient blocks go through `_refill`, which drops directions below 1e-10)
        return
    v = np.arange(1, n + 1.0)
    S1 = np.outer(v, v)                                           # rank one, dense
    lam, C = hip_ops.eigh_psd(hip_ops.to_device(S1))
    lam, C = hip_ops.to_host(lam), hip_ops.to_host(C)
    assert np.isclose(lam[0], v @ v) and np.abs(lam[1:]).max() < 1e-9 * lam[0]
    assert np.abs(C.T @ C - np.eye(n)).max() < 1e-10


@pytest.mark.parametrize('n', [1, 5, 64, 128, 136, 200])
def test_chol_rinv(hip_ops, n):
    rng = np.random.RandomState(n)
    X = rng.randn(4 * n + 3, n) * np.exp(rng.randn(n) * 2)          # graded columns: cond(G) ~ 1e7
    G = X.T @ X
    Rinv, info = hip_ops.chol_rinv(hip_ops.to_device(G))
    Rinv = hip_ops.to_host(Rinv)
    assert int(hip_ops.to_host(info)[0]) == 0
    assert np.allclose(Rinv, np.triu(Rinv))
    Q = X @ Rinv
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 1e-8
    R = np.linalg.cholesky(G).T
    assert np.allclose(Rinv, np.linalg.inv(R), rtol=1e-7, atol=1e-9 * np.abs(np.linalg.inv(R)).max())
    # relative shift: G + s*trace(G)*I
    Rs, _ = hip_ops.chol_rinv(hip_ops.to_device(G), 1e-3)
    Rs = hip_ops.to_host(Rs)
    Gs = G + 1e-3 * np.trace(G) * np.eye(n)
    assert np.abs(Rs.T @ Gs @ Rs - np.eye(n)).max() < 1e-8
    if n >= 5:   # exactly singular input: reported, not silently wrong
        Xd = X.copy()
        Xd[:, -1] = Xd[:, 0]
        Gd = Xd.T @ Xd
        Gd[-1, -1] = Gd[0, 0] * (1 - 1e-12)
        _, info = hip_ops.chol_rinv(hip_ops.to_device(Gd))
        assert int(hip_ops.to_host(info)[0]) != 0


@pytest.mark.parametrize('n', [5, 16, 64, 128, 200])
def test_chol_rinv_of_the_column_scaled_gram_matrix(hip_ops, n):
    """pk_chol_rinv_scaled_f64: Rinv = D R'^-1 from the Cholesky factor of D G D (D = diag(G)^-1/2): X Rinv is orthonormal
    like the unscaled form's — and stays so on a block whose columns differ by twelve orders of magnitude and are otherwise
    well conditioned (filtered Ritz vectors), where the unscaled, unshifted factorisation breaks down or loses everything."""
    import ctypes as C
    import torch
    from polara_amd import _lib
    from polara_amd.ops import _ptr
    rng = np.random.RandomState(n)
    Q, _ = np.linalg.qr(rng.randn(4 * n + 3, n))
    X = (Q + 0.05 * rng.randn(4 * n + 3, n)) * np.exp(rng.uniform(-14, 14, n))       # cond(X) ~ 1e12, cond of the scaled block ~ 2
    G = hip_ops.to_device(X.T @ X)
    Rinv = hip_ops.empty(n, n)
    info = torch.zeros(1, dtype=torch.int32, device=G.device)
    need = hip_ops.lib.pk_chol_work_bytes(n)
    work = hip_ops.empty((need + 7) // 8) if need else None
    _lib.check(hip_ops.lib.pk_chol_rinv_scaled_f64(hip_ops.stream(), n, _ptr(G), n, 0.0, _ptr(Rinv), n, _ptr(work), _ptr(info)),
               'pk_chol_rinv_scaled_f64')
    assert int(hip_ops.to_host(info)[0]) == 0
    R = hip_ops.to_host(Rinv)
    assert np.allclose(R, np.triu(R))
    Y = X @ R
    assert np.abs(Y.T @ Y - np.eye(n)).max() < 1e-9
    # a zero column is reported
    Xz = X.copy(); Xz[:, n // 2] = 0.0
    Gz = hip_ops.to_device(Xz.T @ Xz)
    _lib.check(hip_ops.lib.pk_chol_rinv_scaled_f64(hip_ops.stream(), n, _ptr(Gz), n, 0.0, _ptr(Rinv), n, _ptr(work), _ptr(info)),
               'pk_chol_rinv_scaled_f64')
    assert int(hip_ops.to_host(info)[0]) != 0


def test_elementwise_and_small_kernels(hip_ops):
    rng = np.random.RandomState(0)
    for n in (1, 7, 1000, 100001):
        Z, Y, X = rng.randn(n), rng.randn(n), rng.randn(n)
        dz, dy, dx = (hip_ops.to_device(a) for a in (Z, Y, X))
        assert np.allclose(hip_ops.to_host(hip_ops.axpbypcz(0.3, dz, -1.2, dy, 2.5, dx)), 0.3 * Z - 1.2 * Y + 2.5 * X, rtol=1e-14, atol=1e-14)
        assert np.allclose(hip_ops.to_host(hip_ops.axpbypcz(0.3, dz, -1.2, dy)), 0.3 * Z - 1.2 * Y, rtol=1e-14, atol=1e-14)
        assert np.allclose(hip_ops.to_host(hip_ops.axpbypcz(2.0, dz)), 2.0 * Z)
    for n, l in ((5000, 64), (1025, 7), (300, 200), (3, 300)):
        Z, X, th = rng.randn(n, l), rng.randn(n, l), rng.rand(l)
        r = hip_ops.to_host(hip_ops.resid_colnorm2(hip_ops.to_device(Z), hip_ops.to_device(X), hip_ops.to_device(th)))
        ref = ((Z - X * th) ** 2).sum(0)
        assert np.allclose(r, ref, rtol=1e-12)
        Xs = hip_ops.to_device(X.copy())
        hip_ops.scale_cols(Xs, hip_ops.to_device(th))
        assert np.allclose(hip_ops.to_host(Xs), X * th)
    A, B = rng.randn(5, 900), rng.randn(5, 900)
    assert np.allclose(hip_ops.to_host(hip_ops.small_mm(hip_ops.to_device(A), hip_ops.to_device(B), transB=True)), A @ B.T)
    assert np.allclose(hip_ops.to_host(hip_ops.small_mm(hip_ops.to_device(A), hip_ops.to_device(B), transA=True)), A.T @ B)


def brute_topk(V, E, seen_ptr, seen_idx, topk, filter_seen=True):
    s = E @ V.T
    n_users, n_items = s.shape
    out = np.empty((n_users, topk), dtype=np.int64)
    for u in range(n_users):
        cls = np.zeros(n_items, dtype=np.int64)
        if filter_seen:
            cls[seen_idx[seen_ptr[u]:seen_ptr[u + 1]]] = 1
        out[u] = np.lexsort((np.arange(n_items), -s[u], cls))[:topk]
    return out, s


@pytest.mark.parametrize('cfg', [dict(n_users=300, n_items=1000, K=10, topk=10),
                                 dict(n_users=97, n_items=4099, K=50, topk=10),
                                 dict(n_users=260, n_items=2500, K=100, topk=20),
                                 dict(n_users=64, n_items=777, K=200, topk=50),
                                 dict(n_users=33, n_items=31, K=7, topk=5),
                                 dict(n_users=70, n_items=640, K=24, topk=1)])
def test_fused_scoring_pipeline_exact(hip_ops, cfg):
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(K)
    V = np.linalg.qr(rng.randn(n_items, K))[0] if n_items >= K else rng.randn(n_items, K)
    heavy = [(1, min(n_items - 2, 900))] if n_items > 100 else []
    indptr, indices, values = rand_csr(rng, n_users, n_items, min(30, n_items // 3), long_rows=heavy,
                                       empty_rows=[0], dtype=np.float32)
    values[::7] = 0.0   # explicit zero-feedback entries stay 'seen'
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    E = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_users, n_items)) @ V
    for filter_seen in (True, False):
        stats = {}
        recs, sc = scoring.recommend(hip_ops, F, T, topk, filter_seen, return_scores=True, stats=stats)
        recs, sc = hip_ops.to_host(recs), hip_ops.to_host(sc)
        ref, s = brute_topk(V, E, indptr, indices, topk, filter_seen)
        # rows with an all-zero profile are pure ties (implementation-defined in the reference too)
        live = np.abs(E).sum(1) > 0
        gap_ok = np.ones(n_users, bool)
        for u in range(n_users):   # exclude rows whose k-th/(k+1)-th exact scores tie within rounding
            ss = np.sort(s[u][np.setdiff1d(np.arange(n_items), indices[indptr[u]:indptr[u + 1]] if filter_seen else [])])[::-1]
            if len(ss) > topk:
                gap_ok[u] = (ss[topk - 1] - ss[topk]) > 1e-12 * max(1.0, abs(ss[0]))
        chk = live & gap_ok
        bad = np.flatnonzero(~orc.topk_sets_equal(recs[chk], ref[chk]))
        assert bad.size == 0, (cfg, filter_seen, 'first bad rows', bad[:5], recs[chk][bad[:2]], ref[chk][bad[:2]], stats)
        assert (recs[chk] == ref[chk]).mean() > 0.999   # ordered equality (up to exact ties)
        picked = np.take_along_axis(s, np.where(recs >= 0, recs, 0), axis=1)
        assert np.allclose(sc[chk], picked[chk], rtol=1e-12, atol=1e-13)
        if filter_seen:   # no seen item may appear while unseen ones remain
            for u in np.flatnonzero(chk):
                seen = set(indices[indptr[u]:indptr[u + 1]])
                if n_items - len(seen) >= topk:
                    assert not (set(recs[u]) & seen), (u, cfg)


@pytest.mark.parametrize('shape', [(50, 10), (200, 50), (160, 24)], ids=lambda s: 'K%d_top%d' % s)
@pytest.mark.parametrize('tiles_per_chunk', [1, 3, 17])
def test_chunked_item_sweep_equals_single_sweep(hip_ops, tiles_per_chunk, shape):
    """The per-user selection state parked between item-chunk launches must make the result independent of the chunking —
    also for the rank > 128 / 64-candidate instance, whose rings hold 32 entries per lane while the parked image holds 16
    (longer rings are merged before parking), and whatever the automatic choice is (round 5: a pruned sweep is ONE launch,
    a full sweep L2-sized chunks)."""
    from polara_amd import scoring
    K, topk = shape
    rng = np.random.RandomState(11 + K)
    n_users, n_items = 150, 3000
    V = np.linalg.qr(rng.randn(n_items, K))[0] * ((1.0 + np.arange(n_items)) ** -0.3)[:, None]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, 2500)], empty_rows=[5])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    hip_ops.score_tiles_per_chunk = 10 ** 6
    try:
        ref, ref_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, two_phase_ok=False)
        for prune in (True, False):
            hip_ops.score_tiles_per_chunk = tiles_per_chunk
            got, got_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, prune=prune, two_phase_ok=False)
            hip_ops.score_tiles_per_chunk = 0
            auto, auto_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, prune=prune, two_phase_ok=False)
            for a, b in ((ref, got), (ref_s, got_s), (ref, auto), (ref_s, auto_s)):
                assert np.array_equal(hip_ops.to_host(a), hip_ops.to_host(b)), prune
    finally:
        hip_ops.score_tiles_per_chunk = 0
    # the launch count behind the automatic choice
    n_tiles = -(-n_items // 32)
    assert hip_ops.lib.pk_score_chunk_launches(n_items, K, 1, 0, 1) == 1
    assert hip_ops.lib.pk_score_chunk_launches(n_items, K, 1, 0, 0) >= 1
    assert hip_ops.lib.pk_score_chunk_launches(10 ** 6, 50, 1, 0, 0) > 1 and hip_ops.lib.pk_score_chunk_launches(10 ** 6, 50, 1, 0, 1) == 1
    assert hip_ops.lib.pk_score_chunk_launches(n_items, K, 1, 7, 0) == -(-n_tiles // 7)


def test_seen_tile_stream(hip_ops):
    """pk_seen_tiles_build: one (tile << 32 | mask) record per touched tile, in order, for short, empty,
    multi-chunk (> 64 entries) and dense rows."""
    rng = np.random.RandomState(5)
    n_users, n_items = 300, 5000
    indptr, indices, values = rand_csr(rng, n_users, n_items, 30, long_rows=[(2, 4000), (9, 64), (10, 65), (11, 129)],
                                       empty_rows=[0, 7])
    sp, si = hip_ops.to_device(indptr.astype(np.int64)), hip_ops.to_device(indices.astype(np.int32))
    tiles, ntiles = hip_ops.seen_tiles(sp, si, n_users)
    tiles, ntiles = hip_ops.to_host(tiles).view(np.uint64), hip_ops.to_host(ntiles)
    for u in range(n_users):
        row = indices[indptr[u]:indptr[u + 1]].astype(np.int64)
        want = {}
        for j in row:
            want[j >> 5] = want.get(j >> 5, 0) | (1 << (j & 31))
        got = tiles[indptr[u]:indptr[u] + ntiles[u]]
        assert ntiles[u] == len(want), u
        assert [int(g >> np.uint64(32)) for g in got] == sorted(want), u
        assert [int(g & np.uint64(0xffffffff)) for g in got] == [want[t] for t in sorted(want)], u
    # rows in arbitrary order (columns only renamed): sorted inside the kernel; a row beyond the LDS capacity
    # takes the device-sort fallback.  Same stream either way.
    for long_len in (4000, 4700):
        ip, ix, _ = rand_csr(rng, n_users, n_items, 30, long_rows=[(2, long_len), (9, 64), (10, 65), (11, 129)],
                             empty_rows=[0, 7])
        shuffled = ix.copy()
        for u in range(n_users):
            rng.shuffle(shuffled[ip[u]:ip[u + 1]])
        sp = hip_ops.to_device(ip.astype(np.int64))
        t_ref, n_ref = hip_ops.seen_tiles(sp, hip_ops.to_device(ix.astype(np.int32)), n_users)
        t_got, n_got = hip_ops.seen_tiles(sp, hip_ops.to_device(shuffled.astype(np.int32)), n_users, rows_sorted=False)
        n_ref, n_got = hip_ops.to_host(n_ref), hip_ops.to_host(n_got)
        assert np.array_equal(n_ref, n_got)
        t_ref, t_got = hip_ops.to_host(t_ref), hip_ops.to_host(t_got)
        for u in range(n_users):
            assert np.array_equal(t_ref[ip[u]:ip[u] + n_ref[u]], t_got[ip[u]:ip[u] + n_ref[u]]), (long_len, u)


def test_spmm_row_range_and_user_batches(hip_ops):
    """A row range of the plan is its own launch; the pipelined (batched, two-stream) scoring pass must
    return exactly what the single-batch pass returns."""
    from polara_amd import scoring
    rng = np.random.RandomState(21)
    n_users, n_items, K, topk = 13000, 1500, 50, 10
    indptr, indices, values = rand_csr(rng, n_users, n_items, 25, long_rows=[(3, 1400), (9000, 1300)], empty_rows=[0, 4097],
                                       dtype=np.float32)
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    V = rng.randn(n_items, K) * ((1.0 + np.arange(n_items)) ** -0.6)[:, None]
    Vd = hip_ops.to_device(V)
    full = hip_ops.to_host(hip_ops.spmm(T, Vd))
    out = hip_ops.zeros(n_users, K)
    for lo, hi in ((0, 3), (3, 4), (4, 8999), (8999, 9001), (9001, n_users)):
        hip_ops.spmm(T, Vd, out=out, rows=(lo, hi))
    assert np.array_equal(hip_ops.to_host(out), full)
    F = scoring.FactorImage(hip_ops, Vd)
    ref, ref_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, batches=1)
    for B in (2, 3):
        got, got_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, batches=B)
        assert np.array_equal(hip_ops.to_host(ref), hip_ops.to_host(got))
        assert np.array_equal(hip_ops.to_host(ref_s), hip_ops.to_host(got_s))


def test_pack_frag_bound_matches_separate_kernels(hip_ops):
    import torch
    rng = np.random.RandomState(8)
    for n, K, ldpad in ((1000, 50, 2), (33, 7, 0), (4097, 200, 4), (64, 100, 0)):
        M = torch.zeros(n, K + ldpad, dtype=torch.float64, device=hip_ops.device)
        M[:, :K] = hip_ops.to_device(rng.randn(n, K) * np.exp(rng.randn(n, 1) * 2))
        Mv = M[:, :K]
        extra = hip_ops.to_device(np.abs(rng.randn(n)) * 100)
        p_ref = hip_ops.pack_frag(Mv.contiguous())
        p_got, b_got = hip_ops.pack_frag_bound(Mv, extra=extra, extra_scale=1.2e-7)
        assert torch.equal(p_ref, p_got)
        nrm = np.linalg.norm(hip_ops.to_host(Mv), axis=1)
        want = nrm + 1.2e-7 * hip_ops.to_host(extra)
        b = hip_ops.to_host(b_got).astype(np.float64)
        assert np.all(b >= want) and np.all(b <= want * (1 + 3e-6) + 1e-300)
        _, b0 = hip_ops.pack_frag_bound(Mv)
        b0 = hip_ops.to_host(b0).astype(np.float64)
        assert np.all(b0 >= nrm) and np.all(b0 <= nrm * (1 + 3e-6) + 1e-300)


def test_pruning_bounds_are_upper_bounds(hip_ops):
    rng = np.random.RandomState(3)
    for n, K in ((1000, 50), (33, 7), (4097, 200)):
        M = rng.randn(n, K) * np.exp(rng.randn(n, 1) * 3)
        M[5] = 0.0
        nrm = np.linalg.norm(M, axis=1)
        ub = hip_ops.to_host(hip_ops.row_norm_bound(hip_ops.to_device(M))).astype(np.float64)
        assert ub.dtype == np.float64 and np.all(ub >= nrm) and np.all(ub <= nrm * (1 + 3e-6))
        tb = hip_ops.to_host(hip_ops.tile_norm_bound(hip_ops.to_device(M))).astype(np.float64)
        suf = np.maximum.accumulate(nrm[::-1])[::-1][::32]
        assert tb.shape == suf.shape and np.all(tb >= suf) and np.all(tb <= suf * (1 + 3e-6))
        assert np.all(np.diff(tb) <= 0)


@pytest.mark.parametrize('cfg', [dict(n_users=1000, n_items=20000, K=50, topk=10, chunk=0),
                                 dict(n_users=333, n_items=9000, K=100, topk=20, chunk=13),
                                 dict(n_users=200, n_items=6000, K=24, topk=50, chunk=5),
                                 dict(n_users=130, n_items=5000, K=50, topk=10, chunk=3, splits=3)])
def test_pruned_sweep_equals_full_sweep(hip_ops, cfg):
    """Exact pruning: with item norms decaying along the catalogue most user groups must leave the
    sweep early, and the result (ids AND scores, flagged users included) must not change."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items)
    decay = (1.0 + np.arange(n_items)) ** -0.7
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    V[rng.randint(n_items // 4, n_items // 2, 3)] *= 3.0   # a few heavy items further down (bound is a suffix max)
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2)], empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    try:
        hip_ops.score_tiles_per_chunk = cfg['chunk']
        hip_ops.score_splits_override = cfg.get('splits', 0)
        st_full, st = {}, {}
        ref, ref_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st_full, prune=False)
        got, got_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st, two_phase_ok=False)
        st2 = {}
        got2, got2_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st2)   # default: two-phase
    finally:
        hip_ops.score_tiles_per_chunk = 0
        hip_ops.score_splits_override = 0
    assert st_full['tiles_scored'] == st_full['tiles_total']
    # the default pruned route for a catalogue of >= 128 tiles is the two-phase sweep (unless splits are forced)
    assert ('two_phase' in st2) == (not cfg.get('splits')) and st2['tiles_scored'] <= st2['tiles_total']
    assert np.array_equal(hip_ops.to_host(ref), hip_ops.to_host(got2))
    assert np.array_equal(hip_ops.to_host(ref_s), hip_ops.to_host(got2_s))
    # few users: the sweep is dealt out to several interleaved splits (auto: as many as fit the candidate budget);
    # each split prunes against the k-th best of its own items only, so the cut is weaker than a single sweep's
    want_splits = cfg.get('splits', hip_ops.lib.pk_score_splits(n_users, hip_ops.candidate_capacity(topk)))
    assert want_splits == {1000: 4, 333: 2, 200: 1, 130: 3}[n_users] and st['item_splits'] == want_splits
    assert st['tiles_scored'] <= st['tiles_total'], st
    if want_splits == 1 or n_users == 1000:
        assert st['tiles_scored'] < 0.6 * st['tiles_total'], st
    assert np.array_equal(hip_ops.to_host(ref), hip_ops.to_host(got))
    assert np.array_equal(hip_ops.to_host(ref_s), hip_ops.to_host(got_s))
    E = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_users, n_items)) @ V
    want, _ = brute_topk(V, E, indptr, indices, topk, True)
    live = np.abs(E).sum(1) > 0
    assert (hip_ops.to_host(got)[live] == want[live]).mean() > 0.999


@pytest.mark.parametrize('cfg', [dict(n_users=1500, n_items=9000, K=50, topk=10, head=32, splits=3, chunk=0),
                                 dict(n_users=700, n_items=5200, K=50, topk=10, head=8, splits=3, chunk=5),
                                 dict(n_users=333, n_items=9000, K=100, topk=20, head=16, splits=3, chunk=7),
                                 dict(n_users=333, n_items=9000, K=100, topk=20, head=40, splits=1, chunk=0),
                                 dict(n_users=200, n_items=6000, K=24, topk=50, head=32, splits=3, chunk=11),
                                 dict(n_users=200, n_items=6000, K=24, topk=50, head=3, splits=2, chunk=0),
                                 dict(n_users=900, n_items=12000, K=50, topk=10, head=32, splits=7, chunk=9),
                                 dict(n_users=900, n_items=12000, K=160, topk=10, head=20, splits=15, chunk=0),
                                 dict(n_users=257, n_items=4100, K=10, topk=3, head=1, splits=3, chunk=2)])
def test_two_phase_sweep_equals_single_sweep(hip_ops, cfg, monkeypatch, pk_options):
    """pk_score_two_phase_f32 (head sweep -> item splits seeded with the head's thresholds -> exact merge of the S + 1
    lists): the lists and scores of the pass are those of the single pruned sweep and of the full sweep, for heads of
    1..40 tiles, 1..15 splits (merge over 64, 128 and 256 entries), tiny item chunks in phase 2 (state parked between
    launches), ranks with and without the dense seen masks, users with very long rows (pruned inside the head: their
    splits must not run) and empty rows; also the kernel-level contract: the merged list holds the KC best fp32 scores
    of the union of the raw lists."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items + cfg['head'])
    decay = (1.0 + np.arange(n_items)) ** -0.6
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    V[rng.randint(n_items // 4, n_items // 2, 3)] *= 3.0
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2), (40, n_items - 3)], empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    pk_options('score_head_tiles', cfg['head'])
    pk_options('score_phase2_splits', cfg['splits'])
    KC = hip_ops.candidate_capacity(topk)
    want_splits = cfg['splits']
    while want_splits > 1 and (want_splits + 1) * KC > 256:
        want_splits -= 1
    assert hip_ops.two_phase_plan(n_users, n_items, KC) == (cfg['head'], want_splits)
    try:
        hip_ops.score_tiles_per_chunk = cfg['chunk']
        st_full, st1, st2 = {}, {}, {}
        ref, ref_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st_full, prune=False)
        one, one_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st1, two_phase_ok=False)
        two, two_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st2)
        ids_only = scoring.recommend(hip_ops, F, T, topk, True)                    # approximate fold-in route
    finally:
        hip_ops.score_tiles_per_chunk = 0
    assert 'two_phase' not in st1 and st2['two_phase'] == dict(st2['two_phase'], head_tiles=cfg['head'], splits=want_splits)
    for a, b in ((ref, one), (ref, two), (ref_s, one_s), (ref_s, two_s), (ref, ids_only)):
        assert np.array_equal(hip_ops.to_host(a), hip_ops.to_host(b))
    assert st2["tiles_scored"] <= st2["tiles_total"]


@pytest.mark.parametrize('cfg', [dict(n_users=1000, n_items=9000, K=50, topk=10), dict(n_users=333, n_items=5200, K=100, topk=20),
                                 dict(n_users=95, n_items=4100, K=200, topk=50), dict(n_users=70, n_items=3000, K=7, topk=5),
                                 dict(n_users=2100, n_items=2000, K=24, topk=10)])
def test_sweep_from_the_rows_of_E_equals_the_packed_route(hip_ops, cfg, monkeypatch):
    """Round 5: the sweep's waves build their users' MFMA fragments and pruning bounds from the fp64 rows of E in their
    prologue (pk_score_candidates_rows_f32 / pk_score_two_phase_rows_f32) instead of reading what pk_pack_frag_bound_f32
    wrote.  Kernel level: the raw candidate lists (scores bit for bit, ids) of both routes are EQUAL — pruned and full,
    with the error-weight column of an approximate fold-in in the bound, partial last groups, strided rows.  Pass level:
    `scoring.recommend` with the route switched off and on returns the same ids and scores (exact and ids-only passes,
    the two-phase route included)."""
    import torch
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items + K)
    decay = (1.0 + np.arange(n_items)) ** -0.6
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2)], empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    KC = hip_ops.candidate_capacity(topk)
    Kx = -(-(K + 1) // 4) * 4
    Ex = hip_ops.to_device(np.ascontiguousarray(np.c_[rng.randn(n_users, K) * np.exp(rng.randn(n_users, 1)), np.abs(rng.randn(n_users, Kx - K)) * 1e3]))
    Ex[5, :K] = 0.0
    E, w = Ex[:, :K], Ex[:, K]
    assert hip_ops.sweep_takes_rows(E) and not hip_ops.sweep_takes_rows(Ex[:, 1:K + 1])
    Ep, ub = hip_ops.pack_frag_bound(E, extra=w, extra_scale=1.2e-7)
    st = T.seen_tiles()
    for tb in (F.tile_bound, None):
        a = hip_ops.score_candidates(F.Vp, Ep, n_users, n_items, K, T.indptr, T.indices, KC, user_bound=ub if tb is not None else None,
                                     tile_bound=tb, seen_tiles=st)
        a = (a[0].clone(), a[1].clone())
        b = hip_ops.score_candidates(F.Vp, None, n_users, n_items, K, T.indptr, T.indices, KC, tile_bound=tb, seen_tiles=st,
                                     E_rows=(E, w, 1.2e-7))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), tb is None
    if n_items >= 4 * 32 * 8:
        a = hip_ops.score_two_phase(F.Vp, Ep, n_users, n_items, K, T.indptr, KC, 8, 3, ub, F.tile_bound, seen_tiles=st)
        a = (a[0].clone(), a[1].clone())
        b = hip_ops.score_two_phase(F.Vp, None, n_users, n_items, K, T.indptr, KC, 8, 3, None, F.tile_bound, seen_tiles=st,
                                    E_rows=(E, w, 1.2e-7))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    res = {}
    for on in (False, True):
        monkeypatch.setattr(scoring, 'SWEEP_FROM_ROWS', on)
        hip_ops.timers = {}
        ids, sc = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True)
        ids_only = scoring.recommend(hip_ops, F, T, topk, True)
        res[on] = (hip_ops.to_host(ids), hip_ops.to_host(sc), hip_ops.to_host(ids_only), set(hip_ops.timers))
        hip_ops.timers = None
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1])
    assert np.array_equal(res[False][2], res[True][2]) and np.array_equal(res[True][0], res[True][2])
    assert 'pack_frag_bound' in res[False][3]
    if K % 2 == 0:
        assert 'pack_frag_bound' not in res[True][3]      # (odd ranks: the exact pass's rows have an odd stride and keep the packing kernel)


@pytest.mark.parametrize('cfg', [dict(K=50, topk=10, splits=4), dict(K=50, topk=10, splits=3),
                                 dict(K=100, topk=20, splits=2), dict(K=24, topk=5, splits=2)])
def test_item_splits_equal_single_range(hip_ops, cfg):
    """Cutting the catalogue into S item ranges (own threshold/state each, merged at re-scoring) must
    not change the result; also with ranges shorter than one chunk and more splits than tiles."""
    from polara_amd import scoring
    rng = np.random.RandomState(cfg['K'] + cfg['splits'])
    K, topk = cfg['K'], cfg['topk']
    for n_users, n_items in ((130, 2900), (40, 70)):
        V = np.linalg.qr(rng.randn(n_items, K))[0] if n_items >= K else rng.randn(n_items, K)
        indptr, indices, values = rand_csr(rng, n_users, n_items, min(35, n_items // 3),
                                           long_rows=[(2, int(0.8 * n_items))], empty_rows=[5])
        T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
        F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
        try:
            hip_ops.score_splits_override = 1
            ref, ref_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True)
            hip_ops.score_splits_override = cfg['splits']
            hip_ops.score_tiles_per_chunk = 7
            st = {}
            got, got_s = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st)
        finally:
            hip_ops.score_splits_override = 0
            hip_ops.score_tiles_per_chunk = 0
        assert st['item_splits'] == cfg['splits']
        assert np.array_equal(hip_ops.to_host(ref), hip_ops.to_host(got))
        assert np.array_equal(hip_ops.to_host(ref_s), hip_ops.to_host(got_s))


@pytest.mark.parametrize('cfg', [dict(K=50, topk=10), dict(K=100, topk=20), dict(K=200, topk=50)])
def test_near_tie_scores_are_resolved_exactly(hip_ops, cfg):
    """Adversarial catalogue for the fp32 candidate pass: families of items whose factors are equal up to
    1 +- a few 1e-8 .. 1e-6 (indistinguishable or nearly so in fp32, and inside the 2^-16 window in which the
    key-only flush sorts may order either way), placed around every user's top-k boundary.  The certified
    pipeline (candidates -> exact fp64 re-scoring -> exact rows for uncertified users) must still return the
    exact fp64 order (score desc, item asc)."""
    from polara_amd import scoring
    K, topk = cfg['K'], cfg['topk']
    rng = np.random.RandomState(K)
    n_users, n_base, fam = 96, 400, 70        # family > the largest candidate capacity (64): certification must fail
    base = rng.randn(n_base, K) / np.sqrt(K) * ((1.0 + np.arange(n_base)) ** -0.5)[:, None]
    rows = []
    for i in range(n_base):
        rows.append(base[i])
        if i < 40:                            # the head of the catalogue, where the top-k boundaries fall
            for f in range(fam):
                rows.append(base[i] * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-8.5, -6.0)))
    V = np.array(rows)
    V = V[rng.permutation(len(V))]            # no helpful order: neither norm- nor id-sorted
    n_items = V.shape[0]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 25, empty_rows=[3])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    E = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_users, n_items)) @ V
    for prune in (True, False):
        st = {}
        recs, sc = scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st, prune=prune)
        recs, sc = hip_ops.to_host(recs), hip_ops.to_host(sc)
        want, s = brute_topk(V, E, indptr, indices, topk, True)
        live = np.abs(E).sum(1) > 0
        # the device sums every score in one fixed order; NumPy's dot may differ in the last bits, which can
        # flip exact-to-1e-16 pairs: compare ORDER wherever consecutive reference scores differ by > 1e-13 rel.
        for u in np.flatnonzero(live):
            ref_s = s[u, want[u]]
            got_s = s[u, recs[u]]
            assert np.allclose(got_s, ref_s, rtol=1e-12, atol=0), (u, prune)
            clear = np.abs(np.diff(ref_s)) > 1e-13 * np.abs(ref_s[:-1])
            firm = np.r_[clear, True] & np.r_[True, clear]
            assert np.array_equal(recs[u][firm], want[u][firm]), (u, prune, st)
        assert st['flagged_users'] > 0        # the near-ties really did defeat the fp32 certification
        # ids-only call: approximate fold-in (fp32 image of V) + certification of the order + exact re-do
        st2 = {}
        ids = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True, stats=st2, prune=prune))
        assert st2['approx_fold_in'] and st2['refolded_users'] > 0
        for u in np.flatnonzero(live):
            ref_s = s[u, want[u]]
            clear = np.abs(np.diff(ref_s)) > 1e-13 * np.abs(ref_s[:-1])
            firm = np.r_[clear, True] & np.r_[True, clear]
            assert np.array_equal(ids[u][firm], want[u][firm]), (u, prune, st2)


@pytest.mark.parametrize('cfg', [dict(n_users=3000, n_items=9000, K=50, topk=10),
                                 dict(n_users=700, n_items=5000, K=100, topk=20),
                                 dict(n_users=300, n_items=4000, K=26, topk=50)])
def test_approximate_fold_in_returns_the_exact_ids(hip_ops, cfg):
    """recommend(ids only) folds in against the fp32 image of V and certifies the order; it must return exactly
    what the fp64 pipeline returns (which the other tests pin to the reference), on decaying-norm factors with
    explicit zero feedback, long and empty rows; negative feedback switches the approximation off."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items + K)
    V = rng.randn(n_items, K) / np.sqrt(K) * ((1.0 + np.arange(n_items)) ** -0.6)[:, None]
    V = V[rng.permutation(n_items)]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 3)], empty_rows=[7],
                                       dtype=np.float32)
    values[::9] = 0.0
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    exact = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True, approx_fold_in=False))
    st = {}
    fast = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True, stats=st))
    assert st['approx_fold_in']
    assert np.array_equal(exact, fast), (np.flatnonzero((exact != fast).any(axis=1))[:5], st)
    for B in (2,):
        if n_users >= 2 * 4096:
            assert np.array_equal(exact, hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True, batches=B)))
    neg = values.copy()
    neg[1::5] *= -1.0
    Tn = hip_ops.csr(indptr, indices, neg, (n_users, n_items))
    st = {}
    scoring.recommend(hip_ops, F, Tn, topk, True, stats=st)
    assert not st['approx_fold_in']


def test_large_topk_uses_exact_rows(hip_ops):
    from polara_amd import scoring
    rng = np.random.RandomState(2)
    n_users, n_items, K, topk = 20, 300, 12, 120
    V = np.linalg.qr(rng.randn(n_items, K))[0]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 25)
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    recs = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True))
    E = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_users, n_items)) @ V
    ref, _ = brute_topk(V, E, indptr, indices, topk, True)
    assert np.array_equal(recs, ref)


def test_few_unseen_items_reenter_after_unseen(hip_ops):
    from polara_amd import scoring
    rng = np.random.RandomState(5)
    n_users, n_items, K, topk = 40, 48, 6, 10
    V = np.linalg.qr(rng.randn(n_items, K))[0]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 20, long_rows=[(0, 46), (1, 44), (2, 48), (3, 39)])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    stats = {}
    recs = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, True, stats=stats))
    test_data = (np.repeat(np.arange(n_users), np.diff(indptr)), indices.astype(np.int64), values.astype(np.float64))
    ref = orc.svd_recommendations(V, test_data, (n_users, n_items), topk, True)   # the reference's own rule
    assert stats['flagged_users'] >= 3
    assert np.array_equal(recs, ref)


@pytest.mark.parametrize('ranks', [(6, 5, 3), (13, 10, 2), (30, 30, 4), (4, 3, 5)])
def test_ttm_matches_dttm_seq(hip_ops, ranks):
    from polara_amd import tucker
    rng = np.random.RandomState(sum(ranks))
    shape = (500, 300, 5)
    nnz = 20000
    idx = np.stack([rng.randint(0, s, nnz) for s in shape], 1).astype(np.intp)
    idx[:3000, 0] = 7   # a long output row
    val = np.ones(nnz)
    r0, r1, r2 = ranks
    u0, u1, u2 = rng.randn(shape[0], r0), rng.randn(shape[1], r1), rng.randn(shape[2], r2)
    for (m0, mu, mv), (Uu, Uv), modes in (((0, 2, 1), (u2, u1), ((2, 0), (1, 0))),
                                          ((1, 2, 0), (u2, u0), ((2, 0), (0, 0))),
                                          ((2, 1, 0), (u1, u0), ((1, 0), (0, 0)))):
        mp = tucker.ModePlan(hip_ops, idx, val, shape, m0, mu, mv)
        res = hip_ops.to_host(tucker.ttm(hip_ops, mp, hip_ops.to_device(Uu), hip_ops.to_device(Uv)))
        ref = orc.ttm3d_seq(idx, val, shape, Uu, Uv, modes).reshape(shape[m0], -1)
        assert np.abs(res - ref).max() <= 1e-12 * np.abs(ref).max(), (ranks, m0)
    valr = rng.rand(nnz)
    mp = tucker.ModePlan(hip_ops, idx, valr, shape, 0, 2, 1)
    res = hip_ops.to_host(tucker.ttm(hip_ops, mp, hip_ops.to_device(u2), hip_ops.to_device(u1)))
    ref = orc.ttm3d_seq(idx, valr, shape, u2, u1, ((2, 0), (1, 0))).reshape(shape[0], -1)
    assert np.abs(res - ref).max() <= 1e-12 * np.abs(ref).max()


def check_factored_products(ops, ranks, weighted):
    """tucker.factored_products (SpMM over the unfolded CSR images + dense contractions on the matrix cores) against the
    reference's per-entry loop, `ttm3d_seq` -> `dttm_seq` (lib/tensor.py:7-19, lib/sparse.py:203-216), all three modes,
    duplicate coordinates included."""
    from polara_amd import tucker
    rng = np.random.RandomState(sum(ranks) + int(weighted))
    shape = (500, 300, 5)
    nnz = 20000
    idx = np.stack([rng.randint(0, s, nnz) for s in shape], 1).astype(np.intp)
    idx[:3000, 0] = 7        # a long output row (and plenty of duplicate coordinates)
    idx[3000:5000, 1] = 11
    val = rng.rand(nnz) if weighted else np.ones(nnz)
    r0, r1, r2 = ranks
    u0, u1, u2 = rng.randn(shape[0], r0), rng.randn(shape[1], r1), rng.randn(shape[2], r2)
    d0, d1, d2 = (ops.to_device(u) for u in (u0, u1, u2))
    uf = tucker.Unfoldings(ops, idx, None if not weighted else val, shape)
    res0, _ = tucker.factored_products(ops, uf, None, d1, d2, 0)
    res1, W1 = tucker.factored_products(ops, uf, d0, d1, d2, 1)
    res2, _ = tucker.factored_products(ops, uf, d0, d1, d2, 2, W1=W1)
    res2b, _ = tucker.factored_products(ops, uf, d0, d1, d2, 2)            # W1 recomputed
    for res, (Uu, Uv), modes, m0 in ((res0, (u2, u1), ((2, 0), (1, 0)), 0), (res1, (u2, u0), ((2, 0), (0, 0)), 1),
                                      (res2, (u1, u0), ((1, 0), (0, 0)), 2), (res2b, (u1, u0), ((1, 0), (0, 0)), 2)):
        ref = orc.ttm3d_seq(idx, val, shape, Uu, Uv, modes).reshape(shape[m0], -1)
        got = ops.to_host(res)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (ranks, m0)


@pytest.mark.parametrize('ranks', [(6, 5, 3), (13, 10, 2), (30, 30, 4), (4, 3, 5), (17, 33, 1)])
@pytest.mark.parametrize('weighted', [False, True])
def test_factored_mode_products_match_dttm_seq(hip_ops, ranks, weighted):
    check_factored_products(hip_ops, ranks, weighted)


def test_dense_scores_rows(hip_ops):
    rng = np.random.RandomState(9)
    V, E = rng.randn(1234, 37), rng.randn(5, 37)
    out = hip_ops.to_host(hip_ops.dense_scores(hip_ops.to_device(V), hip_ops.to_device(E)))
    assert np.allclose(out, E @ V.T, rtol=1e-12, atol=1e-12)


def test_errors_are_loud(hip_ops):
    from polara_amd._lib import PolaraHipError
    with pytest.raises(PolaraHipError):
        hip_ops.eigh_psd(hip_ops.zeros(2000, 2000))
    import torch
    A = hip_ops.csr(np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int32), np.array([1.0]), (1, 10))
    X32 = torch.zeros(10, 6, dtype=torch.float32, device=hip_ops.device)
    with pytest.raises(PolaraHipError):
        hip_ops.spmm(A, X32)   # an fp32 dense block needs nc % 4 == 0
    # more than 256 columns go panel by panel (ops.spmm), not to an error
    X = hip_ops.to_device(np.arange(3000, dtype=np.float64).reshape(10, 300))
    assert np.array_equal(hip_ops.to_host(hip_ops.spmm(A, X))[0], np.arange(300, dtype=np.float64))


@pytest.mark.parametrize('K', [8, 50, 100, 200, 256])
def test_split_bf16_candidate_scores_stay_inside_the_certified_error(hip_ops, K):
    """The candidate sweep computes every product on the bf16 matrix cores with each operand split into two bf16
    (score.hip).  Its error model — |s32 - e.v| <= (3 * 2^-16 + (4 K + 10) * 2^-23) ||e|| ||v||, the `bound` of
    rescore.hip — is what the certification of the final lists rests on: checked here directly, on well-scaled rows and
    on rows whose entries span 12 orders of magnitude, with every item scored (no seen items, no pruning)."""
    ops = hip_ops
    rng = np.random.RandomState(K)
    n_users, n_items = 256, 2048
    for wide in (False, True):
        scale_v = 10.0 ** rng.uniform(-6, 6, (n_items, K)) if wide else 1.0
        scale_e = 10.0 ** rng.uniform(-6, 6, (n_users, K)) if wide else 1.0
        V = rng.randn(n_items, K) * scale_v
        E = rng.randn(n_users, K) * scale_e
        V /= np.linalg.norm(V, axis=1).max()
        Vp = ops.pack_frag(ops.to_device(V))
        Ep, ub = ops.pack_frag_bound(ops.to_device(E))
        KC = 64
        cs, ci = ops.score_candidates(Vp, Ep, n_users, n_items, K, None, None, KC)      # full sweep, nothing masked
        cs, ci = ops.to_host(cs)[:n_users * KC].reshape(n_users, KC), ops.to_host(ci)[:n_users * KC].reshape(n_users, KC)
        exact = E @ V.T
        rel = 3 * 2.0 ** -16 + (4 * K + 10) * 2.0 ** -23
        bound = rel * np.linalg.norm(E, axis=1)[:, None] * np.linalg.norm(V, axis=1)[None, :]
        valid = ci >= 0
        assert valid.all()
        got_exact = np.take_along_axis(exact, ci, axis=1)
        err = np.abs(cs.astype(np.float64) - got_exact)
        lim = np.take_along_axis(bound, ci, axis=1)
        assert (err <= lim).all(), (K, wide, float((err / lim).max()))
        # the error budget is used, not vacuous: the worst observed error is within two orders of it
        assert (err / lim).max() > 1e-3 or K < 16
        # and the kept candidates are the top-KC up to that error: every excluded item is below the KC-th kept + 2 bounds
        kth = cs.min(axis=1)
        excl = np.ones_like(exact, dtype=bool)
        np.put_along_axis(excl, ci, False, axis=1)
        slack = bound.max(axis=1) * 2 + np.abs(kth) * 2.0 ** -15
        assert (np.where(excl, exact, -np.inf).max(axis=1) <= kth + slack).all()


@pytest.mark.parametrize('n_items,K,topk', [(5000, 50, 10), (1024, 7, 50), (1025, 200, 256), (3000, 33, 300), (40, 5, 64),
                                            (2500, 64, 2500)])
def test_exact_rows_chunk_path_and_one_workgroup_path_agree_with_numpy(hip_ops, n_items, K, topk):
    """pk_score_exact_rows_f64 / pk_score_exact_list_f64: catalogues that span several 1024-item chunks, a ragged last
    chunk, lists longer than a chunk's share (topk > 256 takes the one-workgroup kernel), more listed users than row
    slots (the overflow takes the one-workgroup kernel), users with nearly everything seen (seen items re-enter after the
    unseen ones), exact ties (broken by index) — ids AND score bits equal between the routes, ids equal to NumPy."""
    import numpy_ops
    rng = np.random.RandomState(n_items + K)
    n_users = 37
    V = rng.randn(n_items, K)
    V[7] = V[3]                                                   # exact ties
    V[n_items - 1] = V[0]
    E = rng.randn(n_users, K)
    rows_seen = []
    for u in range(n_users):
        m = n_items - 3 if u == 5 else (n_items if u == 6 else rng.randint(0, min(n_items, 400)))
        rows_seen.append(np.sort(rng.choice(n_items, m, replace=False)))
    seen_ptr = np.r_[0, np.cumsum([len(x) for x in rows_seen])].astype(np.int64)
    seen_idx = np.concatenate(rows_seen).astype(np.int32)
    Vd, Ed = hip_ops.to_device(V), hip_ops.to_device(E)
    sp, si = torch.from_numpy(seen_ptr).to(Vd.device), torch.from_numpy(seen_idx).to(Vd.device)
    rows = torch.from_numpy(rng.permutation(n_users).astype(np.int32)).to(Vd.device)
    idx, sc = hip_ops.score_exact_rows(rows, Vd, Ed, n_items, sp, si, topk)
    idx, sc = hip_ops.to_host(idx), hip_ops.to_host(sc)
    k_eff = min(topk, n_items)
    for r, u in enumerate(hip_ops.to_host(rows)):
        s = (V * E[u]).sum(axis=1)                 # identical rows of V -> identical scores (a BLAS gemv need not)
        cls = np.zeros(n_items, dtype=np.int64)
        cls[rows_seen[u]] = 1
        want = np.lexsort((np.arange(n_items), -s, cls))[:k_eff]
        if not np.array_equal(idx[r, :k_eff], want):
            # the two summation orders may swap items whose scores agree to rounding: nothing else
            bad = np.flatnonzero(idx[r, :k_eff] != want)
            assert np.array_equal(np.sort(idx[r, :k_eff]), np.sort(want)) or len(bad) <= 4, (u, bad[:10])
            assert np.allclose(s[idx[r, bad]], s[want[bad]], rtol=1e-12, atol=1e-12) and (cls[idx[r, bad]] == cls[want[bad]]).all(), (u, bad[:10])
        assert np.allclose(sc[r, :k_eff], s[idx[r, :k_eff]], rtol=1e-12, atol=1e-12)
        assert (idx[r, k_eff:] == -1).all()
    # the device list with only 8 row slots: 8 users through the chunk kernels, 29 through the one-workgroup kernel
    out_i = torch.full((n_users, topk), -7, dtype=torch.int64, device=Vd.device)
    out_s = hip_ops.empty(n_users, topk)
    cnt = torch.tensor([n_users], dtype=torch.int32, device=Vd.device)
    hip_ops._exact_work = None
    hip_ops.score_exact_list(rows, cnt, Vd, Ed, n_items, sp, si, topk, out_i, out_s, n_wg=8)
    hip_ops._exact_work = None
    got_i, got_s = hip_ops.to_host(out_i), hip_ops.to_host(out_s)
    order = hip_ops.to_host(rows)
    assert np.array_equal(got_i[order], idx)
    assert np.array_equal(got_s[order][:, :k_eff].view(np.int64), sc[:, :k_eff].view(np.int64))     # same BITS on both routes
    # a shorter device-side count leaves the other users' rows alone
    out_i.fill_(-7)
    cnt.fill_(3)
    hip_ops.score_exact_list(rows, cnt, Vd, Ed, n_items, sp, si, topk, out_i, out_s, n_wg=8)
    hip_ops._exact_work = None
    got_i = hip_ops.to_host(out_i)
    assert np.array_equal(got_i[order[:3]], idx[:3]) and (got_i[order[3:]] == -7).all()


def test_scatter_rows_to_device_and_to_pinned_host_memory(hip_ops):
    """pk_scatter_rows_i64: the lists of a pass back in the caller's user order; the destination may be the pinned host
    array itself (written by the kernel over PCIe)."""
    rng = np.random.RandomState(3)
    n, w = 1234, 7
    src = torch.from_numpy(rng.randint(-5, 10**12, (n, w))).cuda()
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    want = np.empty((n, w), dtype=np.int64)
    want[perm.cpu().numpy()] = src.cpu().numpy()
    assert np.array_equal(hip_ops.to_host(hip_ops.scatter_rows(src, perm)), want)
    host = torch.zeros((n, w), dtype=torch.int64).pin_memory()
    hip_ops.scatter_rows(src, perm, out=host)
    torch.cuda.synchronize()
    assert np.array_equal(host.numpy(), want)
    assert np.array_equal(hip_ops.to_host(hip_ops.scatter_rows(src, None)), src.cpu().numpy())
    with pytest.raises(AssertionError):
        hip_ops.scatter_rows(src, perm, out=torch.zeros((n, w), dtype=torch.int64))     # pageable host memory


def test_dense_seen_masks_equal_the_stream(hip_ops, monkeypatch):
    """pk_seen_dense_build against NumPy, and the sweep with a dense window of 1, 3, 8, all tiles against the sweep on
    the seen-tile stream alone: same candidates, same lists (the window ends in the middle of the catalogue, so the
    hand-over from the dense masks to the stream cursor is exercised; heavy users have a record in every tile)."""
    from polara_amd import scoring
    ops = hip_ops
    rng = np.random.RandomState(8)
    n_users, n_items, K, topk = 700, 1000, 16, 10
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(0, 900), (1, 640), (33, 990), (699, 500)])
    V = np.linalg.qr(rng.randn(n_items, K))[0] * np.linspace(4, 0.2, n_items)[:, None]
    F = scoring.FactorImage(ops, ops.to_device(V))
    monkeypatch.setattr(ops, 'score_splits_override', 1)      # one sweep per group: item splits read the stream only

    def lists(window):
        monkeypatch.setattr(ops, 'seen_dense_tiles', window)
        T = ops.csr(indptr, indices, values, (n_users, n_items))
        sd = T.seen_dense()
        idx, sc = scoring.recommend(ops, F, T, topk, True, return_scores=True)
        ids = scoring.recommend(ops, F, T, topk, True)
        return sd, ops.to_host(idx), ops.to_host(sc), ops.to_host(ids)

    sd0, idx0, sc0, ids0 = lists(0)
    assert sd0 is None
    for window in (1, 3, 8, 32):
        sd, idx, sc, ids = lists(window)
        dense, skip, dt = sd
        assert dt == min(window, -(-n_items // 32))
        want = np.zeros((-(-n_users // 32), dt, 32), dtype=np.uint32)
        want_skip = np.zeros(n_users, dtype=np.int32)
        for u in range(n_users):
            row = indices[indptr[u]:indptr[u + 1]]
            row = row[row < 32 * dt]
            np.bitwise_or.at(want[u // 32, :, u % 32], row // 32, (1 << (row % 32)).astype(np.uint32))
            want_skip[u] = len(np.unique(row // 32))
        assert np.array_equal(ops.to_host(dense).view(np.uint32), want) and np.array_equal(ops.to_host(skip), want_skip)
        assert np.array_equal(idx, idx0) and np.array_equal(sc, sc0) and np.array_equal(ids, ids0), window


@pytest.mark.parametrize('cfg', [dict(n_users=700, n_items=6000, K=50, topk=10, chunk=0),
                                 dict(n_users=333, n_items=3000, K=100, topk=20, chunk=7),
                                 dict(n_users=200, n_items=2600, K=24, topk=50, chunk=0),
                                 dict(n_users=130, n_items=900, K=16, topk=5, chunk=3)])
def test_threshold_bootstrap_changes_nothing(hip_ops, cfg, monkeypatch, pk_options):
    """The threshold bootstrap in front of a cold sweep (score.hip: the first tiles scored once without selecting, the
    sweep then starts from a lower bound of every user's KC-th best score): ids AND scores of the pass equal those of
    the cold start, for KC = 16 / 32 / 64, bootstraps shorter and longer than the catalogue, pruned and full sweeps,
    `filter_seen` off, users who have seen nearly everything (their bootstrap finds fewer than KC unseen items and
    yields no threshold) and empty users; the raw candidate lists of a bootstrapped sweep are full (no PK_IDX_FLOOR
    mark) and hold the same KC best fp32 scores."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items)
    decay = (1.0 + np.arange(n_items)) ** -0.5
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2), (40, n_items - 3), (41, n_items - 20)],
                                       empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    pk_options('score_head_tiles', 0)
    out = {}
    try:
        hip_ops.score_tiles_per_chunk = cfg['chunk']
        for boot in (0, 16, 3, 4000):
            pk_options('score_boot_tiles', boot)
            st = {}
            out[boot] = [scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st),
                         scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, prune=False),
                         scoring.recommend(hip_ops, F, T, topk, False, return_scores=True),
                         scoring.recommend(hip_ops, F, T, topk, True)]
            assert st['flagged_users'] <= 4, st          # the three nearly-all-seen users and nobody else goes the exact way
    finally:
        hip_ops.score_tiles_per_chunk = 0
    for boot in (16, 3, 4000):
        for a, b in zip(out[0], out[boot]):
            if isinstance(a, tuple):
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), boot
            else:
                assert torch.equal(a, b), boot
    # kernel level: full lists, same candidates (as sets of fp32 scores) with and without the bootstrap
    KC = hip_ops.candidate_capacity(topk)
    E = hip_ops.spmm(T, F.V)
    Ep, ub = hip_ops.pack_frag_bound(E)
    lists = {}
    for boot in (0, 16):
        pk_options('score_boot_tiles', boot)
        cs, ci = hip_ops.score_candidates(F.Vp, Ep, n_users, n_items, K, T.indptr, T.indices, KC, user_bound=ub,
                                          tile_bound=F.tile_bound, seen_tiles=T.seen_tiles())
        lists[boot] = (hip_ops.to_host(cs)[:n_users * KC].reshape(n_users, KC), hip_ops.to_host(ci)[:n_users * KC].reshape(n_users, KC))
    assert (lists[16][1] != -2).all()
    unseen = n_items - np.diff(indptr)
    full = unseen >= KC
    assert (lists[16][1][full] >= 0).all() and (lists[0][1][full] >= 0).all()
    # the same KC best fp32 scores; the ids may differ only among scores that agree to the key-sort tolerance (2^-16
    # relative) with the KC-th one: the flush sorts order those arbitrarily and the two sweeps flush at different times
    s0, s16 = -np.sort(-lists[0][0][full], axis=1), -np.sort(-lists[16][0][full], axis=1)
    assert np.allclose(s0, s16, rtol=2.0 ** -14, atol=0)
    kth = np.minimum(s0[:, -1], s16[:, -1])
    clear0 = lists[0][0][full] > (kth + np.abs(kth) * 2.0 ** -14)[:, None]
    for row0, row16, c in zip(lists[0][1][full], lists[16][1][full], clear0):
        assert np.isin(row0[c], row16).all()


def test_rescore_sends_unbounded_lists_to_the_exact_path(hip_ops):
    """A list whose last slot carries PK_IDX_FLOOR (-2: the sweep started from a bootstrapped threshold and did not fill
    the list — nothing bounds the items it left out) must be flagged for the exact path by the re-scoring kernel, and the
    pass must then still return the exact lists."""
    from polara_amd import scoring
    rng = np.random.RandomState(5)
    n_users, n_items, K, topk = 300, 2000, 20, 10
    V = rng.randn(n_items, K)
    indptr, indices, values = rand_csr(rng, n_users, n_items, 30)
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    KC = hip_ops.candidate_capacity(topk)
    E = hip_ops.spmm(T, F.V)
    Ep, ub = hip_ops.pack_frag_bound(E)
    cs, ci = hip_ops.score_candidates(F.Vp, Ep, n_users, n_items, K, T.indptr, T.indices, KC, seen_tiles=T.seen_tiles())
    marked = torch.tensor([3, 64, 65, 299], device=ci.device)
    ci2 = ci.clone()
    ci2.view(-1, KC)[marked, KC - 1] = -2
    idx0, s0, f0 = hip_ops.rescore_topk(F.V, E, n_items, T.indptr, KC, cs, ci, topk, F.vmax)
    idx1, s1, f1 = hip_ops.rescore_topk(F.V, E, n_items, T.indptr, KC, cs, ci2, topk, F.vmax)
    f0, f1 = hip_ops.to_host(f0), hip_ops.to_host(f1)
    m = np.zeros(n_users, bool)
    m[hip_ops.to_host(marked)] = True
    assert (f1[m] & 1).all() and np.array_equal(f1[~m], f0[~m]) and not (f0[m] & 1).any()


def test_topk_rows_matches_the_reference_topsort(hip_ops):
    """pk_topk_rows_f64 (the array form of get_topk_elements / topsort, models.py:488-491, 561-563) against the oracle's
    argpartition + argsort on rows without ties, and against (score desc, column asc) on rows with ties, NaNs last."""
    rng = np.random.RandomState(12)
    for n_rows, n_cols, k in ((7, 1000, 10), (1, 33, 33), (300, 257, 5), (3, 70000, 50)):
        s = rng.randn(n_rows, n_cols)
        got = hip_ops.to_host(hip_ops.topk_rows(hip_ops.to_device(s), k))
        assert np.array_equal(got, orc.get_topk_elements(s.copy(), k))
    s = np.round(rng.randn(50, 400), 1)                      # plenty of exact ties
    s[3, 5] = np.nan
    got = hip_ops.to_host(hip_ops.topk_rows(hip_ops.to_device(s), 25))
    key = np.where(np.isnan(s), -np.inf, s)
    want = np.stack([np.lexsort((np.arange(400), -key[r]))[:25] for r in range(50)])
    assert np.array_equal(got, want)
    # strided rows
    w = hip_ops.to_device(rng.randn(9, 130))
    assert np.array_equal(hip_ops.to_host(hip_ops.topk_rows(w[:, :100], 4)), orc.get_topk_elements(hip_ops.to_host(w)[:, :100].copy(), 4))


def _eigh_top_cases():
    rs = np.random.RandomState(3)
    cases = []
    for n in (120, 150, 176, 64, 9, 33):
        M = rs.randn(2000, n) * np.exp(-np.arange(n) / 12.0)[None, :]
        M = M @ np.linalg.qr(rs.randn(n, n))[0]
        cases.append(('graded_%d' % n, M.T @ M, min(30, n // 2)))
    n = 150
    Q = np.linalg.qr(rs.randn(n, n))[0]
    w = np.r_[np.full(10, 5.0), np.full(10, 5.0 - 1e-9), np.linspace(4, 1, 20), rs.rand(n - 40) * 0.5]
    cases.append(('ten_equal_ten_near', (Q * w) @ Q.T, 30))
    w = np.r_[np.linspace(3, 1, 25), np.zeros(n - 25)]
    cases.append(('rank_25_of_150_r_30', (Q * w) @ Q.T, 30))
    A = rs.randn(n, n)
    cases.append(('dense_r32', A @ A.T, 32))
    cases.append(('scale_1e200', (A @ A.T) * 1e200, 30))
    cases.append(('scale_1e-200', (A @ A.T) * 1e-200, 30))
    cases.append(('r_1', A @ A.T, 1))
    cases.append(('identity', np.eye(n), 30))
    cases.append(('diagonal', np.diag(np.arange(n, 0, -1.0)), 30))
    cases.append(('zero', np.zeros((n, n)), 30))
    return cases


@pytest.mark.parametrize('case', _eigh_top_cases(), ids=lambda c: c[0])
def test_eigh_top_leading_pairs_vs_lapack(hip_ops, case):
    """csrc/eigh_top.hip (the r leading eigenpairs of an unfolding's Gram matrix — the k of `svds(unfolding, k=r)`,
    lib/tensor.py:70-80) against numpy.linalg.eigh: eigenvalues to 1e-13 ||S||, residuals and orthonormality to 1e-12,
    the leading invariant subspace where the r-th gap defines one; graded, clustered, rank-deficient, badly scaled and
    degenerate spectra; `ops.eigh_top` must hand back a valid answer whatever the direct kernel's own verdict was (the
    zero matrix is refused by it: Jacobi fallback).  Same bits on a second call."""
    name, S, r = case
    n = S.shape[0]
    Sd = hip_ops.to_device(S)
    lam, C = hip_ops.eigh_top(Sd, r)
    lam2, C2 = hip_ops.eigh_top(Sd, r)
    lam, X = hip_ops.to_host(lam), hip_ops.to_host(C)
    assert lam.shape == (r,) and X.shape == (n, r)
    assert np.array_equal(lam, hip_ops.to_host(lam2)) and np.array_equal(X, hip_ops.to_host(C2))
    w, V = np.linalg.eigh(S)
    w, V = w[::-1], V[:, ::-1]
    scale = max(abs(w).max(), 1e-300)
    assert np.all(np.diff(lam) <= 0) and lam.min() >= 0
    assert abs(lam - w[:r]).max() <= 1e-13 * scale, name
    assert abs(S @ X - X * lam).max() <= 1e-12 * scale, name
    assert abs(X.T @ X - np.eye(r)).max() <= 1e-12, name
    if r < n and w[r - 1] - w[r] > 1e-6 * scale:
        assert abs(X @ X.T - V[:, :r] @ V[:, :r].T).max() <= 1e-9, name
    assert np.array_equal(hip_ops.to_host(Sd), S)                      # the input is read only


def test_eigh_top_kernel_reports_what_it_cannot_do(hip_ops):
    """The C entry point refuses shapes outside its range with an error message, and its verdict for the zero matrix is
    0 (nothing written) — the contract `ops.eigh_top` and driver.hip's `eigh_lead` build their fallback on."""
    from polara_amd import _lib
    from polara_amd.ops import _ptr
    lib = hip_ops.lib
    assert lib.pk_eigh_top_supported(150, 30) == 1 and lib.pk_eigh_top_supported(177, 30) == 0
    assert lib.pk_eigh_top_supported(150, 33) == 0 and lib.pk_eigh_top_supported(7, 2) == 0 and lib.pk_eigh_top_supported(20, 21) == 0
    n, r = 150, 30
    Sd = torch.zeros(n, n, dtype=torch.float64, device=hip_ops.device)
    R = torch.full((r, n), 7.0, dtype=torch.float64, device=hip_ops.device)
    lam = torch.full((r,), 7.0, dtype=torch.float64, device=hip_ops.device)
    info = torch.full((4,), 5, dtype=torch.int32, device=hip_ops.device)
    work = hip_ops._work(lib.pk_eigh_top_work_bytes(n))
    rc = lib.pk_eigh_top_f64(hip_ops.stream(), n, _ptr(Sd), n, r, _ptr(R), n, _ptr(lam), _ptr(work), _ptr(info))
    assert rc == 0 and int(info[0].item()) == 0 and float(R.min()) == 7.0 and float(lam.min()) == 7.0
    rc = lib.pk_eigh_top_f64(hip_ops.stream(), 200, _ptr(Sd), 200, r, _ptr(R), 200, _ptr(lam), _ptr(work), _ptr(info))
    assert rc != 0 and b'pk_eigh_top_f64' in lib.pk_last_error()
