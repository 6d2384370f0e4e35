"""The certification logic of the scoring pass (scoring.py + the formulas of rescore.hip, mirrored in
tests/numpy_ops.py) on the CPU double: fp32 candidate selection, re-scoring against the fp32 factor image with the
error budget delta_u = 2^-24 (w_u + ||E'_u||) max||V_i||, exact re-do of the users that do not clear it.  The HIP
kernels themselves face the same catalogue in tests/test_gpu_kernels.py::test_near_tie_scores_are_resolved_exactly."""
import numpy as np
import pytest
import scipy.sparse as sps

from numpy_ops import NumpyOps
from test_gpu_kernels import rand_csr, brute_topk


@pytest.mark.parametrize('cfg', [dict(K=20, topk=10), dict(K=50, topk=20)])
def test_near_ties_are_certified_or_redone_exactly(cfg):
    from polara_amd import scoring
    ops = NumpyOps()
    K, topk = cfg['K'], cfg['topk']
    rng = np.random.RandomState(K)
    n_users, n_base, fam = 40, 160, 70
    base = rng.randn(n_base, K) / np.sqrt(K) * ((1.0 + np.arange(n_base)) ** -0.5)[:, None]
    rows = []
    for i in range(n_base):
        rows.append(base[i])
        if i < 12:                                      # near-copies of the head items: fp32 cannot tell them apart
            for f in range(fam):
                rows.append(base[i] * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-8.5, -6.0)))
    V = np.array(rows)
    V = V[rng.permutation(len(V))]
    n_items = V.shape[0]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 25, empty_rows=[3])
    T = ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(ops, ops.to_device(V))
    E = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_users, n_items)) @ V
    want, s = brute_topk(V, E, indptr, indices, topk, True)
    live = np.abs(E).sum(1) > 0

    def check(ids):
        for u in np.flatnonzero(live):
            ref_s = s[u, want[u]]
            clear = np.abs(np.diff(ref_s)) > 1e-13 * np.abs(ref_s[:-1])
            firm = np.r_[clear, True] & np.r_[True, clear]
            assert np.array_equal(ids[u][firm], want[u][firm]), u

    st = {}
    ids, sc = scoring.recommend(ops, F, T, topk, True, return_scores=True, stats=st)
    check(ops.to_host(ids))
    assert not st['approx_fold_in'] and st['flagged_users'] > 0          # fp32 could not certify the tied families
    st2 = {}
    check(ops.to_host(scoring.recommend(ops, F, T, topk, True, stats=st2)))
    assert st2['approx_fold_in'] and st2['refolded_users'] > 0           # ... nor could the fp32 images: exact re-do
    # the error budget is not vacuous: without any near-copies (next to) nobody needs the re-do
    Vc = base[rng.permutation(n_base)]
    ip, ix, vl = rand_csr(rng, n_users, n_base, 12)
    st3 = {}
    ids3 = ops.to_host(scoring.recommend(ops, scoring.FactorImage(ops, ops.to_device(Vc)), ops.csr(ip, ix, vl, (n_users, n_base)),
                                         topk, True, stats=st3))
    w3, _ = brute_topk(Vc, sps.csr_matrix((vl.astype(np.float64), ix, ip), shape=(n_users, n_base)) @ Vc, ip, ix, topk, True)
    assert st3['approx_fold_in'] and st3['refolded_users'] <= 2 and st3['flagged_users'] == 0
    assert np.array_equal(ids3, w3)
