"""-m gpu: a seeded sweep of awkward shapes through the whole scoring pipeline (ids-only call with the certified
approximate fold-in AND the exact call with scores) against a brute-force fp64 ranking: sizes around the tile /
group / candidate-capacity boundaries, odd ranks, dense and empty rows, decaying and flat factor norms."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _configs():
    rng = np.random.RandomState(20260926)
    out = []
    ks = [1, 2, 7, 10, 16, 25, 26, 33, 50, 64, 100, 129, 200, 256]
    topks = [1, 5, 10, 11, 20, 24, 25, 50, 52]
    for i in range(72):
        n_items = int(rng.choice([31, 32, 33, 64, 95, 640, 1000, 2049, 5000, 12000]))
        topk = int(rng.choice([t for t in topks if t <= n_items]))
        out.append(dict(seed=i, n_users=int(rng.choice([1, 31, 32, 33, 64, 100, 257, 700, 2100])), n_items=n_items,
                        K=int(rng.choice(ks)), topk=topk, decay=float(rng.choice([0.0, 0.4, 1.0, 1.5])),
                        per_row=int(rng.choice([0, 3, 20, 60, 150])), filter_seen=bool(rng.rand() < 0.8)))
    return out


@pytest.mark.parametrize('cfg', _configs(), ids=lambda c: 'u%d_i%d_K%d_k%d_s%d' % (c['n_users'], c['n_items'], c['K'], c['topk'], c['seed']))
def test_random_config_against_brute_force(hip_ops, cfg):
    from polara_amd import scoring
    rng = np.random.RandomState(cfg['seed'])
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    V = rng.randn(n_items, K) / np.sqrt(K) * ((1.0 + np.arange(n_items)) ** -cfg['decay'])[:, None]
    V = V[rng.permutation(n_items)]
    per_row = min(cfg['per_row'], n_items - 1)
    rows, cols = [], []
    for u in range(n_users):
        n = 0 if per_row == 0 else int(rng.randint(0, per_row + 1))
        if u == 0 and n_items > 40:
            n = n_items - 3                                    # nearly everything seen: fewer than topk unseen
        c = np.sort(rng.choice(n_items, size=min(n, n_items), replace=False))
        rows.append(np.full(len(c), u))
        cols.append(c)
    rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    vals = rng.randint(0, 6, size=len(rows)).astype(np.float64)   # zeros stay "seen" but carry no weight
    A = sps.csr_matrix((vals, (rows, cols)), shape=(n_users, n_items))
    A.sort_indices()
    indptr, indices = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    # rebuild explicit zeros (scipy drops nothing here, but keep the triplet as given)
    T = hip_ops.csr(indptr, indices, A.data, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    E = A @ V
    s = E @ V.T
    fs = cfg['filter_seen']
    want = np.empty((n_users, topk), dtype=np.int64)
    for u in range(n_users):
        cls = np.zeros(n_items, dtype=np.int64)
        if fs:
            cls[indices[indptr[u]:indptr[u + 1]]] = 1
        want[u] = np.lexsort((np.arange(n_items), -s[u], cls))[:topk]
    ids = hip_ops.to_host(scoring.recommend(hip_ops, F, T, topk, fs))
    ids2, sc = scoring.recommend(hip_ops, F, T, topk, fs, return_scores=True)
    ids2, sc = hip_ops.to_host(ids2), hip_ops.to_host(sc)
    for u in range(n_users):
        ref_s = s[u, want[u]]
        # positions whose reference score is separated from its neighbours (ties at 1e-13 relative and all-zero
        # profiles are implementation-defined: device and NumPy sum in different orders)
        scale = max(np.abs(ref_s).max(), 1e-300)
        clear = np.abs(np.diff(ref_s)) > 1e-12 * scale
        firm = np.r_[clear, True] & np.r_[True, clear]
        if fs:                                               # a seen/unseen class change is always a firm boundary
            seen = np.isin(want[u], indices[indptr[u]:indptr[u + 1]])
            firm |= np.r_[seen[1:] != seen[:-1], False] & np.r_[False, seen[1:] != seen[:-1]]
        last = s[u, want[u][-1]]
        unseen_left = n_items - (indptr[u + 1] - indptr[u] if fs else 0)
        if unseen_left > topk:                               # the k-th must also beat the best excluded item clearly
            excl = np.setdiff1d(np.arange(n_items), want[u])
            if fs:
                excl = np.setdiff1d(excl, indices[indptr[u]:indptr[u + 1]])
            if len(excl) and not (last - s[u, excl].max() > 1e-12 * scale):
                firm[-1] = False
        assert np.array_equal(ids[u][firm], want[u][firm]), (cfg, u, ids[u], want[u])
        assert np.array_equal(ids2[u][firm], want[u][firm]), (cfg, u)
        assert np.allclose(sc[u][firm], ref_s[firm], rtol=1e-11, atol=1e-13 * scale), (cfg, u)
