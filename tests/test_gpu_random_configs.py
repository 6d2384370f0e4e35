"""-m gpu: a seeded sweep of awkward shapes through the whole scoring pipeline (ids-only call with the certified
approximate fold-in AND the exact call with scores) against a brute-force fp64 ranking: sizes around the tile /
group / candidate-capacity boundaries, odd ranks, dense and empty rows, decaying and flat factor norms."""
import os

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

# Bug hunting beyond the committed sweep: PK_SWEEP_SEED=<int> draws a different set of configurations and
# PK_SWEEP_N=<int> more of them (both sweeps); the defaults are what the suite runs.
SWEEP_SEED = int(os.environ.get('PK_SWEEP_SEED', '0'))
SWEEP_N = int(os.environ.get('PK_SWEEP_N', '0'))


def _configs():
    rng = np.random.RandomState(20260926 + SWEEP_SEED)
    out = []
    ks = [1, 2, 7, 10, 16, 25, 26, 33, 50, 64, 100, 129, 200, 256]
    topks = [1, 5, 10, 11, 20, 24, 25, 50, 52]
    for i in range(SWEEP_N or 72):
        n_items = int(rng.choice([31, 32, 33, 64, 95, 640, 1000, 2049, 5000, 12000]))
        topk = int(rng.choice([t for t in topks if t <= n_items]))
        out.append(dict(seed=i + 1000 * SWEEP_SEED, n_users=int(rng.choice([1, 31, 32, 33, 64, 100, 257, 700, 2100])), n_items=n_items,
                        K=int(rng.choice(ks)), topk=topk, decay=float(rng.choice([0.0, 0.4, 1.0, 1.5])),
                        per_row=int(rng.choice([0, 3, 20, 60, 150])), filter_seen=bool(rng.rand() < 0.8)))
    # round 3: the shape of the sweep is drawn too (a second generator, so that the configurations above stay what they
    # were): threshold bootstrap off / short / default, two-phase off / forced with tiny heads, 1-7 seeded splits, the
    # LDS-shared instance, item chunks of a few tiles (state parked and resumed between launches)
    rng2 = np.random.RandomState(777 + SWEEP_SEED)
    for c in out:
        c['knobs'] = dict(score_boot_tiles=int(rng2.choice([0, 2, 16])), score_head_tiles=int(rng2.choice([0, 0, 1, 3, 8])),
                          score_phase2_splits=int(rng2.choice([1, 3, 7])))
        rng2.choice([0, 0, 1])     # (the draw of the LDS-shared instance, rounds 3-4: it left the product library; the sequence stays)
        c['chunk'] = int(rng2.choice([0, 0, 2, 5]))
    return out


@pytest.mark.parametrize('cfg', _configs(), ids=lambda c: 'u%d_i%d_K%d_k%d_s%d' % (c['n_users'], c['n_items'], c['K'], c['topk'], c['seed']))
def test_random_config_against_brute_force(hip_ops, cfg, monkeypatch, pk_options):
    from polara_amd import scoring
    for k, v in cfg['knobs'].items():
        pk_options(k, v)
    monkeypatch.setattr(hip_ops, 'score_tiles_per_chunk', cfg['chunk'])
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
    if cfg['seed'] % 3 == 0:
        # the same pass as its recorded library calls (scoring.RecordedPass), handing its lists to the host itself: bit for bit
        # the launched pass, on every shape of the sweep (chunked, split, two-phase, empty rows, one user)
        import torch
        pinned = torch.empty((n_users, topk), dtype=torch.int64).pin_memory()
        rp = scoring.RecordedPass(hip_ops, F, T, topk, fs, host_out=pinned)
        for _ in range(2):
            pinned.fill_(-9)
            assert rp.replay() is pinned
            torch.cuda.synchronize()
            assert np.array_equal(pinned.numpy(), ids), cfg
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


def _svd_configs():
    rng = np.random.RandomState(77 + SWEEP_SEED)
    out = []
    for i in range(SWEEP_N or 14):
        n_items = int(rng.choice([12, 40, 130, 500, 1500]))
        n_users = int(rng.choice([n_items, 3 * n_items, 2000]))
        k = int(rng.choice([1, 3, 10, 25, 50]))
        out.append(dict(seed=i + 1000 * SWEEP_SEED, n_users=max(n_users, n_items), n_items=n_items, k=min(k, n_items - 1 if i % 3 else n_items),
                        density=float(rng.choice([0.02, 0.1, 0.5])), kind=str(rng.choice(['plain', 'lowrank', 'dupcols']))))
    return out


@pytest.mark.parametrize('cfg', _svd_configs(), ids=lambda c: 'm%d_n%d_k%d_%s_s%d' % (c['n_users'], c['n_items'], c['k'], c['kind'], c['seed']))
def test_random_svd_build_against_dense_svd(hip_ops, cfg):
    """svd_topk on small random matrices — incl. k = n_items, numerically rank-deficient and duplicated-column
    inputs (clustered / repeated singular values) — against NumPy's dense SVD: singular values to 1e-10 relative,
    and the computed V spans an invariant subspace (||A^T A V - V diag(s^2)|| small)."""
    from polara_amd.solver import svd_topk
    rng = np.random.RandomState(1000 + cfg['seed'])
    m, n, k = cfg['n_users'], cfg['n_items'], cfg['k']
    A = sps.random(m, n, density=cfg['density'], random_state=rng, format='csr', data_rvs=lambda s: rng.randint(1, 6, s).astype(np.float64))
    if cfg['kind'] == 'lowrank':
        r = max(1, min(n // 3, 20))
        B = (rng.rand(m, r) < 0.3) * rng.randint(1, 4, (m, r))
        C = (rng.rand(r, n) < 0.3) * 1.0
        A = sps.csr_matrix((B @ C).astype(np.float64))
    elif cfg['kind'] == 'dupcols' and n >= 4:
        D = A.toarray()
        D[:, n // 2:n // 2 + n // 4] = D[:, :n // 4]            # repeated columns: repeated singular values / null space
        A = sps.csr_matrix(D)
    A.sort_indices()
    if A.nnz == 0:
        pytest.skip('empty draw')
    T = hip_ops.csr(A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data, (m, n))
    _, sigma, V, st = svd_topk(hip_ops, T, k)
    sigma, V = hip_ops.to_host(sigma), hip_ops.to_host(V)
    ref = np.linalg.svd(A.toarray(), compute_uv=False)[:k]
    assert st['converged'], st
    big = ref > 1e-6 * ref[0]
    assert np.allclose(sigma[big], ref[big], rtol=1e-10, atol=0), (sigma[:5], ref[:5])
    # singular values in the numerical null space come out of the Gramian as sqrt(rounding of sigma_1^2): O(1e-8 sigma_1),
    # for this solver as for the reference's ARPACK-on-A^T A (svds)
    assert np.all(np.abs(sigma[~big]) < 1e-6 * ref[0])
    assert np.abs(V.T @ V - np.eye(k)).max() < 1e-9
    G = A.T @ (A @ V)
    assert np.abs(G - V * sigma ** 2).max() <= 1e-9 * ref[0] ** 2
