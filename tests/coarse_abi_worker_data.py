"""The seeded test matrix of the coarse-ABI workers (NumPy only)."""
import numpy as np


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
