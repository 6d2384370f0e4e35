"""(round 2, CPU study for the next round) What would a two-phase sweep buy?  On the ML-20M-shaped catalogue (all 26 744
items, a sample of the users) simulate, per group of 32 users in activity order, the tile at which the group leaves
 (a) the single sweep (threshold = running KC-th best unseen score, refreshed every tile: an optimistic version of the
     kernel, whose threshold is as fresh as its last flush),
 (b) S independent item splits (tiles h, h+S, ...: each split prunes against ITS KC-th best),
 (c) a head sweep over T0 tiles, then S splits that all start from the head's threshold,
and report the dependent chain (tile steps on the critical path of a group) and the tile-waves executed.
usage: python tools/probes/two_phase_study.py [n_users_sample]   (CPU only; ~2 min)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sps
from scipy.sparse.linalg import svds
from polara_amd.synth import make_workload, csr_to_numpy

n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t0 = time.time()
csr, cfg = make_workload('ml20m', device='cpu')
c = csr_to_numpy(csr)
A = sps.csr_matrix((c['values'].astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
n_users, n_items = A.shape
_, s, vt = svds(A, k=50)
V = np.ascontiguousarray(vt.T)
norms = np.linalg.norm(V, axis=1)
order = np.argsort(-norms, kind='stable')
V, norms = V[order], norms[order]
inv = np.empty(n_items, np.int64); inv[order] = np.arange(n_items)
n_tiles = -(-n_items // 32)
tile_bound = np.maximum.accumulate(np.r_[norms, np.zeros(n_tiles * 32 - n_items)].reshape(n_tiles, 32).max(1)[::-1])[::-1]
counts = np.diff(A.indptr)
by_act = np.argsort(-counts, kind='stable')
# the heaviest 2048 users and a spread of the rest, in activity order, as whole groups of 32
pick = np.r_[by_act[:2048], by_act[2048::max(1, (n_users - 2048) // (n_sample - 2048))][:n_sample - 2048]]
pick = pick[:len(pick) // 32 * 32]
E = A[pick] @ V
en = np.linalg.norm(E, axis=1)
KC = 16
print('setup %.0f s; %d users in %d groups' % (time.time() - t0, len(pick), len(pick) // 32))


def exit_tiles(scores, seen, en_g, tiles, tau0=None):
    """tiles: the tile indices this sweep visits, in order.  Returns (steps taken, final tau per user)."""
    top = np.full((scores.shape[0], KC), -np.inf)
    tau = np.full(scores.shape[0], -np.inf) if tau0 is None else tau0.copy()
    steps = 0
    for t in tiles:
        if not (en_g * tile_bound[t] > tau).any():
            break
        steps += 1
        sc = scores[:, 32 * t:32 * t + 32].copy()
        sc[seen[:, 32 * t:32 * t + 32]] = -np.inf
        top = -np.sort(-np.concatenate([top, sc], 1), 1)[:, :KC]
        tau = np.maximum(tau, top[:, KC - 1])
    return steps, tau


res = {k: [] for k in ('single', 'split2', 'split4', 'head32_split4', 'head64_split4', 'head64_split2')}
for g in range(len(pick) // 32):
    rows = slice(32 * g, 32 * g + 32)
    Sg = E[rows] @ V.T
    Sg = np.pad(Sg, ((0, 0), (0, n_tiles * 32 - n_items)), constant_values=-np.inf)
    seen = np.zeros_like(Sg, dtype=bool)
    for r, u in enumerate(pick[rows]):
        seen[r, inv[A.indices[A.indptr[u]:A.indptr[u + 1]]]] = True
    eg = en[rows]
    st, _ = exit_tiles(Sg, seen, eg, range(n_tiles))
    res['single'].append((st, st))
    for S in (2, 4):
        steps = [exit_tiles(Sg, seen, eg, range(h, n_tiles, S))[0] for h in range(S)]
        res['split%d' % S].append((max(steps), sum(steps)))
    for T0, S in ((32, 4), (64, 4), (64, 2)):
        h_steps, tau = exit_tiles(Sg, seen, eg, range(min(T0, n_tiles)))
        if h_steps < T0:
            res['head%d_split%d' % (T0, S)].append((h_steps, h_steps))
        else:
            steps = [exit_tiles(Sg, seen, eg, range(T0 + h, n_tiles, S), tau0=tau)[0] for h in range(S)]
            res['head%d_split%d' % (T0, S)].append((T0 + max(steps), T0 + sum(steps)))
n_heavy = 2048 // 32
for k, v in res.items():
    v = np.array(v)
    print('%-14s chain: heavy groups mean %5.1f max %4d | other groups mean %5.1f max %4d | tile-waves total %7d (x%.2f of single)' % (
        k, v[:n_heavy, 0].mean(), v[:n_heavy, 0].max(), v[n_heavy:, 0].mean(), v[n_heavy:, 0].max(), v[:, 1].sum(),
        v[:, 1].sum() / np.array(res['single'])[:, 1].sum()))
