"""(round 3, CPU study) Where do the candidate pushes of the sweep happen, and what would a cheap threshold bootstrap buy?

The trace of a two-phase pass (profiles/r03_trace_pass_two_phase.txt) shows the 32-tile head of the ML-20M-shaped sweep
taking 242 us of a 368 us single sweep: the lists are BUILT there (cold threshold: nearly every score is pushed, rings
fill, flush sorts run).  This study counts, per user and per tile, the pushes (score > tau with tau refreshed only when
a lane's ring of 8 fills, as in the kernel) for
  (a) the kernel as it is (tau starts at -inf),
  (b) a bootstrap pass over the first HA tiles that only keeps, per lane, the L largest of the maxima of groups of 16/G of
      the lane's 16 scores per tile (branch-free insertion into a sorted register list): tau0 = the smallest of the
      2 L values of the user's two lanes — at least 2 L >= KC items score that much, so tau0 <= the final KC-th best —
      and the sweep then starts from tau0.
usage: python tools/probes/warmup_study.py   (CPU only, ~1 min)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sps
from scipy.sparse.linalg import svds
from polara_amd.synth import make_workload, csr_to_numpy

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
pick = np.r_[by_act[:1024], by_act[1024::(n_users - 1024) // 1024][:1024]]
E = A[pick] @ V
en = np.linalg.norm(E, axis=1)
KC, RG = 16, 8
lane_items = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)] for h in range(2)])   # items of a tile per lane
print('setup %.0f s' % (time.time() - t0))


def sweep(sc, tau0, max_tiles=400):
    """one user: sc [n_tiles x 32] masked scores.  Returns (pushes per tile, flushes per tile, exit tile)"""
    top = []
    tau = tau0
    ring = [[], []]
    pushes = np.zeros(max_tiles, int); flushes = np.zeros(max_tiles, int)
    for t in range(max_tiles):
        if en_u * tile_bound[t] <= tau:
            return pushes, flushes, t
        for h in range(2):
            for x in sc[t, lane_items[h]]:
                if x > tau:
                    if len(ring[h]) == RG:
                        top = sorted(top + ring[0] + ring[1], reverse=True)[:KC]
                        ring = [[], []]
                        flushes[t] += 1
                        if len(top) == KC:
                            tau = max(tau, top[-1])
                        if not x > tau:
                            continue
                    ring[h].append(x)
                    pushes[t] += 1
    return pushes, flushes, max_tiles


def bootstrap(sc, HA, G, L):
    vals = []
    for h in range(2):
        m = sc[:HA][:, lane_items[h]].reshape(HA, G, 16 // G).max(2).ravel()
        vals += sorted(m, reverse=True)[:L]
    vals = [v for v in vals if np.isfinite(v)]
    return min(vals) if len(vals) >= KC else -np.inf


variants = [('as is', None), ('HA=8 G=4 L=8', (8, 4, 8)), ('HA=16 G=2 L=8', (16, 2, 8)), ('HA=16 G=4 L=8', (16, 4, 8)),
            ('HA=32 G=1 L=8', (32, 1, 8)), ('HA=32 G=2 L=8', (32, 2, 8)), ('HA=32 G=4 L=8', (32, 4, 8)), ('HA=32 G=16 L=8', (32, 16, 8)),
            ('HA=64 G=2 L=8', (64, 2, 8))]
tot = {k: [] for k, _ in variants}
for j, u in enumerate(pick):
    if j % 16:
        continue                                  # 128 users: 64 heavy, 64 spread
    sc = np.pad(E[j] @ V.T, (0, n_tiles * 32 - n_items), constant_values=-np.inf)
    sc[inv[A.indices[A.indptr[u]:A.indptr[u + 1]]]] = -np.inf
    sc = sc.reshape(n_tiles, 32)
    en_u = en[j]
    for name, par in variants:
        tau0 = -np.inf if par is None else bootstrap(sc, *par)
        p, f, ex = sweep(sc, np.nextafter(tau0, -np.inf) if np.isfinite(tau0) else tau0)
        tot[name].append((p.sum(), f.sum(), p[:8].sum(), p[:32].sum(), ex))
for grp, sl in (('heavy', slice(0, 64)), ('spread', slice(64, 128))):
    print('--', grp, 'users: mean per user of  pushes | flushes | pushes in tiles 0-7 | in tiles 0-31 | exit tile')
    for name, _ in variants:
        a = np.array(tot[name][sl], dtype=float)
        print('%-18s %8.1f %7.1f %8.1f %8.1f %7.1f' % ((name,) + tuple(a.mean(0))))
