"""(round 2) cycle breakdown inside the candidate sweep (library built with PK_SCORE_PROFILE=1 PK_FAST_BUILD=1) for the
HEAVIEST users of the ML-20M-shaped pass against typical ones: which part of a tile makes the slow chains slow?"""
import os, sys, json, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
from polara_amd.solver import svd_topk
from polara_amd import scoring
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m', device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, _ = popularity_order(None, n_items, counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, 50)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(n_items, device=order2.device)
V = V[order2].contiguous(); A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
P, perm = A.by_activity()
buf = (ctypes.c_ulonglong * 8)()
ops.lib.pk_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ('kernel', 'flush', 'walk', 'push_incl_flush', 'prologue', 'epilogue', 'n_flush', 'tiles')
for tag, lo, hi in (('heaviest 2048 users', 0, 2048), ('users 60000..62048 of the activity order', 60000, 62048), ('all users', 0, n_users)):
    T = ops.csr_rows(P, lo, hi)
    for _ in range(3): scoring.recommend(ops, F, T, 10, True, order_users=False)
    torch.cuda.synchronize()
    ops.lib.pk_debug_profile(None, 1)
    n = 5
    for _ in range(n): scoring.recommend(ops, F, T, 10, True, order_users=False)
    torch.cuda.synchronize()
    ops.lib.pk_debug_profile(buf, 0)
    d = {k: int(v) / n for k, v in zip(names, buf)}
    tw = max(d['tiles'], 1)
    print('%-42s tile-waves %8d  cycles per tile-wave %6.0f = flush %5.0f + walk %5.0f + push (no flush) %5.0f + rest %5.0f;  flushes per tile-wave %.2f, nnz per user %.0f' % (
        tag, tw, d['kernel'] / tw, d['flush'] / tw, d['walk'] / tw, (d['push_incl_flush'] - d['flush']) / tw,
        (d['kernel'] - d['walk'] - d['push_incl_flush']) / tw, d['n_flush'] / tw, float(T.indices.numel()) / (hi - lo)))
