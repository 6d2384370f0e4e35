"""(round 2) exit tile of the user groups of a pass against their position in the launch (users grouped by activity):
do the slow groups start first?   usage: python tools/probes/exit_by_group.py [ml20m|s1m]"""
import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, 50)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(V.shape[0], device=order2.device)
V = V[order2].contiguous()
A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
for order_users in (True, False):
    scoring.recommend(ops, F, A, 10, True, order_users=order_users)
    torch.cuda.synchronize()
    ex = ops.score_exit_tiles(A.shape[0], 1).flatten().cpu().numpy().astype(np.int64)
    n = len(ex)
    dec = [int(ex[i * n // 20:(i + 1) * n // 20].mean()) for i in range(20)]
    mx = [int(ex[i * n // 20:(i + 1) * n // 20].max()) for i in range(20)]
    print('order_users=%s groups=%d  mean exit tile per 5%% of the launch order: %s' % (order_users, n, dec))
    print('                          max  exit tile per 5%% of the launch order: %s' % mx)
