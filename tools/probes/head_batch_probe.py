"""(round 2) the heaviest users of an activity-ordered pass as a batch of their own (item splits, second stream): pass time
against the size of that head batch.   usage: python tools/probes/head_batch_probe.py [ml20m|s1m] [rank] [topk]"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
topk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, rank)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(V.shape[0], device=order2.device)
V = V[order2].contiguous()
A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
want = scoring.recommend(ops, F, A, topk, True, head_users=0).clone()
for H in (0, 1024, 2048, 4096, 8192, 16384, 32768):
    if H and 4 * H > A.shape[0]:
        continue
    for _ in range(5):
        got = scoring.recommend(ops, F, A, topk, True, head_users=H)
    torch.cuda.synchronize()
    same = bool(torch.equal(got, want))
    t0 = time.perf_counter()
    for _ in range(50):
        scoring.recommend(ops, F, A, topk, True, head_users=H)
    torch.cuda.synchronize()
    print('head %6d users: %.3f ms per pass   identical lists: %s' % (H, 1e3 * (time.perf_counter() - t0) / 50, same))
import numpy as np
for H in (0, 2048, 16384):
    ops.timers = {}
    for _ in range(10):
        scoring.recommend(ops, F, A, topk, True, head_users=H)
    torch.cuda.synchronize()
    rows = {}
    for k, v in ops.timers.items():
        per = len(v) // 10
        for j in range(per):
            rows['%s[%d]' % (k, j)] = round(float(np.median([v[i * per + j][0].elapsed_time(v[i * per + j][1]) for i in range(10)])), 4)
    ops.timers = None
    print('head', H, rows)
