import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import planted_csr, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd import scoring
ops = HipOps('cuda:0')
which = int(sys.argv[1]); splits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfgs = ((6000, 900, 40, 300, 12, 10), (500, 40, 30, 39, 6, 10), (40000, 3000, 40, 300, 12, 10))
n_users, n_items, mean, max_items, rank, topk = cfgs[which]
ops.score_splits_override = splits
c = csr_to_numpy(planted_csr(n_users, n_items, mean, rank, seed=77, min_items=5, max_items=max_items))
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
_, _, V, st = svd_topk(ops, A, rank)
F = scoring.FactorImage(ops, V)
stats = {}
want = scoring.recommend(ops, F, A, topk, True, stats=stats)
print('cfg', which, 'splits', stats['item_splits'], 'flagged', stats['flagged_users'], flush=True)
cap = scoring.CapturedPass(ops, F, A, topk, True)
print('captured', flush=True)
for i in range(3):
    got = cap.replay().clone(); torch.cuda.synchronize()
    print('replay', i, bool(torch.equal(got, want)), flush=True)
