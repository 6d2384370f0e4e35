import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import planted_csr, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd import scoring
ops = HipOps('cuda:0')
mode = sys.argv[1]
n_users, n_items, mean, max_items, rank, topk = (40000, 3000, 40, 300, 12, 10)
c = csr_to_numpy(planted_csr(n_users, n_items, mean, rank, seed=77, min_items=5, max_items=max_items))
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
_, _, V, st = svd_topk(ops, A, rank)
F = scoring.FactorImage(ops, V)
want = scoring.recommend(ops, F, A, topk, True)
cap = scoring.CapturedPass(ops, F, A, topk, True)
print('captured', mode, flush=True)
buf = torch.empty_like(want)
for i in range(4):
    out = cap.replay()
    if mode == 'nosync':
        pass
    elif mode == 'sync':
        torch.cuda.synchronize()
    elif mode == 'copy':
        buf.copy_(out); torch.cuda.synchronize()
    elif mode == 'clone':
        g = out.clone(); torch.cuda.synchronize()
    elif mode == 'equal':
        torch.cuda.synchronize(); ok = torch.equal(out, want)
    elif mode == 'kernel':
        torch.cuda.synchronize(); z = buf + 1; torch.cuda.synchronize()
    elif mode == 'item':
        torch.cuda.synchronize(); v = buf[0, 0].item()
    elif mode == 'ne_any':
        torch.cuda.synchronize(); v = (out != want).any()
        torch.cuda.synchronize()
    elif mode == 'ne_any_item':
        torch.cuda.synchronize(); v = bool((out != want).any().item())
    elif mode == 'eq_own':
        torch.cuda.synchronize(); v = torch.equal(buf, want)
    elif mode == 'alloc':
        torch.cuda.synchronize(); z = torch.empty(1000, device='cuda')
    print('replay', i, flush=True)
torch.cuda.synchronize()
print('final equal', bool(torch.equal(cap.out, want)))
