import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
mode = sys.argv[1]
if mode == 'torch_only':
    a = torch.randn(1000, 1000, device='cuda'); b = torch.randn(1000, 1000, device='cuda')
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        c = (a @ b).relu().sum(dim=1)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c = (a @ b).relu().sum(dim=1)
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); z = a + 1; torch.cuda.synchronize(); print('replay', i, float(c.sum()), flush=True)
    sys.exit(0)
from polara_amd.ops import HipOps
from polara_amd.synth import planted_csr, csr_to_numpy
ops = HipOps('cuda:0')
c = csr_to_numpy(planted_csr(40000, 3000, 40, 12, seed=77, min_items=5, max_items=300))
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
X = ops.randn(3000, 16, 1)
_ = A.plan
ref = ops.spmm(A, X); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ops.spmm(A, X)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = ops.spmm(A, X)
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    if mode == 'spmm_then_torch_kernel':
        z = X + 1; torch.cuda.synchronize()
    elif mode == 'spmm_then_own_kernel':
        z = ops.spmm(A, X); torch.cuda.synchronize()
    print('replay', i, bool(torch.equal(out.cpu(), ref.cpu())), flush=True)
