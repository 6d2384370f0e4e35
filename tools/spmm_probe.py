import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
ops = HipOps('cuda:0')
csr, cfg = make_workload('s1m', device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
out = {}
def timeit(A, X, n=5):
    ops.spmm(A, X); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.spmm(A, X)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
At = A.T
for nc in (64, 32, 16, 8):
    Y = ops.randn(n_users, nc, 1)
    out['AtY_nc%d' % nc] = timeit(At, Y)
    # the same gather volume but from a strided view of a wide Y (footprint stays 512 MB)
Yw = ops.randn(n_users, 64, 1)
out['AtY_nc32_of64'] = timeit(At, Yw[:, :32])
# users sorted by activity: does the gather locality change?
print(json.dumps(out))
