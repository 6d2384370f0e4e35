import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
ops = HipOps('cuda:0')
csr, cfg = make_workload('s1m', device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(c['indices'], c['shape'][1])
A = ops.csr_relabel_cols(A, rank_of)
n_items = c['shape'][1]
g = torch.Generator(device='cuda:0'); g.manual_seed(0)
for nc in (64, 52, 50, 32, 26, 24, 16, 8):
    V = torch.randn(n_items, nc, generator=g, dtype=torch.float64, device='cuda:0')
    out = ops.empty(A.shape[0], nc)
    for _ in range(2): ops.spmm(A, V, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.spmm(A, V, out=out)
    torch.cuda.synchronize(); print('nc %d: %.3f ms' % (nc, (time.perf_counter() - t0) / 5 * 1e3))
