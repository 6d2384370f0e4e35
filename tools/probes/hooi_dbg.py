import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd import tucker
from polara_amd.synth import make_workload, csr_to_coo_triplets
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml1m')
u, i, v = csr_to_coo_triplets(csr)
levels = np.unique(v); f = np.searchsorted(levels, v)
idx = np.stack([u, i, f], 1).astype(np.int64)
shape = (csr['shape'][0], csr['shape'][1], len(levels))
for mlrank in ((30, 30, 4), (30, 30, 5)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tucker.hooi(ops, idx, None, shape, mlrank, num_iters=25, growth_tol=1e-4, seed=0, verbose=(rep == 2))
        torch.cuda.synchronize(); print(mlrank, rep, round(time.perf_counter() - t0, 4), len(out[4]))
