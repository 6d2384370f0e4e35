"""Sweeps and time of every eigensolve inside one warm HOOI build (ML-1M-shaped tensor): which solves are expensive and
how much the warm start saves.  usage: hooi_eigh_sweeps.py 30,30,5 [cold]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd import tucker
from polara_amd.synth import make_workload, csr_to_coo_triplets
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml1m')
u, i, v = csr_to_coo_triplets(csr)
levels = np.unique(v)
idx = np.stack([u, i, np.searchsorted(levels, v)], 1).astype(np.int64)
shape = (csr['shape'][0], csr['shape'][1], len(levels))
mlrank = tuple(int(x) for x in sys.argv[1].split(','))
warm = not (len(sys.argv) > 2 and sys.argv[2] == 'cold')
log = []
orig = ops.eigh_psd
def eigh(S, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(S, *a, **k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    log.append((S.shape[0], int(ops._info[0].item()), int(ops._info[1].item()), round(dt * 1e3, 3)))
    return out
tucker.hooi(ops, idx, None, shape, mlrank, num_iters=25, growth_tol=1e-4, seed=0, warm_start=warm)
ops.eigh_psd = eigh
out = tucker.hooi(ops, idx, None, shape, mlrank, num_iters=25, growth_tol=1e-4, seed=0, warm_start=warm)
print('warm_start', warm, 'iterations', len(out[4]))
print('(n, sweeps, converged, ms) per solve:')
for k in range(0, len(log), 3):
    print('  ', log[k:k + 3])
print('total eigh ms', round(sum(l[3] for l in log), 2))
