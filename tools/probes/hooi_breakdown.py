"""Where the wall time of a warm CoFFee build goes: the two CSR unfoldings, then per iteration the three mode
products + small SVDs, with a device sync after each part (so the parts add up; the sum exceeds the un-synced build)."""
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
f = np.searchsorted(levels, v)
idx = np.stack([u, i, f], 1).astype(np.int64)
shape = (csr['shape'][0], csr['shape'][1], len(levels))
mlrank = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '30,30,4').split(','))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = tucker.hooi(ops, idx, None, shape, mlrank, num_iters=25, growth_tol=1e-4, seed=0)
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
uf = tucker.Unfoldings(ops, idx, None, shape)
torch.cuda.synchronize(); t_unf = time.perf_counter() - t0
print(json.dumps(dict(mlrank=mlrank, build_s=round(t_all, 4), iterations=len(out[4]), unfoldings_s=round(t_unf, 4))))

# the model-level build around it (data object -> device coordinates -> hooi -> factors on the host)
from polara_amd.data import ArrayData
from polara_amd.models import CoffeeModel
n_users, n_items = csr['shape']
hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
m = CoffeeModel(d, ops=ops)
m.verbose = False
m.mlrank, m.seed = mlrank, 0
ts = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.build()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps(dict(model_build_s=[round(t, 4) for t in ts], training_time_s=round(m.training_time[-1], 4))))
