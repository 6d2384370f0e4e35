"""One warm CoFFee / HOOI build on the ML-1M-shaped tensor at the multilinear rank given on the command line (run under
rocprofv3 --kernel-trace --stats: the per-kernel table behind DESIGN.md's HOOI time breakdown, BASELINE configs[3])."""
import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.data import ArrayData
from polara_amd.models import CoffeeModel
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_coo_triplets
mlrank = tuple(int(x) for x in sys.argv[1].split(','))
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml1m')
u, i, v = csr_to_coo_triplets(csr)
n_users, n_items = csr['shape']
hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
m = CoffeeModel(d, ops=ops)
m.verbose = False
m.mlrank, m.seed, m.topk = mlrank, 0, 10
m.build()
torch.cuda.synchronize()
t0 = time.perf_counter()
m.build()
torch.cuda.synchronize()
t1 = time.perf_counter()
recs = m.get_recommendations()
t2 = time.perf_counter()
print(json.dumps(dict(mlrank=mlrank, build_s=t1 - t0, iterations=len(m.core_norm_trace), recommend_s=t2 - t1)))
