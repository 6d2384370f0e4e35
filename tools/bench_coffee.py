"""CoFFee (HOOI) build + scoring on an ML-1M-shaped tensor (BASELINE.json configs[3]) — timings for
DESIGN.md.  mlrank (30,30,4): the reference raises for r2 == n_feedback (lib/tensor.py:79), so
r2 = 4 is the largest rank it can run.  The CPU side is the oracle (vectorised dttm restatement)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from polara_amd.data import ArrayData
from polara_amd.models import CoffeeModel
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_coo_triplets, csr_to_numpy


def main():
    cpu = '--cpu' in sys.argv
    ops = HipOps('cuda:0')
    csr, cfg = make_workload('ml1m', device='cuda:0')
    u, i, v = csr_to_coo_triplets(csr)
    c = csr_to_numpy(csr)
    n_users, n_items = c['shape']
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    out = {}
    # (30, 30, 5): BASELINE.json configs[3]; r2 == number of rating levels — the reference raises there
    for mlrank in ((13, 10, 2), (30, 30, 4), (30, 30, 5)):
        d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
        m = CoffeeModel(d, ops=ops)
        m.verbose = False
        m.mlrank, m.seed, m.topk = mlrank, 0, 10
        m.build()                      # warm-up (allocations, plans)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.build()
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        recs = m.get_recommendations()
        t_rec = time.perf_counter() - t0
        res = dict(build_s=t_build, iterations=len(m.core_norm_trace), recommend_s=t_rec,
                   core_norm=m.core_norm_trace[-1], nnz=len(v))
        if cpu and mlrank[2] < 5:
            from oracle import polara_oracle as orc
            idx, val, shp = d.to_coo(tensor_mode=True)
            trace = []
            t0 = time.perf_counter()
            o0, o1, o2, og = orc.hooi(idx, val, shp, mlrank, growth_tol=m.growth_tol, num_iters=m.num_iters,
                                      seed=0, trace=trace)
            res['cpu_build_s'] = time.perf_counter() - t0
            res['cpu_iterations'] = len(trace)
            res['core_norm_rel_diff'] = abs(trace[-1] - m.core_norm_trace[-1]) / trace[-1]
            a = m.factors[d.fields.itemid]
            res['item_projector_diff'] = float(np.abs(a @ a.T - o1 @ o1.T).max())
        out[str(mlrank)] = res
    print(json.dumps(out))


if __name__ == '__main__':
    main()
