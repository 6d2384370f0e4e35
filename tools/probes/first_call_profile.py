"""cProfile of SVDModel.build() and of the FIRST get_recommendations() of a model on the ML-20M-shaped matrix (the plugin
surface as a user drives it): where the host time of the one-off preparation goes."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.data import ArrayData
from polara_amd.models import SVDModel
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_coo_triplets
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m')
u, i, v = csr_to_coo_triplets(csr)
n_users, n_items = csr['shape']
hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
def fresh():
    d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
    m = SVDModel(d, ops=ops)
    m.verbose = False
    m.rank, m.topk = 50, 10
    return m
m = fresh(); m.build(); m.get_recommendations(); torch.cuda.synchronize()      # process warm-up
for what in ('build', 'first get_recommendations'):
    m = fresh()
    if what != 'build':
        m.build()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    (m.build if what == 'build' else m.get_recommendations)()
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(32)
    print('=====', what, '%.1f ms' % (dt * 1e3))
    print('\n'.join(l for l in s.getvalue().splitlines()[4:] if l.strip())[:3800])
