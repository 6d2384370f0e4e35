"""Worker for the world_size-2 gloo test (launched by torch.distributed.run).  Uses the TEST-ONLY
NumPy double of the device ops: what is under test is the sharding + all-reduce orchestration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np

from conftest import load_golden, GoldenData
from numpy_ops import NumpyOps
from polara_amd.dist import init_from_env
from polara_amd.models import SVDModel, CoffeeModel


def main():
    comm = init_from_env(backend='gloo')
    out = {}
    for name in ('svd_warm', 'svd_known'):
        g = load_golden(name)
        m = SVDModel(GoldenData(g), ops=NumpyOps(), comm=comm)
        m.verbose = False
        m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
        m.build(return_factors=True)
        recs = m.get_recommendations()
        notie = g['boundary_gap'] > 0
        U = m.factors[m.data.fields.userid]
        V = m.factors[m.data.fields.itemid]
        ok = (np.array_equal(recs[notie], g['recs'][notie])
              and np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9)
              and U.shape == (int(g['train_shape'][0]), int(g['rank']))
              and np.allclose(U.T @ U, np.eye(U.shape[1]), atol=1e-8)
              and np.allclose(V @ V.T, g['V'] @ g['V'].T, atol=1e-8))
        out[name] = bool(ok)
        out[name + '_allreduces'] = comm.n_allreduce
    for name in ('coffee_small', 'coffee_warm'):
        # sharded HOOI: user-mode factor rows stay local, item/feedback-mode TTMs are all-reduced
        g = load_golden(name)
        m = CoffeeModel(GoldenData(g), ops=NumpyOps(), comm=comm)
        m.verbose = False
        m.mlrank, m.topk, m.seed = tuple(int(x) for x in g['mlrank']), int(g['topk']), int(g['seed'])
        m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
        m.build()
        f = m.data.fields
        proj_ok = all(np.abs(m.factors[k] @ m.factors[k].T - ref @ ref.T).max() < 1e-8
                      for k, ref in ((f.userid, g['u0']), (f.itemid, g['u1']), (f.feedback, g['u2'])))
        notie = g['boundary_gap'] > 0
        out[name] = bool(proj_ok and np.allclose(m.core_norm_trace, g['core_norm_trace'], rtol=1e-9)
                          and np.array_equal(m.get_recommendations()[notie], g['recs'][notie]))
    comm.barrier()
    if comm.rank == 0:
        print('DIST_RESULT', out)
    assert all(v for k, v in out.items() if not k.endswith('_allreduces')), out
    assert comm.world == 2 and comm.n_allreduce > 0


if __name__ == '__main__':
    main()
