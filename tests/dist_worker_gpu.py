"""Worker for the -m gpu two-rank test: both ranks run the REAL HIP kernels (sharing cuda:0, the
GPU box has one device), the Gramian all-reduce goes through gloo staged via the host.  What is
under test: the user-sharded build + scoring on the product backend gives the single-GPU result."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

from conftest import load_golden, GoldenData
from polara_amd.data import ArrayData
from polara_amd.dist import init_from_env
from polara_amd.models import SVDModel, CoffeeModel
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_coo_triplets, csr_to_numpy


def main():
    os.environ['LOCAL_RANK'] = '0'          # every rank on the only GPU
    comm = init_from_env(backend='gloo')
    ops = HipOps('cuda:0')
    ok = {}
    for name in ('svd_warm', 'svd_known', 'svd_fewunseen'):
        g = load_golden(name)
        m = SVDModel(GoldenData(g), ops=ops, comm=comm)
        m.verbose = False
        m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
        m.build(return_factors=True)
        recs = m.get_recommendations()
        notie = g['boundary_gap'] > 0
        U = m.factors[m.data.fields.userid]
        ok[name] = bool(np.array_equal(recs[notie], g['recs'][notie])
                        and np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9)
                        and U.shape[0] == int(g['train_shape'][0])
                        and np.allclose(U.T @ U, np.eye(U.shape[1]), atol=1e-8))
    for name in ('coffee_small', 'coffee_warm'):
        # sharded HOOI: user-mode factor rows stay local, item/feedback-mode TTMs are all-reduced
        g = load_golden(name)
        m = CoffeeModel(GoldenData(g), ops=ops, comm=comm)
        m.verbose = False
        m.mlrank, m.topk, m.seed = tuple(int(x) for x in g['mlrank']), int(g['topk']), int(g['seed'])
        m.num_iters, m.growth_tol = int(g['num_iters']), float(g['growth_tol'])
        m.build()
        f = m.data.fields
        proj_ok = all(np.abs(m.factors[k] @ m.factors[k].T - ref @ ref.T).max() < 1e-8
                      for k, ref in ((f.userid, g['u0']), (f.itemid, g['u1']), (f.feedback, g['u2'])))
        notie = g['boundary_gap'] > 0
        ok[name] = bool(proj_ok and np.allclose(m.core_norm_trace, g['core_norm_trace'], rtol=1e-9)
                          and np.array_equal(m.get_recommendations()[notie], g['recs'][notie]))
    # ML-1M-shaped: sharded result must equal the single-process result bit for bit in the recs
    csr, cfg = make_workload('ml1m')
    c = csr_to_numpy(csr)
    u, i, v = csr_to_coo_triplets(csr)
    n_users = c['shape'][0]
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    d = ArrayData((u, i, v), n_users=n_users, n_items=c['shape'][1], holdout=hold, warm_start=False)
    sharded = SVDModel(d, ops=ops, comm=comm)
    single = SVDModel(d, ops=ops)
    for m in (sharded, single):
        m.verbose = False
        m.rank, m.topk = cfg['rank'], cfg['topk']
        m.build()
    r_sh, r_si = sharded.get_recommendations(), single.get_recommendations()
    ok['ml1m_sigma'] = bool(np.allclose(sharded.factors['singular_values'], single.factors['singular_values'], rtol=1e-11))
    ok['ml1m_recs_equal_frac'] = float((r_sh == r_si).all(axis=1).mean())
    ok['allreduces'] = comm.n_allreduce
    comm.barrier()
    if comm.rank == 0:
        print('DIST_GPU_RESULT', ok)
    assert all(v for k, v in ok.items() if isinstance(v, bool)), ok
    assert ok['ml1m_recs_equal_frac'] > 0.9995, ok
    assert comm.world == 2 and comm.n_allreduce > 0


if __name__ == '__main__':
    main()
