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
    # Round 6: the block Lanczos build of a user-sharded matrix with the steps INSIDE the library — every rank computes the
    # products of its rows (pk_lanczos_products), W is summed over the ranks (ONE all-reduce of n_items x 16 per step), every
    # rank runs the same orthogonalisation (pk_lanczos_orth) and the same looks on the replicated projected matrix: the
    # factors of the one-process build of the whole matrix; also with the late exchanges rounded to fp32 (exchange='relaxed')
    from polara_amd.csr import nnz_balanced_row_partition
    from polara_amd.solver import svd_topk
    from polara_amd.synth import planted_csr
    m = planted_csr(30000, 4000, 60, 40, levels=5, seed=11, min_items=8, max_items=400)
    c = csr_to_numpy(m)
    whole = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    k = 24
    _, s1, V1, st1 = svd_topk(ops, whole, k, method='lanczos', krylov_block=16)
    bounds = nnz_balanced_row_partition(c['indptr'], comm.world)
    lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
    sub = slice(int(c['indptr'][lo]), int(c['indptr'][hi]))
    part = ops.csr(c['indptr'][lo:hi + 1] - c['indptr'][lo], c['indices'][sub], c['values'][sub], (hi - lo, c['shape'][1]))
    n0 = comm.n_allreduce
    U2, s2, V2, st2 = svd_topk(ops, part, k, comm=comm, method='lanczos', krylov_block=16, want_u=True)
    n1 = comm.n_allreduce
    s1h, s2h, V1h, V2h = (ops.to_host(t) for t in (s1, s2, V1, V2))
    ok['library_recurrence_sharded'] = bool(
        st1['recurrence'] == 'library' and st2['recurrence'] == 'library' and not st2['items_sharded'] and st2['converged']
        and st2['verified_rel_residual'] <= 1e-12 and np.allclose(s1h, s2h, rtol=1e-10)
        and np.abs(V1h @ V1h[:300].T - V2h @ V2h[:300].T).max() < 1e-8
        and n1 - n0 == st2['gramian_steps'] + 1 and U2.shape == (hi - lo, k))      # one all-reduce per product (steps + verification) + the entry count of the plan: nothing else
    ok['library_recurrence_sharded_steps'] = (st1['lanczos_steps'], st2['lanczos_steps'], n1 - n0)
    _, s3, V3, st3 = svd_topk(ops, part, k, comm=comm, method='lanczos', krylov_block=16, monitor_lag=1, first_look=4, exchange='relaxed')
    s3h, V3h = ops.to_host(s3), ops.to_host(V3)
    ok['library_recurrence_relaxed_exchange'] = bool(
        st3['recurrence'] == 'library' and st3['converged'] and st3['verified_rel_residual'] <= 1e-12
        and st3.get('exchange_relaxed_from') is not None and np.allclose(s1h, s3h, rtol=1e-10)
        and np.abs(V1h @ V1h[:300].T - V3h @ V3h[:300].T).max() < 1e-8)
    comm.barrier()
    if comm.rank == 0:
        print('DIST_GPU_RESULT', ok)
    assert all(v for k, v in ok.items() if isinstance(v, bool)), ok
    assert ok['ml1m_recs_equal_frac'] > 0.9995, ok
    assert comm.world == 2 and comm.n_allreduce > 0


if __name__ == '__main__':
    main()
