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


def presharded(comm):
    """On-disk CSR shards (polara_amd/shards.py): rank 0 writes a 3-shard dataset, every rank maps its own run of
    shards and builds on it; global factors must equal those of the single-process model of the whole matrix,
    per-user outputs must be that model's rows of the local users."""
    import tempfile
    import torch.distributed as dist
    from polara_amd import shards
    from polara_amd.data import ShardedArrayData
    from polara_amd.models import ScaledSVD
    from polara_amd.synth import planted_csr, csr_to_numpy
    c = csr_to_numpy(planted_csr(260, 70, mean_items=14, rank=4, seed=21, min_items=1, max_items=40))
    box = [tempfile.mkdtemp(prefix='pkcsr_') if comm.rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    path = os.path.join(box[0], 'ds')
    if comm.rank == 0:
        shards.write_csr_shards(path, c['indptr'], c['indices'], c['values'], c['shape'][1], 3, value_dtype=np.float64)
    comm.barrier()
    local = ShardedArrayData.from_shards(path, comm.rank, comm.world)
    whole = ShardedArrayData.from_shards(path)              # the single-process model of the same dataset
    lo, hi = local.user_range
    out = {}
    for cls, cfg in ((SVDModel, dict(rank=5)), (ScaledSVD, dict(rank=5)), (CoffeeModel, dict(mlrank=(4, 4, 3), seed=1))):
        res = []
        for data, cm in ((local, comm), (whole, None)):
            m = cls(data, ops=NumpyOps(), comm=cm)
            m.verbose = False
            for k, v in cfg.items():
                setattr(m, k, v)
            m.topk = 6
            if cls is CoffeeModel:
                m.build()
            else:
                m.build(return_factors=True)
            res.append((m, m.get_recommendations()))
        (ml, rl), (mw, rw) = res
        f = local.fields
        Vl, Vw = ml.factors[f.itemid], mw.factors[f.itemid]
        Ul, Uw = ml.factors[f.userid], mw.factors[f.userid]
        ok = Ul.shape[0] == hi - lo and np.abs(Vl @ Vl.T - Vw @ Vw.T).max() < 1e-8
        ok = ok and np.abs(Ul @ Vl.T[:Ul.shape[1]] - Uw[lo:hi] @ Vw.T[:Uw.shape[1]]).max() < 1e-8 if cls is not CoffeeModel else ok
        # local lists = the whole model's rows of the local users (both cover the users with >= 1 interaction, in order)
        present = np.flatnonzero(np.diff(c['indptr']) > 0)
        sel = (present >= lo) & (present < hi)
        ok = ok and rl.shape == (int(sel.sum()), 6) and np.array_equal(rl, rw[sel])
        out['presharded_' + cls.__name__] = bool(ok)
    # evaluate() on a sharded dataset is rank-local: the hit counts of the ranks add up to the whole model's
    rng = np.random.RandomState(3)
    hu, hi_ = [], []
    for u in present:
        row = c['indices'][c['indptr'][u]:c['indptr'][u + 1]]
        hu.append(u)
        hi_.append(rng.choice(np.setdiff1d(np.arange(c['shape'][1]), row)))
    hu, hi_ = np.asarray(hu), np.asarray(hi_)
    mine = (hu >= lo) & (hu < hi)
    counts = []
    for data, cm, hold in ((ShardedArrayData.from_shards(path, comm.rank, comm.world, holdout=(hu[mine] - lo, hi_[mine], np.ones(mine.sum()))), comm, None),
                           (ShardedArrayData.from_shards(path, holdout=(hu, hi_, np.ones(len(hu)))), None, None)):
        m = SVDModel(data, ops=NumpyOps(), comm=cm)
        m.verbose = False
        m.rank, m.topk = 5, 20
        m.build()
        h = m.evaluate('hits')
        counts.append(np.array([h.true_positive, h.false_negative], dtype=np.float64))
    import torch
    total = comm.allreduce(torch.from_numpy(counts[0].copy())).numpy()
    out['presharded_evaluate'] = bool(np.array_equal(total, counts[1]) and counts[1].sum() == len(hu) and counts[1][0] > 0)
    return out


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
    out.update(presharded(comm))
    comm.barrier()
    if comm.rank == 0:
        print('DIST_RESULT', out)
    assert all(v for k, v in out.items() if not k.endswith('_allreduces')), out
    assert comm.world == 2 and comm.n_allreduce > 0


if __name__ == '__main__':
    main()
