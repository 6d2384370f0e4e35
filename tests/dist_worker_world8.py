"""Worker for the world_size-8 gloo test (launched by torch.distributed.run; the shape of the driver's 8-GPU job on
CPU).  TEST-ONLY NumPy double of the device ops: what is under test is the orchestration at EIGHT ranks — an item
count that is not a multiple of 8 (203 = 8 * 26 - 5: the last rank owns 21 real rows + 5 padding rows of every
item-side block), a test set of 5 users (nnz-balanced partition: most ranks score ZERO users and still take part in
the result gather), every rank returning the single-process model's factors and lists, and the collective counts
bench.py prints in its line (one all-gather + one reduce-scatter per Gramian step, none in scoring)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np

from numpy_ops import NumpyOps
from polara_amd.csr import nnz_balanced_row_partition
from polara_amd.data import ArrayData
from polara_amd.dist import init_from_env
from polara_amd.models import SVDModel
from polara_amd.synth import planted_csr, csr_to_coo_triplets, csr_to_numpy


def main():
    comm = init_from_env(backend='gloo')
    assert comm.world == 8
    n_users, n_items = 700, 203
    csr = planted_csr(n_users, n_items, 18, 6, seed=8, min_items=3, max_items=60)
    u, i, v = csr_to_coo_triplets(csr)
    c = csr_to_numpy(csr)
    # five test users (new ids 0..4) with the profiles of training users 10, 11, 300, 301, 699
    src = [10, 11, 300, 301, 699]
    tu, ti, tv = [], [], []
    for new, old in enumerate(src):
        sl = slice(int(c['indptr'][old]), int(c['indptr'][old + 1]))
        tu.append(np.full(sl.stop - sl.start, new)), ti.append(c['indices'][sl]), tv.append(c['values'][sl])
    test = (np.concatenate(tu), np.concatenate(ti).astype(np.int64), np.concatenate(tv).astype(np.float64))
    out = {}
    res = []
    for cm in (comm, None):
        d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, test=test, warm_start=True)
        m = SVDModel(d, ops=NumpyOps(), comm=cm)
        m.verbose = False
        m.rank, m.topk = 6, 7
        before = (comm.n_allgather, comm.n_reduce_scatter)
        m.build(return_factors=True)
        steps = (comm.n_allgather - before[0], comm.n_reduce_scatter - before[1])
        before_score = (comm.n_allreduce, comm.n_reduce_scatter)
        recs = m.get_recommendations()
        res.append((m, recs, steps, (comm.n_allreduce - before_score[0], comm.n_reduce_scatter - before_score[1])))
    (ms, rs, steps_s, score_coll), (m1, r1, _, _) = res
    f = ms.data.fields
    Vs, V1 = ms.factors[f.itemid], m1.factors[f.itemid]
    Us, U1 = ms.factors[f.userid], m1.factors[f.userid]
    st = ms.build_stats
    out['sigma'] = bool(np.allclose(ms.factors['singular_values'], m1.factors['singular_values'], rtol=1e-10))
    out['projector'] = bool(np.abs(Vs @ Vs.T - V1 @ V1.T).max() < 1e-8)
    out['user_factors'] = bool(Us.shape == (n_users, 6) and np.abs(Us @ Vs.T - U1 @ V1.T).max() < 1e-8)
    out['lists'] = bool(rs.shape == (5, 7) and np.array_equal(rs, r1))
    out['items_sharded'] = bool(st['items_sharded'] and st['item_rows_per_rank'] == 26)
    out['one_reduce_scatter_per_gramian_step'] = bool(steps_s[1] == st['gramian_steps'] and steps_s[0] >= st['gramian_steps'])
    out['no_reduction_in_scoring'] = bool(score_coll == (0, 0))
    # the partition of the five test users: at least one rank owns none of them
    tip = np.r_[0, np.cumsum(np.bincount(test[0], minlength=5))]
    b = nnz_balanced_row_partition(tip.astype(np.int64), 8)
    out['some_rank_scores_nobody'] = bool((np.diff(b) == 0).any())
    comm.barrier()
    if comm.rank == 0:
        print('WORLD8_RESULT', out, 'gramian steps', st['gramian_steps'])
    assert all(out.values()), out


if __name__ == '__main__':
    main()
