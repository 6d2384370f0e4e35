"""Worker for the real-RCCL test: one process per GPU, backend "nccl" (= RCCL over xGMI), the product HIP kernels.
What is under test: the user-sharded build (Gramian all-reduce of device buffers) + collective-free scoring + the
typed result gather give the single-GPU lists, and bench.py's timing path (barrier, MAX over ranks) runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

from conftest import load_golden, GoldenData
from polara_amd.dist import init_from_env
from polara_amd.models import SVDModel
from polara_amd.ops import HipOps


def main():
    comm = init_from_env()                     # nccl, LOCAL_RANK -> its own GPU
    assert torch.distributed.get_backend() == 'nccl'
    ops = HipOps('cuda:%d' % torch.cuda.current_device())
    ok = {}
    for name in ('svd_warm', 'svd_known'):
        g = load_golden(name)
        m = SVDModel(GoldenData(g), ops=ops, comm=comm)
        m.verbose = False
        m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
        m.build(return_factors=True)
        recs = m.get_recommendations()
        notie = g['boundary_gap'] > 0
        ok[name] = bool(np.array_equal(recs[notie], g['recs'][notie])
                        and np.allclose(m.factors['singular_values'], g['sigma'], rtol=1e-9))
    t = torch.tensor([float(comm.rank)], dtype=torch.float64, device=ops.device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ok['max_reduce'] = float(t.item()) == comm.world - 1
    comm.barrier()
    if comm.rank == 0:
        print('DIST_NCCL_RESULT', ok, 'allreduces', comm.n_allreduce, 'bytes', comm.bytes_reduced)
    assert all(ok.values()), ok


if __name__ == '__main__':
    main()
