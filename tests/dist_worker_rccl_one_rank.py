"""Worker for the one-GPU RCCL test: backend "nccl" (= RCCL) with a group of ONE rank and TorchComm's
`exercise_collectives` switch, so that every collective of polara_amd/dist.py is actually issued to the library — the
all-reduce of device buffers, `all_gather_into_tensor`, `reduce_scatter_tensor`, the typed result gather, the barrier — on
the shapes the sharded solver uses, and the item-sharded solver itself runs through them.  With one rank every collective
is the identity: results must equal the communicator-free run bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

from polara_amd.dist import init_from_env, TorchComm
from polara_amd.ops import HipOps
from polara_amd.solver import svd_topk
from polara_amd.synth import planted_csr


def main():
    init_from_env()
    assert torch.distributed.get_backend() == 'nccl' and torch.distributed.get_world_size() == 1
    comm = TorchComm(exercise_collectives=True)
    ops = HipOps('cuda:%d' % torch.cuda.current_device())
    dev = ops.device
    ok = {}
    g = torch.Generator(device='cpu').manual_seed(1)
    X = torch.randn(3001, 64, generator=g, dtype=torch.float64).to(dev)
    ok['allreduce_f64'] = bool(torch.equal(comm.allreduce(X.clone()), X))
    c = torch.arange(977, dtype=torch.int64, device=dev)
    ok['allreduce_i64'] = bool(torch.equal(comm.allreduce(c.clone()), c))
    ok['all_gather_rows'] = bool(torch.equal(comm.all_gather_rows(X), X))
    ok['reduce_scatter_rows'] = bool(torch.equal(comm.reduce_scatter_rows(X.clone(), X.shape[0]), X))
    recs = np.arange(500 * 10, dtype=np.int64).reshape(500, 10)
    ok['gather_rows'] = bool(np.array_equal(comm.gather_rows(recs, 500, 10), recs))
    comm.barrier()
    # the item-sharded solver through those calls against the communicator-free run
    csr = planted_csr(4000, 900, 30, rank=12, seed=5)
    A = ops.csr(np.asarray(csr['indptr']), np.asarray(csr['indices']), np.asarray(csr['values']), csr['shape'])
    _, s0, V0, st0 = svd_topk(ops, A, 10, seed=3)
    n0 = comm.n_allreduce
    _, s1, V1, st1 = svd_topk(ops, A, 10, seed=3, comm=comm)
    ok['solver_sharded_layout_used'] = bool(st1['items_sharded']) and comm.n_allgather > 0 and comm.n_reduce_scatter > 0 and comm.n_allreduce > n0
    ok['solver_same_bits'] = bool(torch.equal(s0, s1) and torch.equal(V0, V1)) and st0['gramian_steps'] == st1['gramian_steps']
    # the asynchronous forms (solver.ItemRows.product: the second column panel's products run next to the first one's sum):
    # started, a kernel enqueued meanwhile, waited for — and the solver with the two-panel exchange forced on this one rank
    h = comm.allreduce_start(X.clone())
    G = ops.gram(X)
    ok['allreduce_start'] = bool(type(h).__name__ == '_Pending' and torch.equal(h.wait(), X)) and bool(torch.isfinite(G).all())
    h1 = comm.reduce_scatter_rows_start(X.clone(), X.shape[0])
    h2 = comm.reduce_scatter_rows_start(X[:, :32].contiguous(), X.shape[0], count=False)
    ok['reduce_scatter_rows_start'] = bool(torch.equal(h1.wait(), X) and torch.equal(h2.wait(), X[:, :32]))
    p0 = comm.n_panel_exchanges
    for shard_items in (True, False):
        _, s2, V2, st2 = svd_topk(ops, A, 18, seed=3, comm=comm, shard_items=shard_items, exchange_overlap='force')      # block 32: two panels of 16
        _, s3, V3, st3 = svd_topk(ops, A, 18, seed=3)
        ok['solver_two_panel_exchange_%s' % ('sharded' if shard_items else 'replicated')] = bool(
            torch.allclose(s2, s3, rtol=1e-12) and float((V2 @ V2.T - V3 @ V3.T).abs().max()) < 1e-10 and st2['gramian_steps'] == st3['gramian_steps'])
    ok['panel_exchanges_started'] = comm.n_panel_exchanges - p0 > 0
    # the DEFAULT form of a user-sharded build: the library's step in its two halves (pk_lanczos_products / pk_lanczos_orth) with
    # RCCL's all-reduce of the block between them, the verification product summed the same way — in a group of one the sums are
    # the identity and the halves are the functions the single call runs: the same bits as the communicator-free build
    split = TorchComm(split_step=True)
    _, sl, Vl, stl = svd_topk(ops, A, 10, seed=3, method='lanczos')
    _, s4, V4, st4 = svd_topk(ops, A, 10, seed=3, comm=split, method='lanczos')
    ok['split_step_library_recurrence'] = st4.get('recurrence') == 'library' and not st4['items_sharded'] and stl.get('recurrence') == 'library'
    ok['split_step_allreduces'] = split.n_allreduce >= st4['gramian_steps']      # one per product (+ the plan's entry count)
    ok['split_step_same_bits'] = bool(torch.equal(sl, s4) and torch.equal(Vl, V4)) and stl['gramian_steps'] == st4['gramian_steps']
    split2 = TorchComm(split_step=True)
    _, s5, V5, st5 = svd_topk(ops, A, 10, seed=3, comm=split2, method='lanczos', exchange='relaxed')
    ok['split_step_relaxed_exchange'] = bool(torch.allclose(sl, s5, rtol=1e-12) and float((Vl @ Vl.T - V5 @ V5.T).abs().max()) < 1e-10
                                             and st5['verified_rel_residual'] <= 1e-12)
    print('RCCL_ONE_RANK_RESULT', ok, 'split-step', split.n_allreduce, st4['gramian_steps'], 'relaxed from', st5.get('exchange_relaxed_from'), 'allgathers', comm.n_allgather, 'reduce_scatters', comm.n_reduce_scatter, 'allreduces', comm.n_allreduce)
    assert all(ok.values()), ok


if __name__ == '__main__':
    main()
