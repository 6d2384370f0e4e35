"""The two-panel exchange of ItemRows.product on ONE rank with its collectives exercised (RCCL, world 1): what the split costs
on the compute side and what the asynchronous calls cost when there is nobody to exchange with.
usage: python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/probes/overlap_one_rank.py [ml20m|s1m]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from polara_amd.dist import init_from_env, TorchComm
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
init_from_env()
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:%d' % torch.cuda.current_device())
csr, cfg = make_workload(wl, device=str(ops.device))
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv, counts, rank_dev = ops.item_order(A)
A = ops.csr_relabel_cols(A, rank_dev); A.transpose_operator(); _ = A.plan
comm = TorchComm(exercise_collectives=True)
def run(tag, **kw):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        _, s, V, st = svd_topk(ops, A, 50, method='lanczos', **kw)
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t) * 1e3, 2))
    print(json.dumps(dict(case=tag, solve_ms=ts, steps=st['gramian_steps'], panels=comm.n_panel_exchanges)), flush=True)
run('no communicator')
for mode in ('0', 'force'):
    os.environ['PK_DIST_OVERLAP'] = mode
    for shard in (True, False):
        run('RCCL one rank, overlap=%s, items %s' % (mode, 'sharded' if shard else 'replicated'), comm=comm, shard_items=shard)
