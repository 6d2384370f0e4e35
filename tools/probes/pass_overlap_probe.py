"""(round 2) why is the fold-in SpMM 60 % slower inside bench.py's pipelined loop (466 vs 290 us, rocprofv3 per-call trace)?
Runs the full scoring pass back to back in variants of the loop and reports the fold-in / sweep time (HIP events).
usage: python tools/probes/pass_overlap_probe.py"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m', device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, 50)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(V.shape[0], device=order2.device)
V = V[order2].contiguous()
A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
n_users = A.shape[0]
host = [torch.empty((n_users, 10), dtype=torch.int64).pin_memory() for _ in range(2)]
static = torch.zeros((n_users, 10), dtype=torch.int64, device='cuda:0')
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
done = [torch.cuda.Event() for _ in range(2)]


def run(mode, n=24):
    for _ in range(3):
        scoring.recommend(ops, F, A, 10, True)
    torch.cuda.synchronize()
    ops.timers = {}
    t0 = time.perf_counter()
    for i in range(n):
        if i >= 2 and 'copy' in mode:
            done[i & 1].synchronize()
        recs = scoring.recommend(ops, F, A, 10, True)
        if mode == 'pass only':
            continue
        if mode == 'sync each':
            torch.cuda.synchronize(); continue
        ready = torch.cuda.Event(); ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            src = static if mode == 'copy static' else recs
            if mode != 'event only':
                host[i & 1].copy_(src, non_blocking=True)
                if mode != 'copy no record_stream':
                    src.record_stream(side)
            done[i & 1].record(side)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / n
    ms = {k: float(np.median([a.elapsed_time(b) for a, b, _ in v][3:])) for k, v in ops.timers.items()}
    ops.timers = None
    return wall, ms


for mode in ('pass only', 'sync each', 'event only', 'copy static', 'copy', 'copy no record_stream'):
    wall, ms = run(mode)
    print('%-24s %.3f ms/pass   fold-in %.3f  sweep %.3f  rescore %.3f' % (mode, wall, ms['spmm'], ms['score_candidates'], ms['rescore_topk']))
