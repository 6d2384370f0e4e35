"""(round 2) per-kernel times (HIP events) of the scoring pass on the user shard rank 0 owns at N = 1, 2, 4, 8 GPUs, and the
pass time with the passes queued back to back (no events).   usage: python tools/probes/shard_pass_breakdown.py [ml20m|s1m]"""
import sys, os, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order, nnz_balanced_row_partition
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
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
out = {}
for N in (1, 2, 4, 8):
    bounds = nnz_balanced_row_partition(c['indptr'], N)
    T = A if N == 1 else ops.csr_rows(A, 0, int(bounds[1]))
    for _ in range(5):
        scoring.recommend(ops, F, T, 10, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        scoring.recommend(ops, F, T, 10, True)
    torch.cuda.synchronize()
    ms_pass = 1e3 * (time.perf_counter() - t0) / 50
    ops.timers = {}
    for _ in range(10):
        scoring.recommend(ops, F, T, 10, True)
    torch.cuda.synchronize()
    ms = {k: round(float(np.median([a.elapsed_time(b) for a, b, _ in v])), 4) for k, v in ops.timers.items()}
    ops.timers = None
    stt = {}
    scoring.recommend(ops, F, T, 10, True, stats=stt)
    out['N=%d' % N] = dict(users=T.shape[0], ms_per_pass=round(ms_pass, 4), kernels_ms=ms, sum_kernels_ms=round(sum(ms.values()), 4),
                           item_splits=stt['item_splits'], swept=round(stt['tiles_scored'] / max(stt['tiles_total'], 1), 4),
                           exit=stt.get('exit_tile_quantiles'))
print(json.dumps(out))
