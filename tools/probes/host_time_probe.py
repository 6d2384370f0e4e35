"""(round 2) host time of one scoring pass launched from Python: wall time of enqueueing passes back to back (no sync
inside), against the GPU time of the same passes, and a cProfile of the enqueue path.
usage: python tools/probes/host_time_probe.py [ml20m|s1m] [rank] [topk]"""
import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
topk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, rank)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(V.shape[0], device=order2.device)
V = V[order2].contiguous()
A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
for _ in range(5):
    scoring.recommend(ops, F, A, topk, True)
torch.cuda.synchronize()
N = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(N):
    scoring.recommend(ops, F, A, topk, True)
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('enqueue (host) %.3f ms/pass   GPU span %.3f ms/pass   wall %.3f ms/pass' % (1e3 * t_host / N, e0.elapsed_time(e1) / N, 1e3 * t_all / N))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    scoring.recommend(ops, F, A, topk, True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
