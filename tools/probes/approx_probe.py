import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 's1m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(c['indices'], c['shape'][1])
A = ops.csr_relabel_cols(A, rank_of); _ = A.T
_, s, V, st = svd_topk(ops, A, cfg['rank'])
n_items = A.shape[1]
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(n_items, device=order2.device)
V = V[order2].contiguous()
A = ops.csr_relabel_cols(A, rank2, sort=False)
F = scoring.FactorImage(ops, V)
for approx in (False, True):
    stt = {}
    scoring.recommend(ops, F, A, cfg['topk'], True, stats=stt, approx_fold_in=approx)
    for _ in range(2): scoring.recommend(ops, F, A, cfg['topk'], True, approx_fold_in=approx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): scoring.recommend(ops, F, A, cfg['topk'], True, approx_fold_in=approx)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    ops.timers = {}
    scoring.recommend(ops, F, A, cfg['topk'], True, approx_fold_in=approx)
    torch.cuda.synchronize()
    sp = [a.elapsed_time(b) for a, b, _ in ops.timers['spmm']]
    ops.timers = None
    print('approx', approx, 'ms/pass %.3f' % ms, 'refolded', stt.get('refolded_users'), 'flagged', stt['flagged_users'], 'spmm launches ms', ['%.3f' % x for x in sp])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    r = scoring.recommend(ops, F, A, cfg['topk'], True)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
