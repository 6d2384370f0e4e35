import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
from polara_amd import scoring
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(c['indices'], c['shape'][1])
A = ops.csr_relabel_cols(A, rank_of); _ = A.T
_, s, V, st = svd_topk(ops, A, cfg['rank'])
F = scoring.FactorImage(ops, V)
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = scoring.recommend(ops, F, A, cfg['topk'], True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('step ms:', ' '.join('%.2f' % t for t in ts))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    r = scoring.recommend(ops, F, A, cfg['topk'], True)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
