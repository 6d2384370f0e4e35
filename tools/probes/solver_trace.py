import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order
wl = sys.argv[1] if len(sys.argv) > 1 else 's1m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(c['indices'], c['shape'][1])
A = ops.csr_relabel_cols(A, rank_of); _ = A.T
svd_topk(ops, A, cfg['rank'])
k = cfg['rank']
import polara_amd.solver as sv
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, s, V, st = svd_topk(ops, A, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('cholqr: build %.3f steps %d degrees %s conv %s res %s' % (dt, st['gramian_steps'], st['degrees'], st['converged'], st['final_rel_residual']))
orig = sv.orthonormalize
sv.orthonormalize = lambda ops_, X, V_lock=None: sv._whiten(ops_, X, V_lock)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, s2, V2, st = svd_topk(ops, A, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('whiten: build %.3f steps %d degrees %s conv %s res %s  dsigma %.2e' % (dt, st['gramian_steps'], st['degrees'], st['converged'], st['final_rel_residual'], float(((s - s2).abs() / s2).max())))
