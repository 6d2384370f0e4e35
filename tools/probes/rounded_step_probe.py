"""One step of the library recurrence (pk_lanczos_steps) with exact and with rounded products, timed with HIP events at a
few basis sizes: what the fp32 images of the dense blocks save per step (round 6).
    python tools/probes/rounded_step_probe.py [ml20m|s1m] [b]"""
import sys, json
import numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
from polara_amd.solver import orthonormalize

wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
n = A.shape[1]
rec = ops.lanczos_recurrence(A, b)
cap = 26
Q = ops.empty(n, cap * b); T = ops.zeros(cap * b, cap * b); S = ops.zeros(b, b); flags = ops.zeros(2)
Q[:, :b] = orthonormalize(ops, ops.randn(n, b, 0))
for j in range(24):
    rec.steps(Q, T, S, flags, j, 1, False)
torch.cuda.synchronize()
out = {}
for rounded in (False, True):
    for j0 in (4, 12, 20):
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rec.steps(Q, T, S, flags, j0, 1, False, rounded=rounded)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out['%s_j%d' % ('rounded' if rounded else 'exact', j0)] = round(float(np.median(ts[1:])), 4)
ops.timers = {}
rec.steps(Q, T, S, flags, 12, 1, False, rounded=False); torch.cuda.synchronize(); rec.collect_timings()
out['spmm_exact_ms'] = [round(e.elapsed_time(None), 4) for e, _, _ in ops.timers['spmm']]
ops.timers = {}
rec.steps(Q, T, S, flags, 12, 1, False, rounded=True); torch.cuda.synchronize(); rec.collect_timings()
out['spmm_rounded_ms'] = [round(e.elapsed_time(None), 4) for e, _, _ in ops.timers['spmm']]
print(json.dumps(dict(workload=wl, b=b, **out)))
