"""Time of the ingest stages of a build, one by one (device synchronised around each): upload, item order, renaming, CSC
image, the two row plans, serving-side images.  usage: python tools/probes/ingest_stages.py [ml20m|s1m]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from polara_amd.ops import HipOps, BlockedTranspose
from polara_amd.synth import make_workload, csr_to_numpy
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
def lap(f, reps=5):
    out = None; ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return out, round(min(ts), 3)
res = {}
A, res['upload'] = lap(lambda: ops.csr(c['indptr'], c['indices'], c['values'], c['shape']))
(rank, inv, counts, rank_dev), res['item_order'] = lap(lambda: ops.item_order(A))
B, res['relabel_sorted'] = lap(lambda: ops.csr_relabel_cols(A, rank_dev))
_, res['relabel_unsorted'] = lap(lambda: ops.csr_relabel_cols(A, rank_dev, sort=False))
def tr():
    B._Tb = None; B._T = None
    return B.transpose_operator()
T, res['transpose_operator'] = lap(tr)
_, res['csr_transpose_kernel_only'] = lap(lambda: ops.csr_transpose(B, rows_per_block=T.rows_per_block if hasattr(T, 'rows_per_block') else 0))
def plan():
    B._plan = None
    return B.plan
_, res['row_plan_A'] = lap(plan)
P, res['by_activity'] = lap(lambda: (setattr(B, '_by_activity', None), B.by_activity())[1])
def st():
    P[0]._seen_tiles = None
    return P[0].seen_tiles()
_, res['seen_tiles'] = lap(st)
print(json.dumps(res))
