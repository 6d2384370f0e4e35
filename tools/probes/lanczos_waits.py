"""Where the calling thread of a block Lanczos build WAITS: host time inside the collections of the monitors (blocked
until the side stream's nested solve is done) and inside the looks on the main stream (final look, verification excluded).
usage: python tools/probes/lanczos_waits.py [ml20m|s1m] [rank]"""
import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd import solver
from polara_amd.csr import popularity_order
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of); A.transpose_operator(); _ = A.plan
acc = dict(join=[], look=[])
j0, r0 = solver._Monitor.join, solver._ritz_check
main = __import__('threading').main_thread()
def join(self):
    t = time.perf_counter(); out = j0(self); acc['join'].append(round((time.perf_counter() - t) * 1e3, 3)); return out
def look(*a, **k):
    if __import__('threading').current_thread() is not main:
        return r0(*a, **k)
    t = time.perf_counter(); out = r0(*a, **k); acc['look'].append(round((time.perf_counter() - t) * 1e3, 3)); return out
solver._Monitor.join = join
solver._ritz_check = look
for rep in range(4):
    acc['join'].clear(); acc['look'].clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    _, s, V, st = solver.svd_topk(ops, A, rank, method='lanczos')
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
    print(json.dumps(dict(solve_ms=round(dt, 2), steps=st['gramian_steps'], join_ms=list(acc['join']), main_look_ms=list(acc['look']))), flush=True)
