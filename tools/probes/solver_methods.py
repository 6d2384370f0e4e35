"""Block Lanczos against the filtered subspace iteration on one GPU: wall time of svd_topk (warm, un-instrumented), Gramian
steps, and where the time goes (every ops call of the solver bracketed by a device sync — inflates the small launches by
the sync, which is why the un-instrumented total is printed first).
    python tools/probes/solver_methods.py [ml20m|s1m|ml1m] [rank]"""
import sys, time, json, collections
import numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order

wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
rank = int(sys.argv[2]) if len(sys.argv) > 2 else {'ml20m': 50, 's1m': 50, 'ml1m': 10}[wl]
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
counts = ops.item_counts(A)
rank_of, inv = popularity_order(None, c['shape'][1], counts=counts)
A = ops.csr_relabel_cols(A, rank_of)
A.transpose_operator(); _ = A.plan


class Prof:
    """ops proxy: every call synchronised and timed"""
    def __init__(self, ops):
        object.__setattr__(self, '_o', ops)
        object.__setattr__(self, 't', collections.defaultdict(float))
        object.__setattr__(self, 'n', collections.defaultdict(int))

    def __getattr__(self, name):
        a = getattr(self._o, name)
        if not callable(a) or name in ('stream',):
            return a

        def f(*args, **kw):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = a(*args, **kw)
            torch.cuda.synchronize()
            self.t[name] += time.perf_counter() - t0; self.n[name] += 1
            return r
        return f

    def __setattr__(self, k, v):
        setattr(self._o, k, v)


out = {}
for meth in ('subspace', 'lanczos'):
    ts = []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, s, V, st = svd_topk(ops, A, rank, method=meth)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ops.timers = {}
    svd_topk(ops, A, rank, method=meth)
    torch.cuda.synchronize()
    spmm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ops.timers.get('spmm', []))
    ops.timers = None
    p = Prof(ops)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    svd_topk(p, A, rank, method=meth)
    torch.cuda.synchronize(); t_prof = time.perf_counter() - t0
    out[meth] = dict(solve_ms=[round(1e3 * t, 2) for t in ts], gramian_steps=st['gramian_steps'], spmm_ms=round(spmm_ms, 2),
                     residual=st['final_rel_residual'], method=st['method'], nested=st.get('nested'), checks=st.get('checks'),
                     fallback=st.get('lanczos_fallback'), sigma_first_last=[float(s[0]), float(s[-1])],
                     instrumented_total_ms=round(1e3 * t_prof, 2),
                     per_call_ms={k: [round(1e3 * v, 2), p.n[k]] for k, v in sorted(p.t.items(), key=lambda kv: -kv[1])})
    print(meth, json.dumps(out[meth]), flush=True)
    if meth == 'subspace':
        s_ref, V_ref = s.clone(), V.clone()
print('sigma max rel diff', float(((s - s_ref).abs() / s_ref).max()), 'projector diff', float((V @ V.T[:, :200] - V_ref @ V_ref.T[:, :200]).abs().max()))
