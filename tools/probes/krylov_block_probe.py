"""Width of a Krylov block of the block Lanczos build against the build time (round 6): narrower blocks take more Gramian steps
but every step gathers fewer columns.  Prints, per width, wall time of svd_topk (warm), steps, SpMM time, residual.
    python tools/probes/krylov_block_probe.py [ml20m|s1m] [rank] [widths...]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order

wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
# widths: 0 = the solver's own choice; 'b:first:lag' also pins the step of the first look and the monitors' lag
specs = [a.split(':') for a in sys.argv[3:]] or [['64'], ['48'], ['32'], ['24'], ['16']]
widths = [(int(sp[0]) or None, (int(sp[1]) or None) if len(sp) > 1 else None, (int(sp[2]) if sp[2] != '' else None) if len(sp) > 2 else None,
           'relaxed' if len(sp) > 3 and sp[3] == 'r' else 'f64') for sp in specs]      # 'b:first:lag:r' — r: rounded late products
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
counts = ops.item_counts(A)
rank_of, inv = popularity_order(None, c['shape'][1], counts=counts)
A = ops.csr_relabel_cols(A, rank_of)
A.transpose_operator(); _ = A.plan
s_ref = None
for kb, first_look, lag, products in widths:
    ts = []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, s, V, st = svd_topk(ops, A, rank, method='lanczos', krylov_block=kb, first_look=first_look, monitor_lag=lag, products=products)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ops.timers = {}
    svd_topk(ops, A, rank, method='lanczos', krylov_block=kb, first_look=first_look, monitor_lag=lag, products=products)
    torch.cuda.synchronize()
    spmm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ops.timers.get('spmm', []))
    ops.timers = None
    if s_ref is None:
        s_ref, V_ref = s.clone(), V.clone()
    print(json.dumps(dict(workload=wl, rank=rank, krylov_block=st.get('krylov_block'), lag=st.get('monitor_lag'), rounded_from=st.get('products_rounded_from'), verified=st.get('verified_rel_residual'), monitor_wait_ms=st.get('monitor_wait_ms'), look_ms=st.get('look_ms'), solve_ms=[round(1e3 * t, 2) for t in ts], steps=st.get('lanczos_steps'),
                          gramian_steps=st['gramian_steps'], spmm_ms=round(spmm_ms, 2), residual=st['final_rel_residual'],
                          method=st['method'], nested=st.get('nested'), checks=[(a, float('%.2e' % b)) for a, b in st.get('checks', [])],
                          fallback=st.get('lanczos_fallback'),
                          sigma_rel_diff=float(((s - s_ref).abs() / s_ref).max()),
                          projector_diff=float((V @ V.T[:, :200] - V_ref @ V_ref.T[:, :200]).abs().max()))), flush=True)
