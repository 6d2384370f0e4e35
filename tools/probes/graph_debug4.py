import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import planted_csr, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd import scoring
ops = HipOps('cuda:0')
stage = int(sys.argv[1])
n_users, n_items, rank, topk = 40000, 3000, 12, 10
c = csr_to_numpy(planted_csr(n_users, n_items, 40, rank, seed=77, min_items=5, max_items=300))
T = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
_, _, V, st = svd_topk(ops, T, rank)
F = scoring.FactorImage(ops, V)
T.nonneg(); T.seen_tiles(); _ = T.plan
K = F.K; KC = ops.candidate_capacity(topk)
def run():
    Ex = ops.empty(n_users, F.Kx); E = Ex[:, :K]
    ops.spmm(T, F.V32x, out=Ex, rows=(0, n_users))
    if stage < 1: return Ex
    w = Ex[:, K]
    Ep, ub = ops.pack_frag_bound(E, extra=w, extra_scale=1.2e-7)
    if stage < 2: return Ep
    st_ = T.seen_tiles()
    cs, ci = ops.score_candidates(F.Vp, Ep, n_users, n_items, K, T.indptr, T.indices, KC, 1, user_bound=ub, tile_bound=F.tile_bound, seen_tiles=st_)
    if stage < 3: return cs
    out_idx = torch.empty(n_users, topk, dtype=torch.int64, device=E.device); out_s = torch.empty(n_users, topk, dtype=torch.float64, device=E.device)
    flags = torch.empty(n_users, dtype=torch.int32, device=E.device)
    outs = (out_idx, out_s, flags)
    ops.rescore_topk(F.V, E, n_items, T.indptr, KC, cs, ci, topk, F.vmax, want_scores=True, splits=1, out=outs, e_err=w, v32=F.V32x)
    if stage < 4: return out_idx
    lst, cnt = ops.flag_compact(flags, 7)
    if stage < 5: return out_idx
    ops.fold_rows(T, lst, cnt, F.V, Ex, row_offset=0)
    if stage < 6: return out_idx
    ops.rescore_topk(F.V, E, n_items, T.indptr, KC, cs, ci, topk, F.vmax, want_scores=True, splits=1, out=outs, rows=lst, n_rows_dev=cnt, e_err=w, e_exact=True)
    if stage < 7: return out_idx
    lst2, cnt2 = ops.flag_compact(flags, 0x7fffffff)
    ops.score_exact_list(lst2, cnt2, F.V, E, n_items, T.indptr, T.indices, topk, out_idx, out_s)
    return out_idx
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run(); run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
buf = torch.zeros(n_users, 10, device='cuda')
for i in range(3):
    g.replay(); torch.cuda.synchronize(); z = buf + 1; torch.cuda.synchronize()
print('stage', stage, 'ok', flush=True)
