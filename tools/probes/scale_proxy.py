import sys, time, json, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
from polara_amd.csr import popularity_order, nnz_balanced_row_partition
from polara_amd import scoring
ops = HipOps('cuda:0')
WL = sys.argv[1] if len(sys.argv) > 1 else 's1m'
csr, cfg = make_workload(WL, device='cuda:0')
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
out = {}
for label, override in (('auto_splits', 0), ('single_sweep', 1)):   # item splits for small user sets vs one sweep per group
    ops.score_splits_override = override
    res, base = {}, None
    for N in (1, 2, 4, 8):
        bounds = nnz_balanced_row_partition(c['indptr'], N)
        T = A if N == 1 else ops.csr_rows(A, 0, int(bounds[1]))
        st = {}
        scoring.recommend(ops, F, T, cfg['topk'], True, stats=st)
        for _ in range(3): scoring.recommend(ops, F, T, cfg['topk'], True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): scoring.recommend(ops, F, T, cfg['topk'], True)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        if base is None: base = ms
        res['N=%d' % N] = {'users_on_rank0': T.shape[0], 'ms_per_pass': ms, 'speedup_vs_N1': base / ms,
                           'item_splits': st['item_splits'], 'swept_fraction': st['tiles_scored'] / max(st['tiles_total'], 1)}
    out[label] = res
ops.score_splits_override = 0
print(json.dumps({'strong_scaling_proxy_scoring_' + WL: out, 'note': 'one GPU scoring the users rank 0 would own at N GPUs (nnz-balanced contiguous shard); no collective in the scoring pass'}))
