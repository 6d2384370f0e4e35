"""Cycle breakdown inside the candidate sweep (library built with PK_SCORE_PROFILE=1 PK_FAST_BUILD=1): wave-cycles of the
whole kernel, of the flush sorts, the seen-tile walk, the push path, prologue; flush count and tiles."""
import os, sys, json, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
from polara_amd.solver import svd_topk
from polara_amd import scoring
ops = HipOps('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
csr, cfg = make_workload(name, device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, _ = popularity_order(None, n_items, counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
_, s, V, st = svd_topk(ops, A, 50)
order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(n_items, device=order2.device)
V = V[order2].contiguous(); A = ops.csr_relabel_cols(A, rank2)
F = scoring.FactorImage(ops, V)
buf = (ctypes.c_ulonglong * 8)()
ops.lib.pk_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ('kernel', 'flush', 'walk', 'push_incl_flush', 'prologue', 'bootstrap', 'n_flush', 'tiles')
configs = [dict(boot=0, head=0), dict(boot=16, head=0), dict(boot=32, head=0), dict(boot=0, head=32), dict(boot=16, head=32)]
if len(sys.argv) > 2:
    configs = [dict(zip(('boot', 'head'), (int(x) for x in a.split(',')))) for a in sys.argv[2:]]
for cfg in configs:
    os.environ['PK_SCORE_BOOT_TILES'] = str(cfg['boot'])
    os.environ['PK_SCORE_HEAD_TILES'] = str(cfg['head'])
    for _ in range(3): scoring.recommend(ops, F, A, 10, True)
    torch.cuda.synchronize()
    ops.lib.pk_debug_profile(None, 1)
    n = 5
    for _ in range(n): scoring.recommend(ops, F, A, 10, True)
    torch.cuda.synchronize()
    ops.lib.pk_debug_profile(buf, 0)
    d = {k: int(v) / n for k, v in zip(names, buf)}
    d['per_tile_wave_cycles'] = d['kernel'] / max(d['tiles'], 1)
    for k in ('flush', 'walk', 'push_incl_flush', 'prologue', 'bootstrap'):
        d[k + '_share'] = round(d[k] / d['kernel'], 4)
    d['cycles_per_flush'] = d['flush'] / max(d['n_flush'], 1)
    ops.timers = {}
    for _ in range(5): scoring.recommend(ops, F, A, 10, True, batches=1)
    torch.cuda.synchronize()
    d['sweep_ms'] = float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in ops.timers['score_candidates']]))
    ops.timers = None
    print(json.dumps(dict(cfg, **d)), flush=True)
