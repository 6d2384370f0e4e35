"""Where the time of bench.py's `reindex_and_images_s` goes on the FIRST build of a size and on the second (VERDICT r5 weak #7:
S-1M 0.124 s first, 0.004 s second): every stage bracketed by a device synchronisation, with the caching allocator's
device allocations (hipMalloc calls and bytes) counted per stage.
    python tools/probes/reindex_stages.py [s1m|ml20m]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
from polara_amd import scoring
from polara_amd.synth import make_workload, csr_to_numpy

wl = sys.argv[1] if len(sys.argv) > 1 else 's1m'
ops = HipOps('cuda:0')
csr, cfg = make_workload(wl, device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
rank = 50


def stats():
    s = torch.cuda.memory_stats()
    return s.get('num_device_alloc', 0), s.get('reserved_bytes.all.current', 0)


for rep in range(3):
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    V = ops.randn(n_items, rank, 7 + rep)
    V = V * torch.linspace(3.0, 0.2, n_items, device=V.device, dtype=V.dtype).unsqueeze(1)
    torch.cuda.synchronize()
    rec = {}
    marks = [time.perf_counter()]
    a0 = stats()

    def lap(name):
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        a1 = stats()
        rec[name] = dict(ms=round(1e3 * (marks[-1] - marks[-2]), 3), mallocs=a1[0] - lap.a[0], reserved_MB=round((a1[1] - lap.a[1]) / 2**20, 1))
        lap.a = a1
    lap.a = a0
    order2, rank2, V2 = ops.norm_order(V)
    lap('norm_order_and_gather')
    A_score = ops.csr_relabel_cols(A, rank2, sort=True)
    lap('relabel_cols')
    F = scoring.FactorImage(ops, V2)
    lap('factor_image')
    Ta = A_score.by_activity()[0] if A_score.shape[0] >= scoring.ORDER_USERS_MIN else A_score
    lap('by_activity')
    Ta.seen_tiles()
    lap('seen_tiles')
    rec['total_ms'] = round(1e3 * (marks[-1] - marks[0]), 3)
    print(json.dumps(dict(workload=wl, rep=rep, **rec)), flush=True)
    del A, A_score, F, Ta, V, V2
