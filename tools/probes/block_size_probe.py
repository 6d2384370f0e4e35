"""rows_per_block of the user-blocked transposed product (ops.BlockedTranspose): time per Z = A^T Y, nc = 64 / 128"""
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
ops = HipOps('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
csr, cfg = make_workload(name, device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
A0 = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, _ = popularity_order(None, n_items, counts=ops.item_counts(A0))
out = {'workload': name}
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for sort in (True, False):
    A = ops.csr_relabel_cols(A0, rank_of, sort=sort)
    tag = 'sorted' if sort else 'unsorted'
    for nc in (64, 128):
        X = ops.randn(n_items, nc, 1); Y = ops.randn(n_users, nc, 2)
        out['%s_AX_nc%d' % (tag, nc)] = timeit(lambda: ops.spmm(A, X))
        if sort:
            out['AtY_plain_nc%d' % nc] = timeit(lambda: ops.spmm(A.T, Y))
            for rpb in (4096, 8192, 16384, 32768, 65536, 131072):
                if rpb * 2 > n_users and rpb > 16384: continue
                Tb = A.T_blocked(rpb)
                out['AtY_rpb%d_nc%d' % (rpb, nc)] = timeit(lambda: ops.spmm(Tb, Y))
            A._Tb = None
    # the fold-in of the scoring pass against an fp32 block
    X32 = torch.randn(n_items, 64, device='cuda:0', dtype=torch.float32)
    out['%s_fold_f32_nc52' % tag] = timeit(lambda: ops.spmm(A, X32[:, :52]))
print(json.dumps(out))
