"""Y = A X with the ITEMS cut into panels (the gathered rows of X of one launch then span `panel` items instead of the
whole catalogue: 8 192 rows of 64 fp64 columns = 4 MiB = one XCD's L2), against the single launch.  The panel image is the
machinery of the user-blocked transposed product (ops.BlockedTranspose) applied to A^T: csr_transpose(A^T, rows_per_block)
orders the entries by (item panel, user, item).  usage: python tools/probes/item_panel_probe.py [ml20m|s1m]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps, BlockedTranspose
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order

ops = HipOps('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
csr, cfg = make_workload(name, device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, _ = popularity_order(None, n_items, counts=ops.item_counts(A))
A = ops.csr_relabel_cols(A, rank_of)
X = ops.randn(n_items, 64, 1)
Y0 = ops.spmm(A, X)


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'workload': name, 'single_launch_ms': timed(lambda: ops.spmm(A, X))}
for panel in (4096, 8192, 16384, 32768):
    if panel >= n_items:
        continue
    P = BlockedTranspose(ops, A.T, rows_per_block=panel)
    Y = P.apply(X)
    err = float((Y - Y0).abs().max() / Y0.abs().max())
    out['panel_%d' % panel] = dict(ms=timed(lambda: P.apply(X)), blocks=P.n_blocks, tasks=int(P.image.n_tasks), rel_err=err)
    del P
    torch.cuda.empty_cache()
print(json.dumps(out))
