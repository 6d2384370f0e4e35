"""Upper bound of what an LDS-resident head of the factor image could save the fold-in: time of E = T V32 with all
entries, with only the entries of items >= H (the part that would still gather from L2), and with only items < H."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
sys.argv = sys.argv[:1]
bench = B.Bench(B.parse())
ops = bench.ops
c = bench.generate('ml20m')
st, _ = bench.build(c, 50, True)
F, A = st['F'], st['A']
T, perm = A.by_activity()
ip, ix, vv = ops.to_host(T.indptr), ops.to_host(T.indices), ops.to_host(T.values)
rows = np.repeat(np.arange(T.shape[0]), np.diff(ip))


def sub(mask):
    cnt = np.bincount(rows[mask], minlength=T.shape[0])
    return ops.csr(np.r_[0, np.cumsum(cnt)].astype(np.int64), ix[mask], vv[mask], T.shape)


def timed(M, n=20):
    Ex = ops.empty(M.shape[0], F.Kx)
    for _ in range(3): ops.spmm(M, F.V32x, out=Ex)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ops.spmm(M, F.V32x, out=Ex)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'all_ms': timed(T), 'nnz': int(len(ix))}
for H in (738, 2048, 4096):
    head = ix < H
    out['H=%d' % H] = dict(head_share=float(head.mean()), tail_only_ms=timed(sub(~head)), head_only_ms=timed(sub(head)))
print(json.dumps(out))
