import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps, DeviceCSR
ops = HipOps('cuda:0')
dev = 'cuda:0'
n_cols, nc, nnz = 100_000, 50, 100_000_000
g = torch.Generator(device=dev); g.manual_seed(0)
# Zipf-like column popularity (as the workload): rank^-0.8
w = torch.arange(1, n_cols + 1, device=dev, dtype=torch.float64).pow(-0.8)
cols_all = torch.multinomial(w, 10_000_000, replacement=True, generator=g).to(torch.int32)
cols_all = cols_all.repeat(nnz // cols_all.numel())
vals = torch.ones(nnz, dtype=torch.float32, device=dev)
X = torch.randn(n_cols, nc, generator=g, dtype=torch.float64, device=dev)
for row_len in (25, 50, 100, 200, 400, 1000):
    n_rows = nnz // row_len
    indptr = torch.arange(0, nnz + 1, row_len, dtype=torch.int64, device=dev)
    A = DeviceCSR.from_device(ops, indptr, cols_all, vals, (n_rows, n_cols))
    out = ops.empty(n_rows, nc)
    for _ in range(2): ops.spmm(A, X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.spmm(A, X, out=out)
    torch.cuda.synchronize(); print('row_len %d: %.3f ms (%d rows)' % (row_len, (time.perf_counter() - t0) / 5 * 1e3, n_rows))
    del A, out
