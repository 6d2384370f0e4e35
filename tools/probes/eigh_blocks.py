"""Block Jacobi (n > 136) against the number of row blocks: time per solve and sweeps on a graded Gram matrix of the size
of the (30,30,5) HOOI unfoldings (150) and of a rank-200 solver block (256).  PK_EIGH_NB is read per call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps
ops = HipOps('cuda:0')
rs = np.random.RandomState(0)
for n in (150, 256):
    M = rs.randn(4000, n) * np.exp(-np.arange(n) / 12.0)[None, :]
    M = M @ np.linalg.qr(rs.randn(n, n))[0]
    S = ops.to_device(M.T @ M)
    ref = np.linalg.eigvalsh(M.T @ M)[::-1]
    for nb in (4, 6, 8, 10, 12, 16, 20):
        os.environ['PK_EIGH_NB'] = str(nb)
        for _ in range(2):
            lam, C = ops.eigh_psd(S.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lam, C = ops.eigh_psd(S.clone())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        err = float(np.abs(ops.to_host(lam) - ref).max() / ref[0])
        Ch = ops.to_host(C)
        orth = float(np.abs(Ch.T @ Ch - np.eye(n)).max())
        print(json.dumps(dict(n=n, nb=nb, ms=round(dt * 1e3, 3), eval_err=err, orth=orth)))
