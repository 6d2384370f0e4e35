"""pk_chol_rinv_f64 at the sizes a build uses (16 x 16 in every step's CholeskyQR3, 64 x 64 in the nested solves, 128 x 128 at
rank 100): HIP-event time per call over 200 back-to-back calls.
    python tools/probes/chol_probe.py"""
import sys, json
import numpy as np, torch
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
ops = HipOps('cuda:0')
out = {}
for n in (8, 16, 24, 32, 48, 64, 96, 128):
    rng = np.random.RandomState(n)
    X = rng.randn(4 * n + 3, n) * np.exp(rng.randn(n))
    G = ops.to_device(X.T @ X)
    info = torch.zeros(1, dtype=torch.int32, device=G.device)
    for _ in range(20):
        ops.chol_rinv(G, 1e-12, info=info)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.chol_rinv(G, 1e-12, info=info)
    e1.record(); torch.cuda.synchronize()
    out[n] = round(1e3 * e0.elapsed_time(e1) / 200, 2)
print(json.dumps({'chol_rinv_us_per_call': out}))
