import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from polara_amd.ops import HipOps
ops = HipOps('cuda:0')
rng = np.random.RandomState(0)
for n in (64, 128, 136, 150, 200, 256, 384):
    X = rng.randn(4 * n, n) * np.exp(rng.randn(n))
    G = ops.to_device(X.T @ X)
    for _ in range(2): ops.eigh_psd(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): lam, C = ops.eigh_psd(G)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    info = ops.to_host(ops._info)
    print('n %d: %.2f ms  sweeps %d converged %d' % (n, dt, info[0], info[1]))
