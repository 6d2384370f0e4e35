"""(round 2) time of the small dense kernels of the eigensolver at the shapes of the headline build: chol_rinv and eigh_psd at
l = 64 / 128 / 256, Gram / tall-skinny GEMM / recurrence / residual at n_items x l.
usage: python tools/probes/dense_small_probe.py [n_items]"""
import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from polara_amd.ops import HipOps
ops = HipOps('cuda:0')
n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 26744


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps   # us


out = {'n_items': n_items}
for l in (64, 128, 200, 256):
    X = torch.randn(n_items, l, dtype=torch.float64, device='cuda:0')
    Y = torch.randn(n_items, l, dtype=torch.float64, device='cuda:0')
    G = ops.gram(X)
    Xg = X * torch.exp(-torch.arange(l, dtype=torch.float64, device='cuda:0') / (l / 12.0))     # graded columns: e^-12 over the block
    Q, _ = torch.linalg.qr(torch.randn(l, l, dtype=torch.float64, device='cuda:0'))
    Gg = ops.gram(Xg @ Q)                                                                           # ... in a rotated basis
    C = torch.randn(l, l, dtype=torch.float64, device='cuda:0')
    th = torch.rand(l, dtype=torch.float64, device='cuda:0')
    out['l=%d' % l] = dict(
        chol_rinv_us=timed(lambda: ops.chol_rinv(G)), eigh_psd_us=timed(lambda: ops.eigh_psd(G), reps=5),
        eigh_psd_graded_us=timed(lambda: ops.eigh_psd(Gg), reps=5),
        gram_us=timed(lambda: ops.gram(X)), gram2_us=timed(lambda: ops.gram(X, Y)), tsmm_us=timed(lambda: ops.tsmm(X, C)),
        axpbypcz_us=timed(lambda: ops.axpbypcz(1.0, X, 2.0, Y, 3.0, X)), resid_us=timed(lambda: ops.resid_colnorm2(X, Y, th)))
print(json.dumps(out))
