"""pk_eigh_top_f64 against numpy.linalg.eigh on the shapes HOOI produces and on awkward spectra: accuracy, the kernel's
own verdict, time per solve next to the Jacobi kernel."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd import _lib
from polara_amd.ops import _ptr
ops = HipOps('cuda:0')
rs = np.random.RandomState(0)

def run(name, S, r, reps=10):
    n = S.shape[0]
    Sd = ops.to_device(S)
    R = ops.empty(r, n); lam = ops.empty(r)
    info = torch.zeros(16, dtype=torch.int32, device=ops.device)
    work = ops._work(ops.lib.pk_eigh_top_work_bytes(n))
    def call():
        _lib.check(ops.lib.pk_eigh_top_f64(ops.stream(), n, _ptr(Sd), n, r, _ptr(R), n, _ptr(lam), _ptr(work), _ptr(info)), 'top')
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    ok = int(info[0].item())
    phases = info[2:8].tolist() + ['P1:'] + info[8:12].tolist()
    t0 = time.perf_counter()
    for _ in range(reps): ops.eigh_psd(Sd)
    torch.cuda.synchronize(); dj = (time.perf_counter() - t0) / reps
    w, V = np.linalg.eigh(S); w = w[::-1]; V = V[:, ::-1]
    out = dict(case=name, n=n, r=r, ok=ok, phase_clocks_x100=phases, top_ms=round(dt * 1e3, 3), jacobi_ms=round(dj * 1e3, 3))
    if ok:
        l = ops.to_host(lam); X = ops.to_host(R)
        nS = max(abs(w).max(), 1e-300)
        out.update(eval_err=float(abs(l - w[:r]).max() / nS), resid=float(abs(S @ X.T - X.T * l).max() / nS),
                   orth=float(abs(X @ X.T - np.eye(r)).max()))
        # subspace agreement with LAPACK where the r-th gap allows it
        P = X.T @ X; Pr = V[:, :r] @ V[:, :r].T
        out['projector_diff'] = float(abs(P - Pr).max())
    print(json.dumps(out))

for n in (120, 150, 176, 64, 9):
    M = rs.randn(4000, n) * np.exp(-np.arange(n) / 12.0)[None, :]
    M = M @ np.linalg.qr(rs.randn(n, n))[0]
    run('graded', M.T @ M, min(30, n))
n = 150
Q = np.linalg.qr(rs.randn(n, n))[0]
w = np.r_[np.full(10, 5.0), np.full(10, 5.0 - 1e-9), np.linspace(4, 1, 20), rs.rand(n - 40) * 0.5]
run('clusters', (Q * w) @ Q.T, 30)
run('identity', np.eye(n), 30)
run('zero', np.zeros((n, n)), 30)
w = np.r_[np.linspace(3, 1, 25), np.zeros(n - 25)]
run('rank 25, r 30', (Q * w) @ Q.T, 30)
run('diagonal', np.diag(np.arange(n, 0, -1.0)), 30)
A = rs.randn(n, n); run('dense random', A @ A.T, 32)
run('huge scale', (A @ A.T) * 1e200, 30)
run('tiny scale', (A @ A.T) * 1e-200, 30)
run('r = 1', A @ A.T, 1)
