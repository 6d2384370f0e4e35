"""Randomised check of ops.eigh_top against numpy.linalg.eigh: sizes 8..176, 1 <= r <= min(32, n / 2), spectra drawn from
graded / flat / clustered / rank-deficient / tiny-gap families.  Prints the worst errors and how often the direct kernel's
own verdict sent the call to the Jacobi fallback.  usage: eigh_top_stress.py [cases] [seed]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps, _ptr
from polara_amd import _lib
ops = HipOps('cuda:0')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(eval=0.0, resid=0.0, orth=0.0)
fallbacks, fams = 0, {}
for case in range(n_cases):
    n = int(rs.randint(8, 177))
    r = int(rs.randint(1, max(2, min(32, n // 2) + 1)))
    fam = rs.choice(['graded', 'flat', 'clustered', 'rankdef', 'tinygap', 'gram'])
    Q = np.linalg.qr(rs.randn(n, n))[0]
    if fam == 'graded':
        w = np.exp(-np.arange(n) / rs.uniform(2, 30))
    elif fam == 'flat':
        w = 1.0 + 1e-3 * rs.rand(n)
    elif fam == 'clustered':
        w = np.repeat(rs.rand(max(1, n // 7)) + 0.1, 7)[:n]
        w = np.r_[w, rs.rand(n - len(w))]
    elif fam == 'rankdef':
        k = int(rs.randint(1, n))
        w = np.r_[rs.rand(k) + 0.1, np.zeros(n - k)]
    elif fam == 'tinygap':
        w = np.sort(rs.rand(n))[::-1].copy()
        w[1::2] = w[0::2][:len(w[1::2])] * (1 - 10.0 ** rs.uniform(-14, -6))
    else:
        M = rs.randn(int(rs.randint(n, 4 * n)), n) * (rs.rand(n) ** 3)[None, :]
        S = M.T @ M
        w = None
    if w is not None:
        S = (Q * w) @ Q.T
        S = 0.5 * (S + S.T)
    S *= 10.0 ** rs.uniform(-30, 30)
    Sd = ops.to_device(S)
    # the direct kernel's verdict, then the op (which falls back by itself)
    info = torch.zeros(1, dtype=torch.int32, device=ops.device)
    if ops.lib.pk_eigh_top_supported(n, r):
        R = ops.empty(r, n); lam0 = ops.empty(r); work = ops._work(ops.lib.pk_eigh_top_work_bytes(n))
        _lib.check(ops.lib.pk_eigh_top_f64(ops.stream(), n, _ptr(Sd), n, r, _ptr(R), n, _ptr(lam0), _ptr(work), _ptr(info)), 'top')
    ok = int(info.item())
    fallbacks += 1 - ok
    fams[fam] = fams.get(fam, 0) + (1 - ok)
    lam, C = ops.eigh_top(Sd, r)
    lam, X = ops.to_host(lam), ops.to_host(C)
    wr = np.linalg.eigvalsh(S)[::-1]
    sc = max(abs(wr).max(), 1e-300)
    e = dict(eval=abs(lam - wr[:r]).max() / sc, resid=abs(S @ X - X * lam).max() / sc, orth=abs(X.T @ X - np.eye(r)).max())
    for k in worst:
        if not (e[k] <= worst[k]):
            worst[k] = float(e[k])
    if not (e['eval'] <= 1e-12 and e['resid'] <= 1e-11 and e['orth'] <= 1e-11):
        print('BAD', case, fam, n, r, ok, e)
print(json.dumps(dict(cases=n_cases, worst=worst, fallbacks=fallbacks, fallbacks_by_family=fams)))
