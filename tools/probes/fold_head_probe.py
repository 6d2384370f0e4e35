"""Fold-in time of the headline matrix: plain groups kernel against the persistent LDS-head instance at several head sizes.
   PK_FOLD_HEAD / PK_FOLD_HEAD_ROWS are read by the library per process: this script re-runs itself per setting."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch, numpy as np
    sys.path.insert(0, '.')
    from polara_amd.ops import HipOps
    from polara_amd.synth import make_workload, csr_to_numpy
    from polara_amd import scoring
    ops = HipOps('cuda:0')
    csr, cfg = make_workload(sys.argv[2], device='cuda:0')
    c = csr_to_numpy(csr); del csr
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    # a factor image with the shape of the real one: random rows, popularity order = item id order of the generator is
    # NOT sorted by popularity, so relabel by counts first (what the model layer does)
    from polara_amd.csr import popularity_order
    rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
    A = ops.csr_relabel_cols(A, rank_of)
    T = A.by_activity()[0] if A.shape[0] >= scoring.ORDER_USERS_MIN else A
    K = int(sys.argv[3])
    Kx = -(-(K + 1) // 4) * 4
    ld = -(-Kx // 32) * 32
    img = torch.randn(c['shape'][1], ld, dtype=torch.float32, device='cuda:0')[:, :Kx]
    out = torch.empty(T.shape[0], Kx, dtype=torch.float64, device='cuda:0')
    for _ in range(5): ops.spmm(T, img, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.spmm(T, img, out=out)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(ms=e0.elapsed_time(e1) / 20, checksum=float(out.sum()))))
else:
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    K = sys.argv[2] if len(sys.argv) > 2 else '50'
    for env in ({'PK_FOLD_HEAD': '0'}, {'PK_FOLD_HEAD_ROWS': '0'}, {'PK_FOLD_HEAD_ROWS': '64'}, {'PK_FOLD_HEAD_ROWS': '256'}, {'PK_FOLD_HEAD_ROWS': '738'}):
        r = subprocess.run([sys.executable, __file__, 'child', wl, K], capture_output=True, text=True, env=dict(os.environ, **env))
        print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
