"""does a narrow SpMM instance (GROUPS = 8 / 16: 8 / 4 lanes per gathered row) make narrow panels cheap?
Per-launch time of A.X and A^T.Y (plain transpose, one launch) at nc = 8 ... 64 with the narrow instances on
(PK_SPMM_NARROW=1: GROUPS = 16 for nc <= 16, 8 for nc <= 32) and off (GROUPS = 4 for every nc <= 64: 16 lanes x 4 columns
per entry, i.e. 3/4 of the lanes idle at nc = 16).  The library reads the switch once per process: one child per setting."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch, numpy as np
    sys.path.insert(0, '.')
    from polara_amd.ops import HipOps
    from polara_amd.synth import make_workload, csr_to_numpy
    from polara_amd.csr import popularity_order
    ops = HipOps('cuda:0')
    csr, cfg = make_workload(sys.argv[2], device='cuda:0')
    c = csr_to_numpy(csr); del csr
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
    A = ops.csr_relabel_cols(A, rank_of)
    At = A.T
    res = {}
    for tag, M, nsrc, nout in (('AX', A, A.shape[1], A.shape[0]), ('AtY', At, A.shape[0], A.shape[1])):
        for nc in (8, 16, 24, 32, 48, 64):
            X = ops.randn(nsrc, nc, 1); O = ops.empty(nout, nc)
            for _ in range(3): ops.spmm(M, X, out=O)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.spmm(M, X, out=O)
            e1.record(); torch.cuda.synchronize()
            res['%s_nc%d' % (tag, nc)] = round(e0.elapsed_time(e1) / 10, 4)
    print(json.dumps(res))
else:
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    out = {}
    for narrow in ('0', '1'):
        r = subprocess.run([sys.executable, __file__, 'child', wl], capture_output=True, text=True, env=dict(os.environ, PK_SPMM_NARROW=narrow))
        out['narrow_' + narrow] = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else r.stderr[-400:]
    print(json.dumps(out, indent=1))
