"""Does an L2-resident column panel of the dense block make the SpMM gathers cheaper?  (round 2)
Times A.X and A^T.Y of the ML-20M-shaped / S-1M matrix at full width against the same product done panel by panel
(nc = 16 / 8 columns per launch, i.e. 128 / 64 byte row pieces of a [n x 64] block) and against user-blocked A^T.Y."""
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.csr import popularity_order
ops = HipOps('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
csr, cfg = make_workload(name, device='cuda:0')
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
rank_of, _ = popularity_order(c['indices'], n_items)
A = ops.csr_relabel_cols(ops.csr(c['indptr'], c['indices'], c['values'], c['shape']), rank_of)
At = A.T
out = {'workload': name}

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for tag, M, nsrc, nout in (('AX', A, n_items, n_users), ('AtY', At, n_users, n_items)):
    X = ops.randn(nsrc, 64, 1)
    O = ops.empty(nout, 64)
    out[tag + '_full64'] = timeit(lambda: ops.spmm(M, X, out=O))
    ref = O.clone()
    for pw in (32, 16, 8):
        def panels():
            for c0 in range(0, 64, pw):
                ops.spmm(M, X[:, c0:c0 + pw], out=O[:, c0:c0 + pw])
        out['%s_panels%d' % (tag, pw)] = timeit(panels)
        assert torch.equal(O, ref) or float((O - ref).abs().max()) < 1e-9 * float(ref.abs().max())
    # panel-major storage: each panel contiguous [nsrc x pw] (a gathered piece = one aligned line)
    for pw in (16, 8):
        Xp = [X[:, c0:c0 + pw].contiguous() for c0 in range(0, 64, pw)]
        Op = [ops.empty(nout, pw) for _ in Xp]
        def panels_c():
            for xp, op in zip(Xp, Op):
                ops.spmm(M, xp, out=op)
        out['%s_panels%d_contig' % (tag, pw)] = timeit(panels_c)
    # one narrow block alone (what a single panel costs)
    for pw in (16, 8):
        xp = X[:, :pw].contiguous(); op = ops.empty(nout, pw)
        out['%s_single%d' % (tag, pw)] = timeit(lambda: ops.spmm(M, xp, out=op))
# source-blocked A^T.Y: users in blocks small enough that a 16-column panel of the block is L2-resident
if True:
    Y = ops.randn(n_users, 64, 1)
    for rows_per_block in (16384, 32768, 65536):
        blocks = [(lo, min(n_users, lo + rows_per_block)) for lo in range(0, n_users, rows_per_block)]
        Ab = [ops.csr_rows(A, lo, hi).T for lo, hi in blocks]     # CSC of every user block
        Zs = [ops.empty(n_items, 64) for _ in blocks]
        for pw in (64, 16):
            def blocked():
                for (lo, hi), M, Z in zip(blocks, Ab, Zs):
                    for c0 in range(0, 64, pw):
                        ops.spmm(M, Y[lo:hi, c0:c0 + pw], out=Z[:, c0:c0 + pw])
            out['AtY_userblocks%d_pw%d' % (rows_per_block, pw)] = timeit(blocked, n=3)
        del Ab, Zs
        if name != 'ml20m' and rows_per_block >= 32768:
            break
print(json.dumps(out))
