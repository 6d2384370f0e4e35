"""Worker for the gloo tests of the ROW-SHARDED item side of the eigensolver (solver.ItemRows): every rank owns a
slice of the rows of X / Z / V_lock, the block is all-gathered in front of A X and A^T Y is reduce-scattered.  Uses
the TEST-ONLY NumPy double of the device ops; what is under test is the layout + exchange orchestration, at item
counts that do and do not divide by the world size, for a full-rank and a rank-deficient matrix (the `_refill` path)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import scipy.sparse as sp

from numpy_ops import NumpyOps
from polara_amd.csr import nnz_balanced_row_partition
from polara_amd.dist import init_from_env
import polara_amd.solver as solver_module
from polara_amd.solver import svd_topk
from polara_amd.synth import planted_csr

# which method `svd_topk(method=None)` runs in this worker: the test's own variable, set on the module (the product reads no environment)
solver_module.DEFAULT_METHOD = os.environ.get('PK_TEST_SVD_METHOD', 'auto')


def matrices():
    m = planted_csr(900, 257, 20, 12, levels=5, seed=5, min_items=4, max_items=60)
    yield 'planted_257_items', sp.csr_matrix((m['values'], m['indices'], m['indptr']), shape=m['shape']), 12
    rng = np.random.default_rng(3)
    low = (rng.standard_normal((300, 6)) @ rng.standard_normal((6, 64)))
    low[np.abs(low) < 1.2] = 0.0
    low = sp.csr_matrix(low[:, :61])
    yield 'dense_ish_61_items', low, 10
    # rank 5 exactly, 9 vectors asked for: the filtered block is rank-deficient and goes through _refill
    B = rng.standard_normal((200, 5)) @ rng.standard_normal((5, 47))
    yield 'rank5_ask9_47_items', sp.csr_matrix(B), 9


def main():
    comm = init_from_env(backend='gloo')
    ops = NumpyOps()
    out = {}
    for name, M, k in matrices():
        whole = ops.csr(M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64), M.shape)
        _, s1, V1, st1 = svd_topk(ops, whole, k)
        bounds = nnz_balanced_row_partition(M.indptr.astype(np.int64), comm.world)
        P = M[int(bounds[comm.rank]):int(bounds[comm.rank + 1])]
        part = ops.csr(P.indptr.astype(np.int64), P.indices.astype(np.int32), P.data.astype(np.float64), P.shape)
        for shard_items in (True, False):
            before = (comm.n_reduce_scatter, comm.n_allgather, comm.n_allreduce)
            U, s, V, st = svd_topk(ops, part, k, comm=comm, want_u=True, shard_items=shard_items)
            s, V, U = s.numpy(), V.numpy(), U.numpy()
            nz = s1.numpy() > 1e-8 * s1.numpy()[0]
            ok = (V.shape == (M.shape[1], k) and st['converged'] and st['items_sharded'] == shard_items
                  and np.allclose(s[nz], s1.numpy()[nz], rtol=1e-9)
                  and np.abs((V[:, nz] @ V[:, nz].T) - (V1.numpy()[:, nz] @ V1.numpy()[:, nz].T)).max() < 1e-8
                  and np.abs(V.T @ V - np.eye(k)).max() < 1e-8
                  and np.abs(P @ V[:, nz] - U[:, nz] * s[nz]).max() < 1e-8 * s[0])
            if shard_items:
                ok = ok and st['item_rows_per_rank'] == -(-M.shape[1] // comm.world) \
                    and comm.n_reduce_scatter - before[0] == st['gramian_steps'] \
                    and comm.n_allgather - before[1] >= st['gramian_steps']
            else:
                ok = ok and comm.n_reduce_scatter == before[0] and st['item_rows_per_rank'] == M.shape[1]
            out['%s_%s' % (name, 'sharded' if shard_items else 'replicated')] = bool(ok)
            out['%s_%s_steps' % (name, 'sharded' if shard_items else 'replicated')] = (st['gramian_steps'], st1['gramian_steps'])
    # the two-panel exchange of a product (block of 32 columns: two panels of 16, the second one's products next to the first
    # one's sum under RCCL) against one exchange per product: the same factors (not the same bits: which rank's share is added first depends on
    # where an element lies in the exchanged buffer, for gloo as for a ring), two exchanges started per split product
    name, M, k = next(matrices())
    bounds = nnz_balanced_row_partition(M.indptr.astype(np.int64), comm.world)
    P = M[int(bounds[comm.rank]):int(bounds[comm.rank + 1])]
    part = ops.csr(P.indptr.astype(np.int64), P.indices.astype(np.int32), P.data.astype(np.float64), P.shape)
    for shard_items in (True, False):
        res = {}
        for overlap, mode in (('force', 'force'), ('0', 'never')):
            p0 = comm.n_panel_exchanges
            _, s, V, st = svd_topk(ops, part, k, comm=comm, shard_items=shard_items, exchange_overlap=mode)
            res[overlap] = (s.numpy(), V.numpy(), comm.n_panel_exchanges - p0, st['gramian_steps'])
        out['panels_%s' % ('sharded' if shard_items else 'replicated')] = bool(
            np.allclose(res['force'][0], res['0'][0], rtol=1e-12) and np.abs(res['force'][1] @ res['force'][1].T - res['0'][1] @ res['0'][1].T).max() < 1e-10
            and 0 < res['force'][2] <= 2 * res['force'][3] and res['force'][2] % 2 == 0 and res['0'][2] == 0 and res['force'][3] == res['0'][3])
        out['panels_%s_steps' % ('sharded' if shard_items else 'replicated')] = (res['force'][2], res['force'][3])     # (not every block of a locking iteration is 16-divisible)
    # fp32 on the wire (north_star's fp32 Gramian exchange) is an explicit choice: 'auto' means fp64; exchange='f32'
    # rounds every exchanged block — half the bytes, factors to the tolerance of a 1e-6 build; exchange='relaxed' rounds only
    # the products of the late steps of a Lanczos build and verifies in fp64: the factors of the DEFAULT 1e-12 build
    for shard_items in (True, False):
        # the block exchanges: all-gather + reduce-scatter (row-sharded item side) or the all-reduce of Z (replicated; the
        # l x l all-reduces of the Gram matrices ride in the same counter there and stay fp64)
        moved = (lambda: comm.bytes_gathered + comm.bytes_scattered) if shard_items else (lambda: comm.bytes_reduced)
        b0 = moved()
        _, s64, V64, st64 = svd_topk(ops, part, k, comm=comm, shard_items=shard_items)
        b1 = moved()
        _, s32, V32, st32 = svd_topk(ops, part, k, comm=comm, shard_items=shard_items, tol=1e-6, exchange='f32')
        b2 = moved()
        _, s32x, V32x, st32x = svd_topk(ops, part, k, comm=comm, shard_items=shard_items, tol=1e-6)
        per64 = (b1 - b0) / max(st64['gramian_steps'], 1)
        per32 = (b2 - b1) / max(st32['gramian_steps'], 1)
        out['fp32_exchange_%s' % ('sharded' if shard_items else 'replicated')] = bool(
            st64['exchange'] == 'f64' and st32['exchange'] == 'f32' and st32x['exchange'] == 'f64' and st32['converged']
            and np.allclose(s32.numpy(), s64.numpy(), rtol=1e-5) and np.abs(V32.numpy() @ V32.numpy().T - V64.numpy() @ V64.numpy().T).max() < 1e-4
            and per32 < 0.75 * per64)
        out['fp32_exchange_%s_steps' % ('sharded' if shard_items else 'replicated')] = (round(per64), round(per32), st32['gramian_steps'], st32x['gramian_steps'])
    # the relaxed exchange of a default-tolerance Lanczos build: monitors every step (lag 1) so that an estimate below 1e-7 is
    # collected while steps remain; from there the blocks travel in fp32, the verification product in fp64 — converged to
    # 1e-12 on a TRUE residual, the factors of the fp64 build to 1e-11, fewer bytes per step
    moved = lambda: comm.bytes_gathered + comm.bytes_scattered
    kw = dict(comm=comm, method='lanczos', krylov_block=8, monitor_lag=2)
    b0 = moved()
    _, sa, Va, sta = svd_topk(ops, part, k, **kw)
    b1 = moved()
    _, sr, Vr, st_r = svd_topk(ops, part, k, exchange='relaxed', **kw)
    b2 = moved()
    if sta['method'] == 'lanczos' and st_r['method'] == 'lanczos':
        out['relaxed_exchange'] = bool(
            sta['exchange'] == 'f64' and st_r['exchange'] == 'relaxed' and st_r['converged'] and st_r['verified_rel_residual'] <= 1e-12
            and np.allclose(sr.numpy(), sa.numpy(), rtol=1e-11) and np.abs(Vr.numpy() @ Vr.numpy().T - Va.numpy() @ Va.numpy().T).max() < 1e-10
            and (st_r.get('exchange_relaxed_from') is None or (b2 - b1) / st_r['gramian_steps'] < (b1 - b0) / sta['gramian_steps']))
        out['relaxed_exchange_steps'] = (st_r.get('exchange_relaxed_from'), st_r['gramian_steps'], sta['gramian_steps'], b1 - b0, b2 - b1)
    comm.barrier()
    if comm.rank == 0:
        print('SOLVER_DIST_RESULT', out)
    assert all(v for k, v in out.items() if not k.endswith('_steps')), out


if __name__ == '__main__':
    main()
