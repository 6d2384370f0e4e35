"""HOOI / Tucker decomposition of a sparse 3-way tensor on device (CoFFee model build).

Restates `polara.lib.tensor.hooi` (lib/tensor.py:37-96) with the hot pieces moved to HIP:
  * `ttm3d_seq` -> `dttm_seq` (lib/tensor.py:7-19, lib/sparse.py:203-216)  -> pk_ttm_f64 (K5);
  * `svds(unfolding, k=r)` (lib/tensor.py:71,75,79)                        -> Gram + Jacobi eigh (K2).
Same iteration structure, same random initialisation (NumPy RandomState + LAPACK QR on the host —
tiny, and it makes the starting point identical to the reference's), same core-growth stopping rule.
Factor columns are defined up to sign (as in the reference: ARPACK's start vector is random), so
parity is asserted on projectors U U^T, the core norm trace and the resulting recommendations.
"""
import numpy as np
import torch

from .csr import build_row_tasks
from .solver import NoComm


class ModePlan:
    """nnz sorted by one output mode + the wave-task plan for pk_ttm_f64."""

    def __init__(self, ops, idx, val, shape, mode0, mode_u, mode_v, idx_dev=None):
        if idx_dev is not None and hasattr(ops, 'mode_plan'):
            # the device path: idx_dev = the [nnz x 3] index array already in HBM (uploaded once per build)
            self.n0 = int(shape[mode0])
            self.plan, self.idx_u, self.idx_v, order = ops.mode_plan(idx_dev, mode0, mode_u, mode_v, self.n0, split=256)
            ones = val is None or bool(np.all(val == 1.0))
            self.vals = None if ones else ops.to_device(np.asarray(val, dtype=np.float64)).index_select(0, order.long())
            return
        order = np.argsort(idx[:, mode0], kind='stable')
        i0 = idx[order, mode0]
        n0 = int(shape[mode0])
        indptr = np.zeros(n0 + 1, dtype=np.int64)
        np.add.at(indptr, i0 + 1, 1)
        np.cumsum(indptr, out=indptr)
        plan = build_row_tasks(indptr, split=256)
        self.n0 = n0
        self.plan = {k: (ops.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in plan.items()}
        self.plan['n_tasks'] = len(plan['task_row'])
        self.plan['n_long'] = len(plan['long_row'])
        self.idx_u = ops.to_device(idx[order, mode_u].astype(np.int32))
        self.idx_v = ops.to_device(idx[order, mode_v].astype(np.int32))
        ones = val is None or bool(np.all(val == 1.0))
        self.vals = None if ones else ops.to_device(np.asarray(val, dtype=np.float64)[order])


def ttm(ops, mp, u, v):
    """res[n0, ra*rb] with res[i0, j*rb + k] = sum val * u[i_u, j] * v[i_v, k]."""
    return ops.ttm(mp.plan, mp.idx_u, mp.idx_v, mp.vals, u.contiguous(), v.contiguous(), mp.n0)


def _polish(ops, U):
    """One Newton-Schulz step  U <- U (1.5 I - 0.5 U^T U): restores orthonormality lost to the
    squared condition number of the Gram route without rotating the basis."""
    G = ops.gram(U)
    r = G.shape[0]
    Cm = 1.5 * torch.eye(r, dtype=torch.float64, device=G.device) - 0.5 * G
    return ops.tsmm(U, Cm.contiguous())


def left_svd(ops, M, r, want_v=False, comm=None, n_total=None):
    """Top-r left singular vectors / values of dense M (n x m), descending; optionally V^T (r x m).
    Mirrors what `svds(M, k=r)` returns to hooi (after its [::-1] reordering).
    With `comm` the ROWS of M are sharded over ranks (the user mode): the m x m Gram matrix is
    all-reduced, the eigenproblem is solved redundantly and each rank keeps its rows of U."""
    n, m = M.shape
    n_all = n if n_total is None else n_total
    if r > min(n_all, m):
        raise ValueError('rank %d exceeds min(shape)=%d' % (r, min(n_all, m)))
    if comm is not None and comm.world > 1:
        if n_all < m:
            raise NotImplementedError('row-sharded unfolding with fewer rows than columns')
        lam, Cm = ops.eigh_psd(comm.allreduce(ops.gram(M)))
        W = Cm[:, :r].contiguous()
        s = torch.sqrt(torch.clamp_min(lam[:r], 0.0))
        U = ops.tsmm(M, W)
        U = ops.scale_cols(U, torch.where(s > 0, 1.0 / s, torch.zeros_like(s)))
        G = comm.allreduce(ops.gram(U))          # Newton-Schulz polish with the GLOBAL Gram
        Cn = 1.5 * torch.eye(r, dtype=torch.float64, device=G.device) - 0.5 * G
        U = ops.tsmm(U, Cn.contiguous())
        return U, s, (W.t().contiguous() if want_v else None)
    if n >= m:
        lam, Cm = ops.eigh_psd(ops.gram(M))
        W = Cm[:, :r].contiguous()
        s = torch.sqrt(torch.clamp_min(lam[:r], 0.0))
        U = ops.tsmm(M, W)
        U = ops.scale_cols(U, torch.where(s > 0, 1.0 / s, torch.zeros_like(s)))
        U = _polish(ops, U)
        Vt = W.t().contiguous() if want_v else None
    else:
        Mt = M.t().contiguous()
        lam, Cm = ops.eigh_psd(ops.gram(Mt))
        U = Cm[:, :r].contiguous()
        s = torch.sqrt(torch.clamp_min(lam[:r], 0.0))
        Vt = None
        if want_v:
            Vt = ops.small_mm(U, M, transA=True)          # r x m  = diag(s) V^T
            inv = torch.where(s > 0, 1.0 / s, torch.zeros_like(s))
            Vt = (Vt * inv[:, None]).contiguous()
    return U, s, Vt


def hooi(ops, idx, val, shape, core_shape, num_iters=25, growth_tol=0.01, seed=None, verbose=False,
         comm=None, user_range=None, item_inv=None):
    """Returns (u0, u1, u2, core, trace): device fp64 factors [n_mode x r_mode] with orthonormal
    columns ordered by descending singular value, core [r0 x r1 x r2], and the per-iteration core
    norms (lib/tensor.py:82-88).

    Multi-GPU: pass `comm` and `user_range=(lo, hi)`; `idx` then holds only the nnz of
    users [lo, hi) (user indices already re-based to 0), the mode-0 factor u0 is returned as this
    rank's [hi-lo x r0] row block, the mode-1 / mode-2 TTMs (which reduce over users) are
    all-reduced, and every rank holds identical u1, u2, core.
    `item_inv`: see the start-block comment below."""
    comm = comm or NoComm()
    idx = np.asarray(idx)
    r0, r1, r2 = (int(r) for r in core_shape)
    n0, n1, n2 = (int(s) for s in shape)
    n0_total = n0
    if user_range is not None:
        n0 = int(user_range[1] - user_range[0])   # local rows of the user mode
        shape = (n0, n1, n2)
    # same random start as the reference (lib/tensor.py:57-63)
    random_state = np.random if seed is None else np.random.RandomState(seed)
    u1 = np.linalg.qr(random_state.rand(n1, r1), mode='reduced')[0]
    u2 = np.linalg.qr(random_state.rand(n2, r2), mode='reduced')[0]
    if item_inv is not None:
        # the caller relabelled the item mode (internal row i = external item item_inv[i]): permute the
        # start block the same way, so the iteration is the reference's up to that relabelling
        u1 = np.ascontiguousarray(u1[item_inv])
    u1 = ops.to_device(u1)
    u2 = ops.to_device(u2)

    # (mode0 ; first matrix mode ; second matrix mode) as in lib/tensor.py:70,74,78
    idx_dev = None
    if hasattr(ops, 'mode_plan'):
        idx_dev = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(ops.device)
    mp0 = ModePlan(ops, idx, val, shape, 0, 2, 1, idx_dev)
    mp1 = ModePlan(ops, idx, val, shape, 1, 2, 0, idx_dev)
    mp2 = ModePlan(ops, idx, val, shape, 2, 1, 0, idx_dev)

    g_norm_old = 0.0
    trace = []
    ss = vv = u0 = None
    for i in range(num_iters):
        u0, _, _ = left_svd(ops, ttm(ops, mp0, u2, u1), r0, comm=comm, n_total=n0_total)   # rows = local users
        u1, _, _ = left_svd(ops, comm.allreduce(ttm(ops, mp1, u2, u0)), r1)
        u2, ss, vv = left_svd(ops, comm.allreduce(ttm(ops, mp2, u1, u0)), r2, want_v=True)
        g_norm_new = float(torch.linalg.vector_norm(ss).item())
        g_growth = (g_norm_new - g_norm_old) / g_norm_new
        g_norm_old = g_norm_new
        trace.append(g_norm_new)
        if verbose:
            print('Step %i of %i, growth of the core: %f' % (i + 1, num_iters, g_growth))
        if g_growth < growth_tol:
            break
    core = (ss[:, None] * vv).reshape(r2, r1, r0).permute(2, 1, 0).contiguous()
    return u0, u1, u2, core, trace
