"""Top-k singular triplets of a (row-sharded) sparse matrix: Chebyshev-filtered subspace iteration
with locking + projector deflation on the item-side Gramian B = A^T A.

This is the MI355X replacement for the `scipy.sparse.linalg.svds(A, k=rank)` call of
`SVDModel.build` (models.py:841-844).  The reference's ARPACK runs a single-vector implicitly
restarted Lanczos on the same operator (n_users >= n_items branch: XH_X = A^T(A x)) with tol=0,
i.e. to machine precision; one Lanczos step = 2 single-threaded CSR mat-vecs.  On a GPU the CSR
stream costs the same for 1 or 64 right-hand sides, so the natural shape is a BLOCK method whose
only heavy operation is the Gramian step  Z = A^T (A Q)  (two SpMM launches, K1) and whose dense
side (Gram matrices, l x l Jacobi eigh, tall-skinny GEMMs, K2) is O(n l^2).

Algorithm (all fp64; results match ARPACK's to ~1e-10 in the projector, far inside the 1e-4
contract, because both are converged to their respective floors):
  X0 = orth(randn(n_items, l)),  l = k + oversample
  repeat:
    Rayleigh-Ritz on the active block:  Y = A X, H = Y^T Y (all-reduce), H = C Theta C^T,
        X <- X C, Y <- Y C, Z = A^T Y (all-reduce)  [= B X, also the first filter step]
    residuals r_j = ||Z_j - theta_j X_j||; lock the leading converged columns
    Chebyshev filter of degree m on the deflated operator P B P, P = I - V_lock V_lock^T,
        damping [0, theta_min(active)]; m chosen so the amplification spread inside the active
        block stays below `spread` (keeps the block numerically full-rank)
    X <- orth(P X)
Multi-GPU: A is row-sharded over users.  The item side is row-sharded too (`ItemRows`): rank r
owns a contiguous slice of the rows of X, Z, V_lock and of every intermediate of the filter, so the Gram
matrices, tall-skinny GEMMs, recurrences and residuals cost n_items / N rows per rank instead of being
recomputed on every rank.  Per Gramian step: ONE all-gather of the block in front of `A X` (the SpMM needs every
item row) and ONE reduce-scatter of `Z_p = A_p^T (A_p X)` behind it — together the volume of the sum all-reduce
they replace — plus l x l (or shorter) all-reduces for Gram matrices and residual norms.  Every rank takes the
same control decisions because they derive from all-reduced data and deterministic kernels.
`shard_items=False` keeps the round-1 layout (item side replicated, one all-reduce of Z per step).
"""
import math
import numpy as np
import torch


class NoConvergence(RuntimeError):
    """The block solver stopped at `max_outer` outer iterations with unconverged leading Ritz pairs — the
    counterpart of ARPACK's `ArpackNoConvergence`, which the reference's `svds` call raises in that situation
    (models.py:844).  Carries the best available factors like ARPACK's exception does."""

    def __init__(self, msg, sigma=None, V=None, stats=None):
        super().__init__(msg)
        self.sigma, self.V, self.stats = sigma, V, stats


class NoComm:
    """Single-process stand-in for the communicator interface (rank, world, allreduce)."""
    rank = 0
    world = 1

    def allreduce(self, t):
        return t


class ItemRows:
    """Row layout of the item-side blocks.  One process (or `shard_items=False`): the whole blocks, every method a
    pass-through.  N ranks: rank r holds rows [r*rows, (r+1)*rows) of the (zero-padded to N*rows) item axis; the
    padding rows are zero in every block and stay zero through every kernel of the solver (linear, row-wise)."""

    def __init__(self, ops, comm, n_items, shard_items=True):
        self.ops, self.comm, self.n = ops, comm, int(n_items)
        self.sharded = (bool(shard_items) and (comm.world > 1 or getattr(comm, '_always', False))      # _always: TorchComm's test switch
                        and hasattr(comm, 'reduce_scatter_rows'))
        self.rows = -(-self.n // comm.world) if self.sharded else self.n
        self.lo = min(self.n, comm.rank * self.rows) if self.sharded else 0
        self.hi = min(self.n, self.lo + self.rows)
        self.padded = self.rows * (comm.world if self.sharded else 1)

    def take(self, full):
        """this rank's rows of a replicated [n_items x b] block"""
        if not self.sharded:
            return full
        out = full.new_zeros((self.rows, full.shape[1]))
        out[:self.hi - self.lo] = full[self.lo:self.hi]
        return out

    def randn(self, l, seed):
        # every rank draws the same seeded block and keeps its rows: the start block does not depend on N
        return self.take(self.ops.randn(self.n, l, seed))

    def full(self, X):
        """[n_items x b] on every rank (what the SpMM gathers from)"""
        if not self.sharded:
            return X
        return self.comm.all_gather_rows(X.contiguous())[:self.n]

    def product(self, At, Y):
        """this rank's rows of  sum_p A_p^T Y_p"""
        ops = self.ops
        if not self.sharded:
            return self.comm.allreduce(ops.spmm(At, Y))
        buf = ops.empty(self.padded, Y.shape[1])
        if self.padded > self.n:
            buf[self.n:].zero_()
        ops.spmm(At, Y, out=buf[:self.n])
        return self.comm.reduce_scatter_rows(buf, self.rows)

    def gram(self, A, B=None):
        G = self.ops.gram(A, B)
        return self.comm.allreduce(G) if self.sharded else G

    def total(self, t):
        """sum over the ranks of per-row-slice partial sums"""
        return self.comm.allreduce(t) if self.sharded else t


def default_block(k, n_items):
    over = max(14, (28 * k + 99) // 100)      # integer ceil(0.28 k): 50 -> 64, 100 -> 128 exactly
    l = k + over
    l = -(-l // 8) * 8
    return int(min(l, n_items))


def _project_out(lay, X, V):
    """X - V (V^T X)."""
    ops = lay.ops
    G = lay.gram(V, X)
    return ops.axpbypcz(1.0, X, -1.0, ops.tsmm(V, G))


def _whiten(lay, X, V_lock=None, passes=2):
    """Orthonormal basis of span(X) (and orthogonal to V_lock) by eigen-whitening, twice: tolerant of
    rank-deficient blocks (tiny eigenvalues are clamped), one Jacobi eigh per pass."""
    ops = lay.ops
    for _ in range(passes):
        if V_lock is not None and V_lock.shape[1] > 0:
            X = _project_out(lay, X, V_lock)
        G = lay.gram(X)
        lam, Cm = ops.eigh_psd(G)
        s = torch.rsqrt(torch.clamp_min(lam, float(1e-300)).clamp_min(lam[0] * 1e-30))
        Cs = ops.scale_cols(Cm.contiguous(), s)
        X = ops.tsmm(X, Cs)
    return X


def _refill(lay, X, V_lock, seed):
    """Orthonormal block of the same width from a numerically RANK-DEFICIENT X (e.g. more vectors asked for
    than the matrix has rank: the filtered copies of null-space directions are pure rounding noise inside the
    range): an orthonormal basis of the numerical range of X (eigen-whitening, directions below 1e-10 of the
    largest dropped) completed by fresh random vectors orthogonal to it and to V_lock."""
    ops = lay.ops
    l = X.shape[1]
    if V_lock is not None and V_lock.shape[1] > 0:
        X = _project_out(lay, X, V_lock)
    lam, Cm = ops.eigh_psd(lay.gram(X))
    lam_h = ops.to_host(lam)
    ng = int((lam_h > lam_h[0] * 1e-20).sum()) if lam_h[0] > 0 else 0
    parts = []
    if ng:
        Cs = ops.scale_cols(Cm[:, :ng].contiguous(), torch.rsqrt(lam[:ng]))
        parts.append(_whiten(lay, ops.tsmm(X, Cs), V_lock, passes=1))
    if ng < l:
        R = lay.randn(l - ng, seed)
        for _ in range(2):
            if parts:
                R = _project_out(lay, R, parts[0])
            R = _whiten(lay, R, V_lock, passes=1)
        parts.append(R)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1).contiguous()


def orthonormalize(lay, X, V_lock=None, seed=12345):
    """Orthonormal basis of span(X), orthogonal to V_lock.  Shifted CholeskyQR3: X <- X R^-1 with
    G + s I = R^T R three times (shift s = 11 (m l + l (l + 1)) u trace(G) on the first pass only), each
    pass one Gram matrix, one l x l Cholesky kernel and one tall-skinny GEMM — the filtered blocks have a
    condition number up to the filter spread (1e7), which the shifted first pass is made for.  The result is
    verified (||Y^T Y - I||): a block that is numerically rank-deficient — a Cholesky pivot breaks down, or the
    three passes end without an orthonormal block — is rebuilt by `_refill`."""
    if not isinstance(lay, ItemRows):     # a bare ops object: one process, whole blocks
        lay = ItemRows(lay, NoComm(), X.shape[0])
    ops = lay.ops
    m, l = lay.n, X.shape[1]
    u = 1.1102230246251565e-16
    info = torch.zeros(3, dtype=torch.int32, device=X.device)
    Y = X
    for p in range(3):
        if V_lock is not None and V_lock.shape[1] > 0:
            Y = _project_out(lay, Y, V_lock)
        G = lay.gram(Y)
        Rinv, _ = ops.chol_rinv(G, 11.0 * (m * l + l * (l + 1)) * u if p == 0 else 0.0, info=info[p:p + 1])
        Y = ops.tsmm(Y, Rinv)
    G = lay.gram(Y)
    err = (G - torch.eye(l, dtype=G.dtype, device=G.device)).abs().max()
    if int(info.abs().sum().item()) != 0 or not (float(err.item()) < 1e-8):
        return _refill(lay, X, V_lock, seed)
    return Y


def _cheb_degree(theta_top, b, spread, m_max):
    """Largest degree whose amplification T_m(x_top) of the top active Ritz value (relative to the
    edge b of the damped interval [0, b], where T_m = 1) stays below `spread`."""
    if b <= 0.0:
        return 2
    x_top = 2.0 * theta_top / b - 1.0
    if x_top <= 1.0 + 1e-12:
        return m_max
    m = int(math.log(2.0 * spread) / math.acosh(x_top))
    return max(2, min(m_max, m))


def svd_topk(ops, A, k, block=None, tol=1e-12, max_outer=200, m_max=24, spread=1e7, seed=0,
             comm=None, want_u=False, verbose=False, even_lock=True, shard_items=True):
    """Returns (U_local | None, sigma[k] desc, V [n_items x k], stats) as device tensors of `ops`.

    A: ops-level CSR of the LOCAL row shard (n_local x n_items).  Convergence: every one of the k
    leading Ritz pairs has ||B x - theta x|| <= tol * theta_1  (B = A^T A, theta = sigma^2).
    """
    comm = comm or NoComm()
    n_items = A.shape[1]
    if not (0 < k <= n_items):
        raise ValueError('k must satisfy 0 < k <= n_items')
    l = int(block or default_block(k, n_items))
    l = max(k, min(l, n_items))
    # the operator of Z = A^T Y: a device matrix offers its user-blocked transpose (ops.BlockedTranspose)
    At = A.transpose_operator() if hasattr(A, 'transpose_operator') else A.T

    lay = ItemRows(ops, comm, n_items, shard_items)
    X = orthonormalize(lay, lay.randn(l, seed))
    V_lock = None
    lam_lock = []
    n_lock = 0
    stats = dict(outer=0, gramian_steps=0, spmm_cols=0, degrees=[], locked_at=[], block=l, converged=False,
                 item_rows_per_rank=lay.rows, items_sharded=lay.sharded)
    theta_host = res_host = None

    def gramian(Xb):
        Z = lay.product(At, ops.spmm(A, lay.full(Xb)))
        stats['gramian_steps'] += 1
        stats['spmm_cols'] += Xb.shape[1]
        return Z

    for it in range(max_outer):
        stats['outer'] = it + 1
        # ---- Rayleigh-Ritz on the active block ------------------------------------------------
        Y = ops.spmm(A, lay.full(X))
        H = comm.allreduce(ops.gram(Y))
        theta, Cm = ops.eigh_psd(H)
        Cm = Cm.contiguous()
        X = ops.tsmm(X, Cm)
        Y = ops.tsmm(Y, Cm)
        Z = lay.product(At, Y)
        stats['gramian_steps'] += 1
        stats['spmm_cols'] += X.shape[1]
        res2 = lay.total(ops.resid_colnorm2(Z, X, theta))
        theta_host = ops.to_host(theta).astype(np.float64)
        res_host = np.sqrt(np.maximum(ops.to_host(res2), 0.0))
        lam1 = lam_lock[0] if lam_lock else float(theta_host[0])
        need = k - n_lock
        # lock the leading run of converged columns (never more than still needed + a few guards)
        thr = tol * lam1
        n_new = 0
        while n_new < len(res_host) and res_host[n_new] <= thr:
            n_new += 1
        if n_new < need and (n_new & 1) and even_lock:
            n_new -= 1       # keep the active block width even: odd widths fall off the paired-column SpMM kernel
        if verbose and comm.rank == 0:
            worst = float(res_host[:max(need, 1)].max() / lam1) if need > 0 else 0.0
            print('[svd] it %3d lock %3d+%-3d active %3d  worst rel.res(first %d) %.2e' %
                  (it, n_lock, n_new, X.shape[1], need, worst))
        if n_new >= need:
            # done: assemble the k leading vectors
            take = need
            Vk = X[:, :take] if V_lock is None else torch.cat([V_lock, X[:, :take]], dim=1)
            lam_k = np.r_[np.asarray(lam_lock, dtype=np.float64), theta_host[:take]]
            stats['converged'] = True
            break
        if n_new > 0 and X.shape[1] - n_new >= max(8, need - n_new):
            newV = X[:, :n_new].contiguous()
            V_lock = newV if V_lock is None else torch.cat([V_lock, newV], dim=1).contiguous()
            lam_lock.extend(float(t) for t in theta_host[:n_new])
            n_lock += n_new
            stats['locked_at'].append((it, n_lock))
            X = X[:, n_new:].contiguous()
            Z = Z[:, n_new:].contiguous()
            theta_host = theta_host[n_new:]
        # ---- Chebyshev filter on P B P, damping [0, b] --------------------------------------------
        b = float(theta_host[-1])
        a0 = float(theta_host[0])
        m = _cheb_degree(a0, b, spread, m_max)
        stats['degrees'].append(m)
        e = 0.5 * b
        c = 0.5 * b
        if e <= 0.0 or a0 <= c:
            # degenerate spectrum estimate (e.g. numerically rank-deficient block): plain power step
            Yc = Z if V_lock is None else _project_out(lay, Z, V_lock)
        else:
            sigma = e / (a0 - c)
            tau = 2.0 / sigma
            Zp = Z if V_lock is None else _project_out(lay, Z, V_lock)
            Xc = X
            Yc = ops.axpbypcz(sigma / e, Zp, -c * sigma / e, Xc)
            for _ in range(2, m + 1):
                sigma_new = 1.0 / (tau - sigma)
                Zc = gramian(Yc)
                if V_lock is not None:
                    Zc = _project_out(lay, Zc, V_lock)
                Yn = ops.axpbypcz(2.0 * sigma_new / e, Zc, -2.0 * sigma_new * c / e, Yc,
                                  -sigma * sigma_new, Xc)
                Xc, Yc = Yc, Yn
                sigma = sigma_new
        X = orthonormalize(lay, Yc, V_lock, seed=seed + 1 + it)
    else:
        # not converged: return the best available (flagged in stats)
        take = min(k - n_lock, X.shape[1])
        Vk = X[:, :take] if V_lock is None else torch.cat([V_lock, X[:, :take]], dim=1)
        lam_k = np.r_[np.asarray(lam_lock, dtype=np.float64), theta_host[:take]]

    Vk = lay.full(Vk[:, :k].contiguous()).contiguous()      # the factors are replicated: scoring needs every item row
    lam_k = np.maximum(lam_k[:k], 0.0)
    order = np.argsort(-lam_k, kind='stable')
    if not np.array_equal(order, np.arange(len(order))):
        Vk = Vk[:, torch.as_tensor(order, device=Vk.device)].contiguous()
        lam_k = lam_k[order]
    sigma_k = np.sqrt(lam_k)
    # residual of the worst of the k leading pairs relative to theta_1 (locked pairs are below tol by construction)
    stats['final_rel_residual'] = float(res_host[:max(1, min(k - n_lock, len(res_host)))].max() / max(lam_k[0], 1e-300))
    stats['tol'] = tol
    U = None
    if want_u:
        U = ops.spmm(A, Vk)
        inv = ops.to_device(np.where(sigma_k > 0, 1.0 / np.maximum(sigma_k, 1e-300), 0.0))
        U = ops.scale_cols(U, inv)
    return U, ops.to_device(sigma_k), Vk, stats
