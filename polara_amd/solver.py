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

DEFAULT_PRODUCTS = 'f64'   # what `svd_topk(products='auto')` means: 'f64', or 'relaxed' = the late steps of a one-process Lanczos build gather fp32 images
DEFAULT_METHOD = 'auto'    # what `svd_topk(method=None)` means; tests pin 'lanczos' / 'subspace' here (a module attribute, not the environment)
MAX_KRYLOV_COLS = 4096     # widest operand of pk_gram_f64 (csrc/dense.hip): the Krylov basis of a block Lanczos build stays below it


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

    def __init__(self, ops, comm, n_items, shard_items=True, exchange_dtype=None, overlap='auto'):
        self.ops, self.comm, self.n = ops, comm, int(n_items)
        if overlap not in ('auto', 'never', 'force'):
            raise ValueError("overlap must be 'auto', 'never' or 'force'")
        self.overlap = overlap          # the two-panel exchange of `product`: by the cost model / never / whenever possible (tests)
        # payload of the two big exchanges of a Gramian step (all-gather of X, reduce-scatter / all-reduce of Z): None = the
        # blocks as they are (fp64); torch.float32 = rounded to fp32 on the wire (half the bytes, north_star's "fp32 Gramian
        # all-reduce").  An fp32 payload perturbs every product by ~6e-8 of its norm: good for builds to a tolerance of 1e-6
        # (the contract's 1e-4 on singular values and scores with two digits to spare), NOT for the default 1e-12 — the
        # inexact-Krylov experiment of round 4 stalled at 5e-12 with rounded products.  svd_topk picks it from
        # the tolerance (`exchange='auto'`); the small l x l all-reduces stay fp64 either way.
        self.exchange_dtype = exchange_dtype
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
        if self.exchange_dtype is not None and X.dtype != self.exchange_dtype:
            return self.comm.all_gather_rows(X.to(self.exchange_dtype).contiguous())[:self.n].to(X.dtype)
        return self.comm.all_gather_rows(X.contiguous())[:self.n]

    def product(self, At, Y):
        """this rank's rows of  sum_p A_p^T Y_p.
        More than one rank, blocks whose exchange is long enough to be worth hiding: the block goes in TWO column panels — the
        products of the second panel run while the first one's sum travels (RCCL: the collective is started asynchronously
        on the library's stream; under gloo the panels are exchanged one after the other, same results).  The split is not
        free: two launches of A^T Y per user block instead of one, and a 32-column panel of A^T Y costs ~0.7 of a 64-column
        launch, not half (the narrow instances pay off for A X, whose dense block sits on chip):
        +0.23 ms per step on ML-20M-shaped (tools/probes/overlap_one_rank.py).  So it is taken when the modelled exchange of
        the block — 2 (N-1)/N n_items nc 8 bytes at 100 GB/s — reaches 0.4 ms (S-1M on 8 ranks: 0.9 ms; ML-20M-shaped:
        0.14-0.24 ms, not taken).  `self.overlap`: 'never', or 'force' = whenever there is something to exchange (tests)."""
        ops, comm = self.ops, self.comm
        nc = Y.shape[1]
        mode = self.overlap
        exchanging = comm.world > 1 or (mode == 'force' and getattr(comm, '_always', False))
        from .machine_model import value as mm
        worth = mode == 'force' or 2.0 * (comm.world - 1) / max(comm.world, 1) * self.n * nc * 8.0 / mm('xgmi_bus_Bps') >= 4e-4
        split = (exchanging and worth and mode != 'never' and nc >= 32 and nc % 16 == 0 and hasattr(comm, 'allreduce_start')
                 and not hasattr(At, 'matvec'))
        panels = ((0, nc // 2), (nc // 2, nc)) if split else ((0, nc),)
        moving = comm.world > 1 or getattr(comm, '_always', False)
        xd = self.exchange_dtype if (moving and self.exchange_dtype is not None) else None
        wire = (lambda t: t.to(xd)) if xd is not None else (lambda t: t)          # the block as it travels
        home = (lambda t: t.to(torch.float64)) if xd is not None else (lambda t: t)
        if not self.sharded:
            if not split:
                return home(comm.allreduce(wire(ops.spmm(At, Y))))
            pending = [comm.allreduce_start(wire(ops.spmm(At, Y[:, c0:c1])), count=(c0 == 0)) for c0, c1 in panels]
            return home(torch.cat([h.wait() for h in pending], 1))
        pending = []
        for c0, c1 in panels:
            buf = ops.empty(self.padded, c1 - c0)
            if self.padded > self.n:
                buf[self.n:].zero_()
            ops.spmm(At, Y if not split else Y[:, c0:c1], out=buf[:self.n])
            if not split:
                return home(comm.reduce_scatter_rows(wire(buf), self.rows))
            pending.append(comm.reduce_scatter_rows_start(wire(buf), self.rows, count=(c0 == 0)))
        return home(torch.cat([h.wait() for h in pending], 1))

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


def orthonormalize(lay, X, V_lock=None, seed=12345, defer=None, first_gram=None):
    """Orthonormal basis of span(X), orthogonal to V_lock.  Shifted CholeskyQR3: X <- X R^-1 with
    G + s I = R^T R three times (shift s = 11 (m l + l (l + 1)) u trace(G) on the first pass only), each
    pass one Gram matrix, one l x l Cholesky kernel and one tall-skinny GEMM — the filtered blocks have a
    condition number up to the filter spread (1e7), which the shifted first pass is made for.  The result is
    verified (||Y^T Y - I||): a block that is numerically rank-deficient — a Cholesky pivot breaks down, or the
    three passes end without an orthonormal block — is rebuilt by `_refill`.
    `defer` (a device tensor of 2 doubles): no host read here — the Cholesky verdicts are ADDED to defer[0] and the
    orthonormality error is max-ed into defer[1]; the caller reads them when it next talks to the host anyway and owns
    the consequences (block Lanczos: one read per convergence check).  `first_gram` (a list): receives the Gram matrix of
    the block after its first projection — the coupling matrix of the Lanczos residual block."""
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
        if p == 0 and first_gram is not None:
            first_gram.append(G)
        Rinv, _ = ops.chol_rinv(G, 11.0 * (m * l + l * (l + 1)) * u if p == 0 else 0.0, info=info[p:p + 1])
        Y = ops.tsmm(Y, Rinv)
    G = lay.gram(Y)
    err = (G - torch.eye(l, dtype=G.dtype, device=G.device)).abs().max()
    if defer is not None:
        defer[0] += info.abs().sum().to(defer.dtype)
        defer[1] = torch.maximum(defer[1], torch.nan_to_num(err, nan=1.0, posinf=1.0).to(defer.dtype))
        return Y
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


class _Lazy:
    """something built at first use"""

    def __init__(self, make):
        self.make = make


class _Gramian:
    """B = A^T A of the (row-sharded) sparse matrix: the operator of the build.  `ritz` hands back H = X^T B X together
    with the carrier Y = A X, from which B (X C) = A^T (Y C) follows with ONE more product — Rayleigh-Ritz and the first
    filter step share a Gramian step."""

    def __init__(self, ops, A, At, lay, comm, stats):
        self.ops, self.A, self._At, self.lay, self.comm, self.stats = ops, A, At, lay, comm, stats

    @property
    def At(self):
        # the operator of Z = A^T Y, built when the first product needs it (a callable: `svd_topk` does not pay for the
        # transposed image of a build whose recurrence runs inside the library, on the library's own image)
        if isinstance(self._At, _Lazy):
            self._At = self._At.make()
        return self._At

    def _count(self, cols):
        self.stats['gramian_steps'] += 1
        self.stats['spmm_cols'] += cols

    def ritz(self, X):
        Y = self.ops.spmm(self.A, self.lay.full(X))
        return self.comm.allreduce(self.ops.gram(Y)), Y

    def rotate(self, Y, Cm):
        Z = self.lay.product(self.At, self.ops.tsmm(Y, Cm))
        self._count(Cm.shape[1])
        return Z

    def apply(self, X):
        Z = self.lay.product(self.At, self.ops.spmm(self.A, self.lay.full(X)))
        self._count(X.shape[1])
        return Z


class _Dense:
    """A small symmetric PSD matrix T (replicated on every rank) as the operator: the projected problem of the block
    Lanczos method.  T X = T^T X is one `gram` launch."""

    def __init__(self, ops, T, stats):
        self.ops, self.T, self.stats = ops, T, stats

    def ritz(self, X):
        Z = self.ops.gram(self.T, X)
        H = self.ops.gram(X, Z)
        self.stats['steps'] += 1
        return 0.5 * (H + H.t()), Z

    def rotate(self, Z, Cm):
        return self.ops.tsmm(Z, Cm)

    def apply(self, X):
        self.stats['steps'] += 1
        return self.ops.gram(self.T, X)


def _subspace_iteration(op, lay, k, X, tol, max_outer, m_max, spread, seed, stats, verbose=False, even_lock=True,
                        rank0=True):
    """Chebyshev-filtered subspace iteration with locking on the operator `op` (ritz / rotate / apply), from the
    orthonormal start block X (rows of `lay` x l).  Returns (basis [rows x >= k]: locked vectors then the active block,
    Ritz values of the basis columns (host), residual norms of the active block (host), n_lock, converged)."""
    ops = lay.ops
    V_lock = None
    lam_lock = []
    n_lock = 0
    theta_host = res_host = None
    converged = False
    for it in range(max_outer):
        stats['outer'] = stats.get('outer', 0) + 1
        # ---- Rayleigh-Ritz on the active block ------------------------------------------------
        H, carrier = op.ritz(X)
        theta, Cm = ops.eigh_psd(H)
        Cm = Cm.contiguous()
        X = ops.tsmm(X, Cm)
        Z = op.rotate(carrier, Cm)
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
        if verbose and rank0:
            worst = float(res_host[:max(need, 1)].max() / lam1) if need > 0 else 0.0
            print('[svd] it %3d lock %3d+%-3d active %3d  worst rel.res(first %d) %.2e' %
                  (it, n_lock, n_new, X.shape[1], need, worst))
        if n_new >= need:
            converged = True
            break
        if n_new > 0 and X.shape[1] - n_new >= max(8, need - n_new):
            newV = X[:, :n_new].contiguous()
            V_lock = newV if V_lock is None else torch.cat([V_lock, newV], dim=1).contiguous()
            lam_lock.extend(float(t) for t in theta_host[:n_new])
            n_lock += n_new
            stats.setdefault('locked_at', []).append((it, n_lock))
            X = X[:, n_new:].contiguous()
            Z = Z[:, n_new:].contiguous()
            theta_host = theta_host[n_new:]
            res_host = res_host[n_new:]
        # ---- Chebyshev filter on P B P, damping [0, b] --------------------------------------------
        b = float(theta_host[-1])
        a0 = float(theta_host[0])
        m = _cheb_degree(a0, b, spread, m_max)
        stats.setdefault('degrees', []).append(m)
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
                Zc = op.apply(Yc)
                if V_lock is not None:
                    Zc = _project_out(lay, Zc, V_lock)
                Yn = ops.axpbypcz(2.0 * sigma_new / e, Zc, -2.0 * sigma_new * c / e, Yc,
                                  -sigma * sigma_new, Xc)
                Xc, Yc = Yc, Yn
                sigma = sigma_new
        X = orthonormalize(lay, Yc, V_lock, seed=seed + 1 + it)
    basis = X if V_lock is None else torch.cat([V_lock, X], dim=1)
    lam_all = np.r_[np.asarray(lam_lock, dtype=np.float64), theta_host]
    return basis, lam_all, res_host, n_lock, converged


def _next_lanczos_block(lay, W, Qbuf, N, C, flags):
    """The next block of the Krylov basis from W = B Q_j and C = Q^T W (already all-reduced: the block column of T):
    shifted CholeskyQR3 of the projected block, RE-projected against the whole basis in every pass — near convergence the
    residual block has singular values over ten orders of magnitude, and what a pass scales up by 1/sigma it also scales up
    along Q; a projection after the scaling is what keeps Q^T Q = I to rounding (`orthonormalize` does the same for the
    filtered blocks).  The block lives where it will stay — columns [N, N + b) of the basis buffer `Qbuf` — so that a
    re-projection pass needs ONE Gram product: [Q | Y]^T Y gives the coefficients C_p = Q^T Y of the re-projection AND the
    l x l Gram matrix of Y in one launch and, row-sharded, ONE all-reduce (four per step instead of six).  The Gram matrix
    of a pass is that of Y BEFORE its re-projection; the two differ by C_p^T C_p, and C_p is what a pass LEFT along Q: at
    most ~1e-6 after the first pass (u times the condition of the residual block), so 1e-12 relative — a pass brings
    delta to ~delta^2 either way, and the pairs are verified by a true product at the end.
    Per pass: that Gram product (pass 0: the l x l one of the projected block), one fused `Y - Q C` product, one Cholesky
    kernel, one tall-skinny product.  No host read: the Cholesky verdicts and the distance of the last pass's Gram matrix
    from I go to `flags` (one launch).  Returns S = W_perp^T W_perp, the coupling behind the residuals of the Ritz pairs."""
    ops = lay.ops
    m, l = lay.n, W.shape[1]
    u = 1.1102230246251565e-16
    fused = hasattr(ops, 'orth_check')
    if fused:      # the Cholesky verdicts of the three passes: a buffer the check kernel leaves zeroed for the next block
        info = getattr(ops, '_orth_info', None)
        if info is None:
            info = ops._orth_info = torch.zeros(3, dtype=torch.int32, device=W.device)
    else:
        info = torch.zeros(3, dtype=torch.int32, device=W.device)
    Qall, Ydst = Qbuf[:, :N], Qbuf[:, N:N + l]
    S = G = None
    for p in range(3):
        if p == 0:
            Yp = ops.tsmm_sub(W, Qall, C)
            S = G = lay.gram(Yp)
        else:
            M = lay.gram(Qbuf[:, :N + l], Ydst)        # rows [0, N): Q^T Y; rows [N, N + l): Y^T Y
            G = M[N:]
            Yp = ops.tsmm_sub(Ydst, Qall, M[:N])
        Rinv, _ = ops.chol_rinv(G, 11.0 * (m * l + l * (l + 1)) * u if p == 0 else 0.0, info=info[p:p + 1])
        ops.tsmm(Yp, Rinv, out=Ydst)
    if fused:
        ops.orth_check(G, info, flags)       # one launch: verdicts summed, max |G - I|, `info` cleared
    else:
        err = (G - torch.eye(l, dtype=G.dtype, device=G.device)).abs().max()
        flags[0] += info.abs().sum().to(flags.dtype)
        flags[1] = torch.maximum(flags[1], torch.nan_to_num(err, nan=1.0, posinf=1.0).to(flags.dtype))
    return S


class _LanczosBreakdown(RuntimeError):
    """The block Krylov recurrence cannot continue (rank-deficient residual block, lost orthogonality, space exhausted):
    `svd_topk` then runs the filtered subspace iteration, which has a rebuild path for exactly these matrices."""


def _ritz_check(ops, Tj, S, X0, k, b, est_tol, prior, seed, inner, final=False, width=None, lam0=None):
    """The k leading Ritz pairs of T_j and estimates of their residuals (one per pair, relative to theta_1).
    The pairs come in stages: while the outer method is far from converged an ESTIMATE is all a check needs, so the nested
    iteration first runs to a loose tolerance and is tightened (warm) only while its own residual, not the coupling to the
    next block, is what limits the estimate.  Returns dict(est, coupling, conv, basis, lam_all, Yk, lam_k)."""
    N = Tj.shape[0]
    cold_width = None
    if X0 is None:           # a COLD look — the first unit vectors (the first b of them are the seed of this very Krylov space); the
        cold_width = min(N, max(b, width or b))      # library starts it from the eigenvectors of T's leading block (pk_sym_eig_topk_f64)
    t_in = max(0.3 * est_tol, 1e-4 if prior is None else 0.03 * prior)
    if final:                # the look at the step the pairs are predicted to have converged at: straight to the end
        t_in = 0.3 * est_tol
    r0 = -1.0 if (lam0 is None or prior is None) else float(prior)     # the start pairs' residual w.r.t. THIS T: their old coupling estimate
    while True:
        basis, lam_all, res_in, n_lock, conv_in = ops.sym_eig_topk(Tj, k, X0, t_in, 200, seed, inner, lam0=lam0, r0_rel=r0, width=cold_width)
        cold_width = None
        lam0, r0 = None, -1.0                # (a tightened pass starts from pairs of THIS T: no Rayleigh-Ritz step is skipped)
        X0 = basis.contiguous()
        Yk = X0[:, :k].contiguous()
        TY = ops.gram(Tj, Yk)
        lam_k = torch.as_tensor(lam_all[:k].copy(), device=Yk.device)
        r_in2 = ops.resid_colnorm2(TY, Yk, lam_k)
        yl = Yk[N - b:].contiguous()
        c2 = (yl * ops.small_mm(S, yl)).sum(0)
        both = ops.to_host(torch.stack([r_in2, c2]))
        lam1 = max(float(lam_all[0]), 1e-300)
        est = np.sqrt(np.maximum(both[0] + both[1], 0.0)) / lam1
        coupling = float(np.sqrt(max(both[1].max(), 0.0)) / lam1)
        if t_in <= 0.3 * est_tol or coupling >= 4.0 * t_in or not conv_in:
            break
        t_in = max(0.3 * est_tol, 0.1 * coupling)
    return dict(est=est, worst=float(est.max()), coupling=coupling, conv=bool(conv_in), basis=X0, lam_all=lam_all, Yk=Yk,
                lam_k=lam_k)


def _raise_on_breakdown(fl, j):
    """fl = [sum of Cholesky verdicts, max distance of the last pass's Gram matrix from I] of the block recurrence"""
    if fl[0] != 0 or not (fl[1] < 1e-4):     # the Gram matrix BEFORE the last pass: 1e-4 there is 1e-8 after it
        raise _LanczosBreakdown('residual block lost rank at step <= %d (Cholesky verdicts %g, orthonormality %.1e)' % (j, fl[0], fl[1]))


class _Monitor:
    """A convergence check of the block Lanczos build that runs NEXT TO the following Gramian steps: a worker thread
    drives the nested solve (one C call that releases the interpreter lock, hundreds of microsecond kernels) on a side
    stream while the main thread keeps enqueueing the sparse products, which leave most of the chip's launch slots to it.
    The result is collected at a FIXED number of steps after the launch (blocking if need be), so every rank of a sharded
    build takes its decisions at the same steps from the same numbers."""

    def __init__(self, ops, j, Tj, S, X0, k, b, est_tol, prior, seed, inner, flags=None, width=None, lam0=None):
        import threading
        self.j = j
        self.out = self.err = None
        # the recurrence's breakdown flags as they stand after step j: read by the WORKER (a host read on the calling
        # thread would drain the main stream at every launch of a monitor and leave it empty while the next step is enqueued)
        fsnap = flags.clone() if flags is not None else None
        dev = getattr(ops, 'device', None)
        cuda = dev is not None and getattr(dev, 'type', 'cpu') == 'cuda'
        side = None
        if cuda:
            side = ops.monitor_stream() if hasattr(ops, 'monitor_stream') else ops.aux_streams(1)[0]
            side.wait_stream(torch.cuda.current_stream(dev))      # the snapshot of T_j and S is complete
            for t in (Tj, S, X0, fsnap):
                if t is not None:
                    t.record_stream(side)

        def work():
            try:
                if cuda:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(side):
                        if fsnap is not None:
                            _raise_on_breakdown(ops.to_host(fsnap), j)
                        self.out = _ritz_check(ops, Tj, S, X0, k, b, est_tol, prior, seed, inner, width=width, lam0=lam0)
                        side.synchronize()
                else:
                    if fsnap is not None:
                        _raise_on_breakdown(ops.to_host(fsnap), j)
                    self.out = _ritz_check(ops, Tj, S, X0, k, b, est_tol, prior, seed, inner, width=width, lam0=lam0)
            except BaseException as exc:        # re-raised by join() in the thread that owns the build
                self.err = exc
        self.thread = threading.Thread(target=work, name='pk-lanczos-monitor', daemon=True)
        self.thread.start()

    def join(self):
        self.thread.join()
        if self.err is not None:
            raise self.err
        return self.out


def _block_lanczos(ops, A, At, lay, comm, k, l, tol, seed, stats, verbose, max_steps, m_max, spread, even_lock, kb=None,
                   monitor_lag=None, t_step=None, steps_model=None, first_look=None, products='f64'):
    """Block Lanczos on B = A^T A with FULL reorthogonalisation and Rayleigh-Ritz over the WHOLE Krylov space
    span[X, B X, ..., B^(q-1) X] — the Krylov-class method behind the reference's `svds` (ARPACK: single-vector implicitly
    restarted Lanczos on the same operator, models.py:844), in the block form a GPU wants.  One Gramian step per block:
        W = B Q_j;   T[:, j] = Q^T W;   Q_(j+1) R = W - Q T[:, j]   (projection + shifted CholeskyQR3, `_next_lanczos_block`)
    The projected matrix T = Q^T B Q (block tridiagonal up to rounding; every block column is COMPUTED, not assumed) is
    a small dense symmetric matrix on the device; its k leading pairs come from the filtered subspace iteration with T as
    the operator (`ops.sym_eig_topk`: driven from C++, warm-started from the previous check).  A Ritz pair (theta, Q y) has
    the residual  Q (T y - theta y) + Q_(j+1) R y_last, so
        ||B x - theta x||^2 = ||T y - theta y||^2 + y_last^T (W_perp^T W_perp) y_last
    — everything in coefficient space.  Checks while the method is still far from converged are MONITORS: they run on a side
    stream next to the following Gramian steps (`_Monitor`) and only serve to predict the step at which the pairs will
    have converged; at that step the pairs are computed on the main stream (warm from the last monitor) and VERIFIED by one
    true product B V (same certificate as the subspace iteration's: `final_rel_residual` is measured, not estimated).
    On the ML-20M-shaped matrix, rank 50, block 64: 14 Gramian steps + 1 verification against 37 of the filtered
    subspace iteration (the Krylov space keeps every block: its Ritz values beyond the block width deflate the tail of
    the planted spectrum, which a fixed-width filter has to damp uniformly)."""
    import time
    n = lay.n
    b = int(kb or l)            # width of a Krylov block; the nested solves keep the width l (k + guard vectors)
    # the Gram products against the whole basis (Q^T W, the re-projections, the nested T X) take at most
    # MAX_KRYLOV_COLS columns (pk_gram_f64's limit): a space that would outgrow them is a breakdown like any other —
    # the filtered subspace iteration takes over — not an error out of a kernel launcher (rank 100: b = 128, 32 blocks)
    qcap = min(n // b, MAX_KRYLOV_COLS // b)
    if qcap < 4 or max_steps < 4:
        raise _LanczosBreakdown('Krylov space of at most %d blocks' % qcap)
    qcap = min(qcap, max_steps)
    gop = _Gramian(ops, A, At, lay, comm, stats)
    # One process, a device matrix: the steps of the recurrence run inside the library (pk_lanczos_steps — the ONE statement of
    # the step, csrc/driver.hip::lanczos_step, which the coarse build runs too): a step is ~40 dependent launches, and at the
    # narrow blocks of round 6 the Python composition below spent more host time enqueueing them than the GPU spent running
    # them.  The looks, their monitors and every decision stay here.  Sharded builds (collectives inside the step), host-side
    # operators and the CPU double of the tests take the composition (`_next_lanczos_block`).
    rec = None
    if _library_recurrence(ops, A, comm, lay.sharded):
        rec = ops.lanczos_recurrence(A, b)
    halves = comm.world > 1 or getattr(comm, 'split_step', False)      # the library's step with the sum over the ranks in the middle
    stats['recurrence'] = 'library' if rec is not None else 'composition'
    if rec is not None:
        # one process: nothing is exchanged, but the same relaxation applies to the PRODUCTS — the late steps gather fp32 images
        # of their dense blocks (half the bytes per gathered row; pk_lanczos_steps(rounded)); `lay.relaxed` / `lay.exchange_dtype`
        # carry the state for both forms (gate, verification in fp64, fall-back)
        lay.relaxed = (products == 'relaxed' and b % 4 == 0) if not halves else bool(getattr(lay, 'relaxed', False))
    S_buf = ops.zeros(b, b) if rec is not None else None
    cap = min(qcap, 20)
    Qbuf = ops.empty(lay.rows, cap * b)
    T = ops.zeros(cap * b, cap * b)
    flags = ops.zeros(2)                  # [sum of Cholesky verdicts, max distance of the last pass's Gram matrix from I]
    Q1 = orthonormalize(lay, lay.randn(b, seed))
    Qbuf[:, :b] = Q1
    warm = warm_lam = None
    inner = dict(steps=0, outer=0, checks=0)
    stats['nested'] = inner
    hist = []                              # (step, worst relative residual estimate of the k leading pairs)
    # Looks are the expensive part of a build with narrow blocks: a nested solve is ~5 ms of dependent small kernels, a step
    # of a 16-column block 0.55 ms (round 6; at the 1.7 ms steps of round 4's 64-column blocks it was the other way round).  So:
    # ONE monitor at 0.7 of the modelled number of steps where steps are cheap (0.5 where they are not: below), collected after as many steps as its solve takes next to the
    # products (never blocking: lag = look time / modelled step time), the final look where that estimate and a PRIOR rate put
    # convergence (a late look wastes cheap steps, an early one costs a whole nested solve and a second look), further
    # monitors only when convergence is far.  Every number is derived from the all-reduced entry count: all ranks decide alike.
    # monitor_lag = 0: no monitors, every look on the calling thread (the form the C++ statement takes).
    from .machine_model import value as mm
    if monitor_lag is None:
        # (a look at nested width l costs ~(l / 64)^1.5 of the 64-wide one: rank 100 waits 24 / 13 ms where rank 50 waits 17 / 5.4)
        monitor_lag = 3 if not t_step else int(min(10, max(3, math.ceil(mm('look_wall_s') * max(1.0, l / 64.0) ** 1.5 / t_step))))
    LAG = int(monitor_lag)
    stats['monitor_lag'] = LAG
    use_monitor = LAG > 0
    first = max(4, -(-2 * k // b) + 2, -(-l // b))
    if first_look is not None:
        first = max(-(-l // b), int(first_look))
    elif steps_model:
        # cheap steps (a look is worth five or more of them): one late monitor; expensive steps (S-1M: 2.9 ms): an early chain of
        # warm monitors — ML-20M-shaped rank 50 30.4 -> 26.9 ms and rank 100 58.5 -> 49.7 ms with 0.7, S-1M 88 -> 93 ms (so: 0.5)
        first = max(first, int(math.ceil((0.7 if int(monitor_lag) >= 5 else 0.5) * steps_model)))
    prior_rate = 0.8 * 1.72 * (b / 16.0) ** 0.27      # natural log per step, late phase: x5.6 / x8.2 / x12 per step at b = 16 / 32 / 64 (measured), less a fifth
    next_look = min(first, qcap)           # the step of the next monitor, or of the final check once the rate is known
    look_is_final = not use_monitor
    est_tol = tol
    result = None
    monitor = None
    j = 0

    def snapshot(N):
        Tj = T[:N, :N]
        return (0.5 * (Tj + Tj.t())).contiguous()

    def breakdown_check(j):
        _raise_on_breakdown(ops.to_host(flags), j)

    relax_at = [None]                      # the step from which a relaxed build rounds its exchanged blocks

    def relax_gate(worst, j_est, j_now):
        # A relaxed build (svd_topk: exchange='relaxed') rounds the exchanged blocks to fp32 once the pairs are within
        # 1e-7 / world of convergence: now, if the estimate just collected says so, else at the step where HALF the prior rate
        # puts that (a conservative extrapolation; should it be wrong the fp64 verification fails and the build goes on exact).
        if not getattr(lay, 'relaxed', False) or lay.exchange_dtype is not None:
            return
        gate = 1e-7 / max(comm.world, 1)
        at = j_est + (0 if worst <= gate else int(math.ceil(np.log(worst / gate) / (0.5 * prior_rate))))
        relax_at[0] = at if relax_at[0] is None else min(relax_at[0], at)

    def relax_now(j_now):
        if relax_at[0] is not None and j_now >= relax_at[0] and getattr(lay, 'relaxed', False) and lay.exchange_dtype is None:
            lay.exchange_dtype = torch.float32
            stats['exchange_relaxed_from' if (rec is None or halves) else 'products_rounded_from'] = j_now + 1

    def plan(j_now):
        """the step of the next look from the history of estimates; (step, final?)"""
        jl, worst = hist[-1]
        rate = prior_rate                  # until two estimates have been seen
        if len(hist) >= 2 and hist[-2][1] > hist[-1][1] > 0:
            rate = max(0.4, np.log(hist[-2][1] / hist[-1][1]) / (hist[-1][0] - hist[-2][0]))
        remaining = np.log(max(worst, est_tol) / est_tol) / rate
        step = min(qcap, max(j_now + 1, jl + max(1, int(math.ceil(remaining)))))
        return step, True

    try:
        while j < qcap:
            j += 1
            N = j * b
            if min(j + 1, qcap) * b > Qbuf.shape[1]:      # grow the basis and the projected matrix (rare: slow convergence)
                cap2 = min(qcap, max(j + 1, int(1.5 * cap) + 1))
                Qn_, Tn_ = ops.empty(lay.rows, cap2 * b), ops.zeros(cap2 * b, cap2 * b)
                Qn_[:, :cap * b] = Qbuf
                Tn_[:cap * b, :cap * b] = T
                Qbuf, T, cap = Qn_, Tn_, cap2
            Qall = Qbuf[:, :N]
            last = j == qcap
            relax_now(j - 1)
            if rec is not None:
                rounded = lay.exchange_dtype is not None
                if not halves:
                    rec.steps(Qbuf, T, S_buf, flags, j - 1, 1, last, rounded=rounded)
                else:
                    # users sharded, item side replicated: this rank's products, ONE sum over the ranks (the blocks travel in
                    # fp32 once a relaxed build has opened its gate), the same orthogonalisation on every rank
                    W = rec.products(Qbuf, j, rounded=False)
                    W = comm.allreduce(W.to(lay.exchange_dtype)).to(torch.float64) if rounded else comm.allreduce(W)
                    rec.orth(Qbuf, T, S_buf, flags, W.contiguous(), j, last, rounded=rounded)
                stats['gramian_steps'] += 1
                stats['spmm_cols'] += b
                S = S_buf
            else:
                Qj = Qbuf[:, N - b:N].contiguous()     # (a compact copy: the SpMM gathers rows of it — 512-byte rows 10 KB apart would spread the gathers over twenty times the pages)
                W = gop.apply(Qj)
                C = lay.gram(Qall, W)                                      # block column j of T, rows of all blocks so far
                T[:N, N - b:N] = C
                if lay.exchange_dtype is not None and getattr(lay, 'relaxed', False) and N > 2 * b:
                    # a ROUNDED product: C = Q^T (B Q_j + E_j) carries E_j (6e-8) in every block row.  In the rows of the early
                    # blocks that noise is harmless where it stands (it multiplies the small late coefficients y_j), but its
                    # MIRROR image would multiply the O(1) early coefficients: the rows of T below the band are Q_j^T W_i with
                    # W_i exact — zero to rounding under full reorthogonalisation — and are left so
                    T[N - b:N, N - 2 * b:N - b] = C[N - 2 * b:N - b].t()
                    T[:N - 2 * b, N - b:N] = 0.0          # (the look symmetrises its snapshot: the noise above the band goes too)
                else:
                    T[N - b:N, :N - b] = C[:N - b].t()
                if not last:
                    S = _next_lanczos_block(lay, W, Qbuf, N, C, flags)
                else:                              # the last block the space can hold: the coupling of W_perp directly
                    S = lay.gram(ops.tsmm_sub(W, Qall, C))
            # ---- a monitor that is due: collect it and plan the next look --------------------------------------
            if monitor is not None and (j >= monitor.j + LAG or last or j + 1 >= next_look):
                t_w = time.perf_counter()
                out = monitor.join()
                stats.setdefault('monitor_wait_ms', []).append(round(1e3 * (time.perf_counter() - t_w), 3))
                jm, monitor = monitor.j, None
                warm, warm_lam = out['basis'], out['lam_all']
                hist.append((jm, out['worst']))
                relax_gate(out['worst'], jm, j)
                if verbose and comm.rank == 0:
                    print('[svd] lanczos monitor of step %2d (seen at %2d)  worst rel.res (first %d) %.2e  nested so far: %d outer, %d products'
                          % (jm, j, k, out['worst'], inner['outer'], inner['steps']))
                next_look, look_is_final = plan(j)
                if not last and next_look > j + 2 * LAG:
                    # convergence is far: another monitor starts at once, warm from the one just collected (it will be back
                    # before the look is due); otherwise the next look is the one on the main stream, where convergence is expected
                    next_look, look_is_final = j, False
            if j < next_look and not last:
                continue
            if monitor is not None:            # (a look is due while a monitor is still out: cannot happen — joined above)
                continue
            inner['checks'] += 1
            if not (look_is_final or last):
                # ---- launch a monitor on the side stream and keep stepping (it reads the breakdown flags too) ---
                monitor = _Monitor(ops, j, snapshot(N), S.clone() if rec is not None else S, warm, k, b, est_tol, hist[-1][1] if hist else None,
                                   seed + 1000 * j, inner, flags=flags, width=l, lam0=warm_lam)
                next_look = qcap + 1           # decided when the monitor comes back
                continue
            # ---- the pairs of T_j on the main stream, and their verification ------------------------------------
            t_w = time.perf_counter()
            breakdown_check(j)
            out = _ritz_check(ops, snapshot(N), S, warm, k, b, est_tol, hist[-1][1] if hist else None, seed + 1000 * j, inner,
                              final=len(hist) >= 1, width=l, lam0=warm_lam)
            stats.setdefault('look_ms', []).append(round(1e3 * (time.perf_counter() - t_w), 3))    # includes draining the steps queued before it
            warm, warm_lam = out['basis'], out['lam_all']
            hist.append((j, out['worst']))
            if verbose and comm.rank == 0:
                print('[svd] lanczos step %2d  dim %4d  worst rel.res (first %d) %.2e  nested so far: %d outer, %d products, inner converged %s'
                      % (j, N, k, out['worst'], inner['outer'], inner['steps'], out['conv']))
            relax_gate(out['worst'], j, j)
            if out['worst'] <= est_tol and out['conv']:
                Vk = ops.tsmm(Qall, out['Yk'])
                rounded, lay.exchange_dtype = lay.exchange_dtype, (None if getattr(lay, 'relaxed', False) else lay.exchange_dtype)
                if rec is not None:                # one true product on the k Ritz vectors
                    Z = rec.gramian(Vk)
                    if halves:
                        Z = comm.allreduce(Z)
                    stats['gramian_steps'] += 1
                    stats['spmm_cols'] += k
                else:
                    Z = gop.apply(Vk)              # (a relaxed build exchanges THIS product in fp64: a pair is accepted on a true residual)
                lay.exchange_dtype = rounded
                lam1 = max(float(out['lam_all'][0]), 1e-300)
                res2 = lay.total(ops.resid_colnorm2(Z, Vk, out['lam_k']))
                res_true = np.sqrt(np.maximum(ops.to_host(res2), 0.0)) / lam1
                stats['verified_rel_residual'] = float(res_true.max())
                if float(res_true.max()) <= tol:
                    result = (Vk, out['lam_all'][:k].copy(), res_true * lam1)
                    break
                est_tol *= 0.1                 # the estimate was optimistic (orthogonality): ask for more, keep going
                if getattr(lay, 'relaxed', False) and lay.exchange_dtype is not None:
                    lay.relaxed, lay.exchange_dtype = False, None      # ... and with exact exchanges from here on
                    stats['exchange_relaxed_failed_at'] = j
            if last:
                break
            next_look, look_is_final = plan(j)     # from here on every look is on the main stream: convergence is near
    finally:
        if monitor is not None:                # never leave a worker behind (exceptions, early exits)
            try:
                monitor.join()
            except BaseException:
                pass
        if rec is not None:
            rec.collect_timings()
    stats['lanczos_steps'] = j
    stats['outer'] = len(hist)
    stats['krylov_dim'] = j * b
    stats['checks'] = hist
    if result is None:
        raise _LanczosBreakdown('not converged in %d blocks' % j)
    return result


KRYLOV_WIDTHS = (16, 32, 64, 128, 256)      # the widths the SpMM has instances for (csrc/spmm.hip: 4 / 8 / 16 / 32 / 64 lanes per gathered row)


def _lanczos_model(nnz, n_items, l, b, world=1):
    """(steps, seconds per step) of a block Lanczos build with Krylov blocks of b columns, nested width l — the cost model
    behind `choose_krylov_block` / `choose_method` (same constants in csrc/driver.hip::lanczos_model).  Measured on one
    MI355X (profiles/r06_krylov_block_*.txt):
      steps: 14 at b = l = 64 (one more per doubling of l), growing like (l / b)^(0.33 + 0.035 log2(l / b)) as the block narrows
             (ML-20M-shaped rank 50: 14 / 15 / 17 / 19 / 23-24 at 64 / 48 / 32 / 24 / 16; rank 100: 15 / 19 / 25-27 / 36-38 at
             128 / 64 / 32 / 16; S-1M: 15 / 19 / 24-25 at 64 / 32 / 16) — the model gives 14 / 18.0 / 24.4 and 15 / 19.3 / 26.1 / 37.0;
      both products of a step: nnz (8 + b) ps  (ML-20M-shaped: 1.22 / 0.78 / 0.50 ms at 64 / 32 / 16; S-1M: 9.0 / 4.6 / 2.5 ms) —
             the row pieces of a narrow block stay on chip, and every gather instruction carries 16 columns whatever b is;
      everything else of a step (the projections against the Krylov basis, CholeskyQR3 — inside the library): 0.3 ms
             plus the basis traffic of the re-orthogonalisation (14 passes over n_items x N x b flop at the fp64 rate)."""
    from .machine_model import value as mm
    ratio = max(float(l) / b, 1.0)
    steps = (14.0 + max(0.0, math.log2(l / 64.0))) * ratio ** (0.33 + 0.035 * math.log2(ratio))
    t_spmm = nnz * (8.0 + max(b, 16)) * 1e-12 / world
    if world > 1:
        t_spmm += 2.0 * (world - 1) / world * n_items * b * 8.0 / mm('xgmi_bus_Bps') + 6 * (world - 1) * mm('collective_step_s')
    n_avg = 0.5 * steps * b
    t_reorth = 14.0 * n_items * n_avg * b / mm('dense_f64_flops') / world
    return steps, t_spmm + mm('lanczos_step_fixed_s') + t_reorth


def choose_krylov_block(nnz, n_items, l, world=1):
    """Width of a Krylov block: the widest block is NOT the cheapest build.  A block of b < l columns needs (l / b)^(0.33 + 0.035 log2(l / b)) times
    the steps, but a step's sparse products shrink almost in proportion to b — the Krylov space reaches a given dimension
    with fewer gathered columns in total — so the width is chosen where steps x (products + the fixed cost of a step) is
    least (round 6; rounds 4-5 tied the block to the nested width l: ML-20M-shaped rank 50 33.3 -> 27.8 ms at b = 32, S-1M
    169 -> 80 ms at b = 16, rank 100 89.7 -> 60 ms)."""
    best, best_t = None, None
    for b in KRYLOV_WIDTHS:
        if b > l and best is not None:
            break
        b = min(b, l)
        steps, t_step = _lanczos_model(nnz, n_items, l, b, world)
        if best_t is None or steps * t_step < best_t:
            best, best_t = b, steps * t_step
    return int(best)


def choose_method(nnz, n_items, l, world=1):
    """'lanczos' or 'subspace' from the cost model of a build (the same rule in csrc/driver.hip::svd_build_impl).
    Block Lanczos (with its best block width, `choose_krylov_block`) against the filtered subspace iteration, which needs
    ~2.6x the Gramian steps of the widest Krylov block, every one of them l columns wide, plus ~8 outer iterations of
    Rayleigh-Ritz and CholeskyQR on the block (0.5 ms each):
      ML-20M-shaped rank 50 (l = 64):      subspace 47.7 ms, lanczos 27.8 ms (b = 32)
      ML-20M-shaped rank 100 (l = 128):    112 / 60 ms
      S-1M rank 50:                        289 / 80 ms (b = 16)
      ML-1M-shaped rank 10 (l = 24): a step is 0.05 ms, the looks and the fixed cost of 11 steps are not: 4.6 / 8.2 ms => subspace
    Sharded over `world` ranks the products shrink by the number of ranks and the exchange of the block joins every step of
    either method."""
    from .machine_model import value as mm
    if not math.isfinite(nnz):
        return 'lanczos'
    b = choose_krylov_block(nnz, n_items, l, world)
    steps, t_step = _lanczos_model(nnz, n_items, l, b, world)
    t_lanczos = steps * t_step + 4.0 * mm('nested_solve_s') * max(1.0, (l / 64.0) ** 2)
    _, t_wide = _lanczos_model(nnz, n_items, l, l, world)
    t_wide -= mm('lanczos_step_fixed_s')
    t_subspace = 2.6 * 14.0 * t_wide + 8 * 0.5e-3 * max(1.0, (l / 64.0) ** 2)
    return 'subspace' if t_subspace <= t_lanczos else 'lanczos'


def plan_build(ops, A, k, block=None, method=None, krylov_block=None, max_steps=None, comm=None):
    """The decisions of a build before its first product: nested width, method, width of a Krylov block, the modelled step
    time (for the monitors' lag), the step limit.  Every rank decides alike: the entry count is summed over the ranks;
    operators that do not say how many entries they hold (host-side LinearOperators: their products are the expensive
    kind) count as large and keep the full block width."""
    comm = comm or NoComm()
    n_items = A.shape[1]
    if not (0 < k <= n_items):
        raise ValueError('k must satisfy 0 < k <= n_items')
    l = int(block or default_block(k, n_items))
    l = max(k, min(l, n_items))
    method = method or DEFAULT_METHOD
    if method not in ('lanczos', 'subspace', 'auto'):
        raise ValueError("method must be 'lanczos', 'subspace' or 'auto'")
    nnz = getattr(A, 'nnz', None)
    total = float('inf')
    if nnz is not None:
        if comm.world > 1:
            t = ops.to_device(np.array([float(nnz)]))
            total = float(ops.to_host(comm.allreduce(t))[0])
        else:
            total = float(nnz)
    if method == 'auto':
        method = choose_method(total, n_items, l, comm.world)
    kb, t_step, steps_model = l, None, None
    if method == 'lanczos':
        if krylov_block is not None:
            kb = max(1, min(int(krylov_block), l))
        elif math.isfinite(total):
            kb = choose_krylov_block(total, n_items, l, comm.world)
        if math.isfinite(total):
            steps_model, t_step = _lanczos_model(total, n_items, l, kb, comm.world)
        if max_steps is None:
            max_steps = min(MAX_KRYLOV_COLS // kb, int(64 * math.sqrt(l / float(kb))))
    elif max_steps is None:
        max_steps = 64
    return dict(block=l, method=method, krylov_block=kb, t_step=t_step, steps_model=steps_model, max_steps=max_steps, nnz_total=total)


def _library_recurrence(ops, A, comm, sharded):
    """True when the steps of a Lanczos build of A run inside the library (ops.lanczos_recurrence): a device matrix and a
    REPLICATED item side — one process, or users sharded over the ranks with ONE all-reduce of W = A^T A Q_j per step between the
    two halves of the library's step (pk_lanczos_products / pk_lanczos_orth).  The row-sharded item layout (`ItemRows.sharded`:
    collectives inside the orthogonalisation), host-side operators and the CPU double take the composition."""
    return (not sharded and (not getattr(comm, '_always', False) or getattr(comm, 'split_step', False)) and hasattr(ops, 'lanczos_recurrence')
            and hasattr(A, 'indptr') and not hasattr(A, 'matvec'))


def prepare_operator(ops, A, k, comm=None, **kw):
    """Builds, ahead of `svd_topk`, the image of A its products will run on — the library's handle with its user-blocked
    transpose (one process) or the layer's own transposed operator — so that a caller can account for it separately
    (bench.py's `transpose_and_plans_s`); `svd_topk` finds it cached on the matrix.  Returns the plan of the build."""
    comm = comm or NoComm()
    plan = plan_build(ops, A, k, comm=comm, **kw)
    if plan['method'] == 'lanczos' and _library_recurrence(ops, A, comm, False):      # svd_topk's own default (item side replicated)
        ops.lanczos_recurrence(A, plan['krylov_block'])
    elif hasattr(A, 'transpose_operator'):
        A.transpose_operator()
    return plan


def svd_topk(ops, A, k, block=None, tol=1e-12, max_outer=200, m_max=24, spread=1e7, seed=0,
             comm=None, want_u=False, verbose=False, even_lock=True, shard_items=None, method=None, max_steps=None,
             exchange='auto', krylov_block=None, monitor_lag=None, exchange_overlap='auto', first_look=None, products='auto'):
    """Returns (U_local | None, sigma[k] desc, V [n_items x k], stats) as device tensors of `ops`.

    A: ops-level CSR of the LOCAL row shard (n_local x n_items).  Convergence: every one of the k
    leading Ritz pairs has ||B x - theta x|| <= tol * theta_1  (B = A^T A, theta = sigma^2), measured on a true product.
    method: 'lanczos' = block Lanczos with full reorthogonalisation and Rayleigh-Ritz over the whole Krylov space
    (`_block_lanczos`), falling back to 'subspace' = Chebyshev-filtered subspace iteration with locking
    (`_subspace_iteration`) when the recurrence breaks down (rank-deficient matrices, tiny item counts); None / 'auto' =
    `choose_method` (a cost model), unless `solver.DEFAULT_METHOD` names one (tests).
    krylov_block: width of a Krylov block of the Lanczos build (None: `choose_krylov_block`; the nested solves and the
    subspace iteration keep the width `block`).  monitor_lag: steps between the launch of a monitor and its collection
    (None: from the modelled step time; 0: every look on the calling thread).  products: 'f64', or 'relaxed' = the sparse
    products of the LATE steps of a one-process Lanczos build gather fp32 images of their dense blocks (same gate, fp64
    verification and fall-back as exchange='relaxed').  first_look: the step of the first look (None:
    half the modelled number of steps).  max_steps: blocks of the Krylov space at
    most (None: 64 at the full width, more for narrow blocks, never beyond the 4096 columns of the Gram kernels).
    """
    comm = comm or NoComm()
    n_items = A.shape[1]
    plan = plan_build(ops, A, k, block=block, method=method, krylov_block=krylov_block, max_steps=max_steps, comm=comm)
    l, method, kb, t_step, max_steps = plan['block'], plan['method'], plan['krylov_block'], plan['t_step'], plan['max_steps']
    stats_method = method
    # the operator of Z = A^T Y: a device matrix offers its user-blocked transpose (ops.BlockedTranspose) — built at the first
    # product that needs it (`_Gramian.At`)
    At = _Lazy((lambda: A.transpose_operator()) if hasattr(A, 'transpose_operator') else (lambda: A.T))

    # Payload of the block exchanges on more than one rank (never silently):
    #   'f64'      the blocks as they are — what 'auto' means;
    #   'relaxed'  fp32 on the wire for the LATE steps of a Lanczos build only: the true residual of a Ritz pair is the estimated
    #              one plus  sum_j E_j y_j  (E_j: what the rounding did to the product of step j, 6e-8 of its norm; y_j: block j
    #              of the pair's coefficients), and the coefficients of a converging pair in the blocks added late are as small
    #              as its residual was when they were added — so a product may be rounded once a collected estimate says the
    #              pairs are within 1e-7 / world (its error then enters at 6e-15), never earlier; the verification product is
    #              exchanged in fp64 (a pair is accepted on a TRUE residual), and a failed verification ends the rounding for
    #              the rest of the build.  (The relaxation theory of inexact Krylov methods: early products exact, late ones
    #              loose — the opposite of "fp32 until the end", which stalled at 5e-12 in round 4)
    #   'f32'      every exchange rounded (an explicit choice for builds to a tolerance >= 1e-6; warns below it).
    if products not in ('auto', 'f64', 'relaxed'):
        raise ValueError("products must be 'auto', 'f64' or 'relaxed'")
    if exchange not in ('auto', 'f64', 'f32', 'relaxed'):
        raise ValueError("exchange must be 'auto', 'f64', 'relaxed' or 'f32'")
    if exchange == 'f32' and tol < 1e-6:
        import warnings
        warnings.warn('svd_topk: exchange=\'f32\' rounds every exchanged product to 6e-8 of its norm; a tolerance of %.1e is out of '
                      'its reach (use \'relaxed\' or \'f64\')' % tol, RuntimeWarning, stacklevel=2)
    xdt = torch.float32 if exchange == 'f32' else None
    if getattr(comm, 'split_step', False) and shard_items is None:
        shard_items = False                 # (TorchComm's test switch: the two halves of the library's step in a group of one)
    if shard_items is None:
        # the item side of a sharded build: REPLICATED where the library runs the step (one all-reduce of a 16-column block per
        # step, the orthogonalisation — 0.3 ms — repeated on every rank), row-sharded (`ItemRows`) for the composition, whose
        # per-rank dense work it divides by the number of ranks
        shard_items = not (method == 'lanczos' and _library_recurrence(ops, A, comm, False))
    lay = ItemRows(ops, comm, n_items, shard_items, exchange_dtype=xdt, overlap=exchange_overlap)
    lay.relaxed = exchange == 'relaxed'
    stats = dict(krylov_block=kb if method == 'lanczos' else None, exchange={'auto': 'f64'}.get(exchange, exchange), outer=0, gramian_steps=0, spmm_cols=0, degrees=[], locked_at=[], block=l, converged=False,
                 item_rows_per_rank=lay.rows, items_sharded=lay.sharded, method=stats_method)
    Vk = lam_k = res_k = None
    if method == 'lanczos':
        try:
            # `max_outer` bounds the work of either method: an outer iteration of the subspace method is worth a few blocks
            Vk, lam_k, res_k = _block_lanczos(ops, A, At, lay, comm, k, l, tol, seed, stats, verbose,
                                              min(max_steps, 4 * max_outer), m_max, spread, even_lock, kb=kb,
                                              monitor_lag=monitor_lag, t_step=t_step, steps_model=plan['steps_model'],
                                              first_look=first_look, products={'auto': DEFAULT_PRODUCTS}.get(products, products))
            stats['converged'] = True
        except _LanczosBreakdown as exc:
            stats['lanczos_fallback'] = str(exc)
            stats['method'] = 'subspace (after a Lanczos breakdown)'
            if verbose and comm.rank == 0:
                print('[svd] block Lanczos gave up (%s): filtered subspace iteration' % exc)
    if Vk is None:
        X = orthonormalize(lay, lay.randn(l, seed))
        gop = _Gramian(ops, A, At, lay, comm, stats)
        basis, lam_all, res_act, n_lock, conv = _subspace_iteration(
            gop, lay, k, X, tol, max_outer, m_max, spread, seed, stats, verbose=verbose, even_lock=even_lock,
            rank0=comm.rank == 0)
        stats['converged'] = bool(conv)
        take = min(k, basis.shape[1])
        Vk = basis[:, :take]
        lam_k = lam_all[:take]
        # residuals of the k leading pairs: locked pairs are below tol by construction
        res_k = res_act[:max(1, min(k - n_lock, len(res_act)))]

    lay.exchange_dtype = None                               # the factors themselves always travel as they are
    Vk = lay.full(Vk[:, :k].contiguous()).contiguous()      # the factors are replicated: scoring needs every item row
    lam_k = np.maximum(np.asarray(lam_k, dtype=np.float64)[:k], 0.0)
    order = np.argsort(-lam_k, kind='stable')
    if not np.array_equal(order, np.arange(len(order))):
        Vk = Vk[:, torch.as_tensor(order, device=Vk.device)].contiguous()
        lam_k = lam_k[order]
    sigma_k = np.sqrt(lam_k)
    # residual of the worst of the k leading pairs relative to theta_1
    stats['final_rel_residual'] = float(np.max(res_k) / max(lam_k[0], 1e-300))
    stats['tol'] = tol
    U = None
    if want_u:
        U = ops.spmm(A, Vk)
        inv = ops.to_device(np.where(sigma_k > 0, 1.0 / np.maximum(sigma_k, 1e-300), 0.0))
        U = ops.scale_cols(U, inv)
    return U, ops.to_device(sigma_k), Vk, stats
