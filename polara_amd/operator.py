"""`SVDModel.build(operator=...)` (models.py:835-844): the reference hands `svds` any SciPy LinearOperator
instead of the training matrix (HybridSVD's  L_K^T A L_S, hybrid/models.py:352-386).  A LinearOperator is
host code by definition, so its products run on the host; the block eigensolver around it (Gram matrices,
Jacobi eigh, tall-skinny GEMMs, Chebyshev recurrence) still runs on the device.  The object below gives the
operator the three things the solver asks of a matrix: `.shape`, `.T`, and a product through `ops.spmm`.
"""
import numpy as np


class HostOperator:
    def __init__(self, ops, op, col_perm=None, _transposed=False):
        """op: scipy.sparse.linalg.LinearOperator-like (matmat / rmatmat, shape) in EXTERNAL item order.
        col_perm: item_rank (external id -> internal row) when the solver works in a relabelled item order."""
        self.ops = ops
        self.op = op
        self.col_perm = None if col_perm is None else np.asarray(col_perm, dtype=np.int64)
        self._inv = None if col_perm is None else np.argsort(self.col_perm)
        self._transposed = _transposed
        m, n = op.shape
        self.shape = (int(n), int(m)) if _transposed else (int(m), int(n))
        self.nnz = None

    @property
    def T(self):
        t = HostOperator.__new__(HostOperator)
        t.__dict__.update(self.__dict__)
        t._transposed = not self._transposed
        t.shape = (self.shape[1], self.shape[0])
        return t

    def apply(self, X, out=None):
        """device [n_cols x nc] -> device [n_rows x nc] through the host operator"""
        ops = self.ops
        Xh = np.ascontiguousarray(ops.to_host(X), dtype=np.float64)
        if not self._transposed:
            if self.col_perm is not None:
                Xh = Xh[self.col_perm]               # internal rows -> external item order
            R = self.op.matmat(Xh)
        else:
            R = self.op.rmatmat(Xh)
            if self.col_perm is not None:
                R = R[self._inv]                     # external item order -> internal rows
        R = ops.to_device(np.ascontiguousarray(np.real(R), dtype=np.float64))
        if out is not None:
            out.copy_(R)
            return out
        return R
