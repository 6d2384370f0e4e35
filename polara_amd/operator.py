"""`SVDModel.build(operator=...)` (models.py:835-844): the reference hands `svds` an operator instead of the
training matrix — HybridSVD passes  L_K^T A L_S  either multiplied out as one sparse matrix
(`precompute_auxiliary_matrix`, hybrid/models.py:357-363) or as a SciPy LinearOperator chaining the three
products (hybrid/models.py:364-381).  Three forms are accepted here:

  * a SciPy sparse matrix              -> a DeviceCSR: the whole build runs on the device like a plain one;
  * `SparseProduct(F_0, ..., F_m)`     -> the factors live on the device as CSR matrices and the operator is
    the chain of their SpMMs (the device form of HybridSVD's matvec/rmatvec closures: pass the sparse
    Cholesky factors themselves instead of wrapping them in host closures);
  * any other LinearOperator           -> host code by definition: its products run on the host, the block
    eigensolver around them (Gram matrices, Jacobi eigh, tall-skinny GEMMs, Chebyshev recurrence) on the device.

What the solver asks of a matrix is `.shape`, `.T` and a product through `ops.spmm`, which hands anything
with an `apply` method its operand: the two classes below provide exactly that.
"""
import numpy as np


class SparseProduct:
    """The product F_0 @ F_1 @ ... @ F_m of SciPy sparse matrices, kept factored (never multiplied out).
    A marker for `build(operator=...)`; `DeviceChain` is its device-resident form."""

    def __init__(self, *factors):
        if not factors:
            raise ValueError('SparseProduct needs at least one factor')
        for a, b in zip(factors[:-1], factors[1:]):
            if a.shape[1] != b.shape[0]:
                raise ValueError('factor shapes do not chain: %s @ %s' % (a.shape, b.shape))
        self.factors = factors
        self.shape = (int(factors[0].shape[0]), int(factors[-1].shape[1]))


class DeviceChain:
    """M = F_0 F_1 ... F_m with every factor a DeviceCSR; a product with a dense block is m+1 SpMMs,
    right to left, the transpose walks the transposed factors the other way."""

    def __init__(self, ops, factors):
        self.ops = ops
        self.factors = list(factors)
        self.shape = (self.factors[0].shape[0], self.factors[-1].shape[1])
        self.nnz = sum(f.nnz for f in self.factors)
        self._T = None

    @classmethod
    def from_scipy(cls, ops, product, col_perm=None):
        """product: SparseProduct.  col_perm: item_rank (external id -> internal position), applied to the
        columns of the LAST factor (a device-side renaming)."""
        devs = []
        for f in product.factors:
            f = f.tocsr()
            if not f.has_canonical_format:
                f = f.copy()
                f.sum_duplicates()
            devs.append(ops.csr(f.indptr, f.indices, f.data, f.shape))
        if col_perm is not None:
            devs[-1] = ops.csr_relabel_cols(devs[-1], col_perm)
        return cls(ops, devs)

    @property
    def T(self):
        if self._T is None:
            self._T = DeviceChain(self.ops, [f.T for f in reversed(self.factors)])
            self._T._T = self
        return self._T

    def apply(self, X, out=None):
        for f in reversed(self.factors[1:]):
            X = self.ops.spmm(f, X)
        return self.ops.spmm(self.factors[0], X, out)


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
