"""Test-only stand-in for the `numba` package (NOT product code, never shipped on a path the
product imports).

Purpose: `/root/reference/polara/lib/*.py` does `from numba import jit, njit, guvectorize, prange`
at import time and numba is not installed in this image (no network).  This shim lets
`tests/golden/make_golden.py` import the *unmodified* reference in this container so that golden
vectors can be generated from it.  The decorators are semantic no-ops: a numba-jitted function has,
by numba's contract, the semantics of the same Python source, so running the source un-jitted gives
the same results (only slower).  `guvectorize` is emulated with `numpy.vectorize(signature=...)`
for the output-argument calling convention the reference uses.
"""
import numpy as _np

float64 = _np.float64
float32 = _np.float32
intp = _np.intp
int32 = _np.int32
int64 = _np.int64
prange = range


def _passthrough(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(func):
        return func
    return deco


jit = _passthrough
njit = _passthrough


def guvectorize(ftylist, signature, **kwargs):
    """Emulates numba.guvectorize for kernels written as `f(in..., out)` with one output."""
    ins, outs = signature.split('->')
    n_out = outs.count('(')
    assert n_out == 1, 'shim supports a single gufunc output'
    in_sigs = [s for s in ins.replace(' ', '').split('),') if s]
    in_sigs = [s if s.endswith(')') else s + ')' for s in in_sigs]
    out_sig = outs.strip()
    out_dims = [d for d in out_sig.strip('()').split(',') if d]

    def deco(func):
        def call(*arrays):
            arrays = [_np.asarray(a) for a in arrays]
            # resolve symbolic core dims from the inputs
            dims = {}
            for sig, arr in zip(in_sigs, arrays):
                names = [d for d in sig.strip('()').split(',') if d]
                if names:
                    for name, size in zip(names, arr.shape[arr.ndim - len(names):]):
                        dims[name] = size
            out_core = tuple(dims[d] for d in out_dims)

            def core(*core_args):
                res = _np.zeros(out_core, dtype=_np.float64) if out_core else _np.zeros(1)
                # scalars arrive as 0-d; the reference indexes them as x[0]
                prepared = [_np.atleast_1d(a) for a in core_args]
                func(*prepared, res)
                return res if out_core else res[0]

            vec = _np.vectorize(core, signature=signature)
            return vec(*arrays)
        return call
    return deco
