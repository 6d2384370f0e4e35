"""Seeded sweep of small random tensors through tucker.hooi against the oracle's restatement of lib/tensor.py
(same start block, same stopping rule): core-norm traces, projectors, core.  Runs on the NumPy double of the
ops here and on the HIP backend under -m gpu."""
import numpy as np
import pytest

from oracle import polara_oracle as orc
from polara_amd import tucker


def _configs():
    rng = np.random.RandomState(4242)
    out = []
    for i in range(10):
        shape = (int(rng.choice([12, 40, 90])), int(rng.choice([10, 25, 60])), int(rng.choice([3, 5, 8])))
        # the reference's svds calls need r_m < min(n_m, product of the other two ranks) (lib/tensor.py:107-119)
        while True:
            ranks = (int(rng.randint(2, min(shape[0], 8))), int(rng.randint(2, min(shape[1], 7))), int(rng.randint(2, shape[2])))
            if all(ranks[m] < np.prod(ranks) // ranks[m] for m in range(3)):
                break
        out.append(dict(seed=i, shape=shape, ranks=ranks, nnz=int(rng.choice([60, 300, 1500])), binary=bool(rng.rand() < 0.6)))
    return out


def _tensor(cfg):
    rng = np.random.RandomState(100 + cfg['seed'])
    n0, n1, n2 = cfg['shape']
    flat = rng.choice(n0 * n1 * n2, size=min(cfg['nnz'], n0 * n1 * n2 // 2), replace=False)
    idx = np.stack(np.unravel_index(flat, cfg['shape']), axis=1).astype(np.int64)
    val = np.ones(len(idx)) if cfg['binary'] else rng.randint(1, 5, len(idx)).astype(np.float64)
    return idx, val


def _check(ops, cfg):
    idx, val = _tensor(cfg)
    shape, ranks = cfg['shape'], cfg['ranks']
    trace_ref = []
    o0, o1, o2, og = orc.hooi(idx, val, shape, ranks, growth_tol=1e-4, num_iters=15, seed=cfg['seed'], trace=trace_ref)
    u0, u1, u2, core, trace = tucker.hooi(ops, idx, val, shape, ranks, growth_tol=1e-4, num_iters=15, seed=cfg['seed'])
    u0, u1, u2, core = (ops.to_host(t) for t in (u0, u1, u2, core))
    # ARPACK (the oracle's svds) and the Gram+Jacobi route agree on well-separated spectra; a random tensor can
    # have near-degenerate trailing singular values in an unfolding, where the subspaces are defined only up to
    # that gap: the fit (core norm) is the robust invariant, the projectors are compared when the traces agree
    n = min(len(trace), len(trace_ref))
    assert n >= 1 and np.allclose(trace[:n], trace_ref[:n], rtol=1e-6), (trace, trace_ref)
    for u in (u0, u1, u2):
        assert np.abs(u.T @ u - np.eye(u.shape[1])).max() < 1e-9
    dense = np.zeros(shape)
    np.add.at(dense, (idx[:, 0], idx[:, 1], idx[:, 2]), val)
    want = np.einsum('uif,ua,ib,fc->abc', dense, u0, u1, u2)
    assert np.allclose(core, want, atol=1e-9 * max(np.abs(want).max(), 1e-300))
    if len(trace) == len(trace_ref) and np.allclose(trace, trace_ref, rtol=1e-9):
        assert np.isclose(np.linalg.norm(core), np.linalg.norm(og), rtol=1e-8)


@pytest.mark.parametrize('cfg', _configs(), ids=lambda c: 's%d_%s_%s' % (c['seed'], 'x'.join(map(str, c['shape'])), 'x'.join(map(str, c['ranks']))))
def test_random_hooi_cpu_double(cfg):
    from numpy_ops import NumpyOps
    _check(NumpyOps(), cfg)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', _configs(), ids=lambda c: 's%d_%s_%s' % (c['seed'], 'x'.join(map(str, c['shape'])), 'x'.join(map(str, c['ranks']))))
def test_random_hooi_gpu(hip_ops, cfg):
    _check(hip_ops, cfg)


def test_factored_mode_products_cpu_double():
    """the factored mode products of tucker.hooi against the reference's loop, through the NumPy double of the device
    operators (the same host code drives the HIP kernels in tests/test_gpu_kernels.py)"""
    from numpy_ops import NumpyOps
    import importlib
    chk = importlib.import_module('test_gpu_kernels').check_factored_products
    for ranks in ((6, 5, 3), (4, 3, 5)):
        for weighted in (False, True):
            chk(NumpyOps(), ranks, weighted)


def test_device_coordinate_build_equals_the_protocol_build():
    """CoffeeModel.build on our own ArrayData sends the three coordinate columns up as they lie (levels, item counts and
    the item renaming on the device, tucker.device_coordinates); a data object that overrides `to_coo` goes through the
    reference's protocol (stacked [nnz x 3] index, data.py:794-817).  Same factors either way, bit for bit."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import CoffeeModel

    class ProtocolData(ArrayData):
        def to_coo(self, *a, **k):
            return ArrayData.to_coo(self, *a, **k)

    rs = np.random.RandomState(5)
    n_users, n_items, n = 300, 90, 6000
    u, i = rs.randint(0, n_users, n), rs.randint(0, n_items, n)
    f = rs.choice([0.5, 1.0, 2.5, 4.0, 5.0], n)
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    built = []
    for cls in (ArrayData, ProtocolData):
        m = CoffeeModel(cls((u, i, f), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False), ops=NumpyOps())
        m.verbose = False
        m.mlrank, m.seed = (7, 6, 3), 1
        m.build()
        built.append(m)
    a, b = built
    assert a.core_norm_trace == b.core_norm_trace
    for k in a.factors:
        assert np.array_equal(a.factors[k], b.factors[k]), k


def test_device_coordinate_build_rejects_feedback_outside_the_level_set():
    """with a FIXED level set (ShardedArrayData: the manifest's levels, so that every rank uses the same ones)
    a shard value that is not a level used to alias silently — above the last level into (user + 1, level 0), between two
    levels into the upper one.  The device check rides in the one host read of the item counts and raises like the
    reference's test_to_coo does ('Not all values of feedback are present')."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import CoffeeModel

    class FixedLevels(ArrayData):
        def _levels(self):
            return np.array([1.0, 2.0, 3.0])

    rs = np.random.RandomState(6)
    n_users, n_items, n = 120, 40, 2500
    u, i = rs.randint(0, n_users, n), rs.randint(0, n_items, n)
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    for f, ok in ((rs.choice([1.0, 2.0, 3.0], n), True), (rs.choice([1.0, 2.0, 3.0, 4.0], n), False),
                  (rs.choice([1.0, 2.5, 3.0], n), False)):
        m = CoffeeModel(FixedLevels((u, i, f), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False), ops=NumpyOps())
        m.verbose = False
        m.mlrank, m.seed = (5, 4, 2), 1
        if ok:
            m.build()
        else:
            with pytest.raises(ValueError, match='Not all values of feedback'):
                m.build()


@pytest.mark.gpu
def test_hooi_redoes_an_iteration_when_a_direct_eigensolve_reports_failure(hip_ops, monkeypatch):
    """tucker.hooi reads the verdicts of the direct eigensolves (csrc/eigh_top.hip) once per iteration, with the core
    norm.  A verdict of 0 must discard the iteration and run it — and every later one — on the Jacobi route, from the
    same inputs: the result is then, bit for bit, that of a build that never used the direct kernel; and the direct
    route itself agrees with the Jacobi route to rounding."""
    import torch
    from polara_amd import tucker
    rs = np.random.RandomState(9)
    shape, ranks, nnz = (400, 260, 5), (20, 18, 4), 30000
    idx = np.stack([rs.randint(0, s, nnz) for s in shape], 1).astype(np.int64)

    def run():
        u0, u1, u2, core, trace = tucker.hooi(hip_ops, idx, None, shape, ranks, num_iters=6, growth_tol=1e-9, seed=3)
        return [hip_ops.to_host(x) for x in (u0, u1, u2, core)], trace

    direct, trace_direct = run()
    calls = []
    real = type(hip_ops).eigh_top_deferred

    def failing_once(self, S, r):
        lam, C, verdict = real(self, S, r)
        calls.append(1)
        if len(calls) == 3:                              # the first solve of the second iteration "fails"
            verdict = torch.zeros_like(verdict)
        return lam, C, verdict
    monkeypatch.setattr(type(hip_ops), 'eigh_top_deferred', failing_once)
    retried, trace_retried = run()
    monkeypatch.setattr(type(hip_ops), 'eigh_top_deferred', real)
    assert len(calls) >= 4 and len(trace_retried) == len(trace_direct)          # the direct route ran, then stopped being used
    n_direct_calls = len(calls)
    # a build that is on the Jacobi route from its second iteration on: the first iteration direct, like the retried run
    calls.clear()

    def first_iteration_only(self, S, r):
        calls.append(1)
        lam, C, verdict = real(self, S, r)
        return (lam, C, verdict) if len(calls) <= 2 else (lam, C, torch.zeros_like(verdict))
    monkeypatch.setattr(type(hip_ops), 'eigh_top_deferred', first_iteration_only)
    jacobi_after_one, trace_j = run()
    assert n_direct_calls == 4 and len(calls) == 4          # two per iteration: the failed iteration finished its direct solves before the verdicts were read
    for a, b in zip(retried, jacobi_after_one):
        assert np.array_equal(a, b)
    assert trace_retried == trace_j
    # and the two routes agree to rounding (subspaces: the factors' projectors)
    assert np.allclose(trace_direct, trace_retried, rtol=1e-10)
    for a, b in zip(direct[:3], retried[:3]):
        assert abs(a @ a.T - b @ b.T).max() < 1e-8
