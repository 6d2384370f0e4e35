"""-m gpu -m probe: the equality tests of the kernel variants that left the product library in round 5 (csrc/experiments/:
the LDS-staged sweeps, the two-groups-per-wave sweep, the LDS-head fold-in).  They run against a KERNEL-TUNING build only:

    python tools/build_probe_lib.py /tmp/libpk_probe.so
    POLARA_HIP_LIB=/tmp/libpk_probe.so python -m pytest tests/test_gpu_probe_variants.py -m gpu

and are skipped (loudly) when the loaded library is the product one — whose sources no longer contain these kernels."""
import os

import numpy as np
import pytest
import scipy.sparse as sps
import torch

from numpy_ops import NumpyOps
from oracle import polara_oracle as orc
from test_gpu_kernels import rand_csr, brute_topk      # noqa: F401

pytestmark = [pytest.mark.gpu, pytest.mark.probe,
              pytest.mark.skipif(not os.environ.get('POLARA_HIP_LIB'), reason='experiment kernels: needs a probe build (tools/build_probe_lib.py) selected with POLARA_HIP_LIB')]


def _spmm_head(hip_ops, A, X, out):
    """the LDS-head fold-in instance of the experiment tree: x_kind PK_VAL_F32 | 16 through pk_spmm_csr_ex"""
    from polara_amd import _lib
    from polara_amd.ops import _ptr
    p = A.plan
    _lib.check(hip_ops.lib.pk_spmm_csr_ex(
        hip_ops.stream(), A.n_tasks, _ptr(p['task_row']), _ptr(p['task_begin']), _ptr(p['task_end']), _ptr(p['task_slot']), A.n_long,
        _ptr(p['long_row']), _ptr(p['long_slot_begin']), _ptr(p['long_slot_end']), _ptr(A.indices), _ptr(A.values), A.val_kind,
        _ptr(X), _lib.PK_VAL_F32 | 16, X.stride(0), X.shape[1], _ptr(out), out.stride(0), _ptr(A.partial(X.shape[1])), 0, 0,
        int(X.shape[0])), 'pk_spmm_csr_ex')
    return out


def _needs_spmm_tree(hip_ops):
    """The LDS-head instance lives in csrc/experiments/spmm_variants.hip, frozen BEFORE the list-driven product entered
    spmm.hip: a probe library carries it only when built with `--trees score,spmm` against a driver of that time (commit
    9e4a9e5 and earlier).  The default probe build (`--trees score`) keeps the product's spmm.hip, which rejects x_kind | 16."""
    import torch
    from polara_amd._lib import PolaraHipError
    A = hip_ops.csr(np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int32), np.ones(1, dtype=np.float32), (1, 4))
    try:
        _spmm_head(hip_ops, A, torch.zeros(4, 2, dtype=torch.float32, device=hip_ops.device),
                   torch.zeros(1, 2, dtype=torch.float64, device=hip_ops.device))
    except PolaraHipError:
        pytest.skip('this probe library was built without the spmm experiment tree (tools/build_probe_lib.py --trees score,spmm)')


@pytest.mark.parametrize('n_cols,nc,ld', [(3000, 52, 64), (300, 52, 64), (20000, 64, 64), (1500, 12, 16)])
def test_fold_in_with_the_head_of_the_image_in_lds(hip_ops, n_cols, nc, ld):
    """The persistent fold-in instance (fold_in_head_kernel: the first rows of the fp32 factor image staged in LDS; opt-in —
    it measured slower than the plain kernel, csrc/spmm.hip): against SciPy, and BIT-identical to the plain groups kernel
    (same mapping, same summation order) — over catalogues shorter than the LDS window (everything is head), long rows
    (split tasks + fix-up), empty rows, strided image and strided output."""
    import torch
    _needs_spmm_tree(hip_ops)
    rng = np.random.RandomState(n_cols + nc)
    n_rows = 20000
    pop = 1.0 / (1.0 + np.arange(n_cols)) ** 0.8            # popular items first: most entries fall into the head
    pop /= pop.sum()
    counts = rng.poisson(25, n_rows).clip(0, n_cols)
    counts[[5, 777]] = 0
    counts[[9, 4000]] = min(n_cols, 2500)                    # rows longer than a task (1024 entries)
    indptr = np.r_[0, np.cumsum(counts)].astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, c, replace=False, p=pop)) for c in counts]).astype(np.int32)
    values = rng.randint(1, 11, indptr[-1]).astype(np.float32) * 0.5
    A = hip_ops.csr(indptr, indices, values, (n_rows, n_cols))
    img = torch.zeros(n_cols, ld, dtype=torch.float32, device=hip_ops.device)[:, :nc]      # strided like FactorImage.V32x
    X32 = rng.randn(n_cols, nc).astype(np.float32)
    img.copy_(hip_ops.to_device(X32))
    out = torch.zeros(n_rows, nc + 4, dtype=torch.float64, device=hip_ops.device)
    _spmm_head(hip_ops, A, img, out[:, :nc])  # one launch: the LDS-head instance
    got = hip_ops.to_host(out)
    ref = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols)) @ X32.astype(np.float64)
    assert np.allclose(got[:, :nc], ref, rtol=1e-13, atol=1e-13) and np.abs(got[:, nc:]).sum() == 0.0
    parts = torch.zeros(n_rows, nc, dtype=torch.float64, device=hip_ops.device)
    for lo in range(0, n_rows, 4000):                       # the plain kernel, in user batches
        hip_ops.spmm(A, img, out=parts, rows=(lo, min(n_rows, lo + 4000)))
    assert np.array_equal(hip_ops.to_host(parts), got[:, :nc])
    again = torch.zeros_like(out)
    _spmm_head(hip_ops, A, img, again[:, :nc])
    assert np.array_equal(hip_ops.to_host(again), got)

@pytest.mark.parametrize('cfg', [dict(n_users=3000, n_items=9000, K=50, topk=10, chunk=0),
                                 dict(n_users=1100, n_items=5000, K=100, topk=10, chunk=7),
                                 dict(n_users=70, n_items=2600, K=24, topk=5, chunk=3)])
def test_lds_shared_tile_sweep_equals_the_register_fed_sweep(hip_ops, cfg, monkeypatch):
    """The SHARED instance of the candidate sweep (PK_SCORE_SHARED=1: sixteen waves per workgroup step through the item
    tiles together, the packed V tile staged once per workgroup in LDS by global_load_lds, three buffers, one barrier per
    tile): ids and scores equal to the barrier-free kernel's — pruned (waves idle until their workgroup is done) and full
    sweeps, tiny item chunks (state parked and resumed under the barrier scheme, groups that finished in an earlier launch),
    a last workgroup with waves that own no users, with and without the threshold bootstrap."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items + 1)
    decay = (1.0 + np.arange(n_items)) ** -0.6
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2), (40, n_items - 3)], empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    monkeypatch.setenv('PK_SCORE_HEAD_TILES', '0')
    out = {}
    try:
        hip_ops.score_tiles_per_chunk = cfg['chunk']
        hip_ops.score_splits_override = 1
        for shared in ('0', '1'):
            monkeypatch.setenv('PK_SCORE_SHARED', shared)
            res = []
            for boot in ('16', '0'):
                monkeypatch.setenv('PK_SCORE_BOOT_TILES', boot)
                st = {}
                res += [scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st),
                        scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, prune=False),
                        scoring.recommend(hip_ops, F, T, topk, False, return_scores=True)]
                res.append(st['tiles_scored'])
            out[shared] = res
    finally:
        hip_ops.score_tiles_per_chunk = 0
        hip_ops.score_splits_override = 0
    for a, b in zip(out['0'], out['1']):
        if isinstance(a, tuple):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        else:
            assert a == b          # the same tiles scored: a group's exit does not depend on its workgroup

@pytest.mark.parametrize('cfg', [dict(n_users=3000, n_items=9000, K=50, topk=10, chunk=0),
                                 dict(n_users=1130, n_items=5000, K=100, topk=10, chunk=7),
                                 dict(n_users=97, n_items=2600, K=24, topk=5, chunk=3)])
def test_two_user_groups_per_wave_sweep_equals_the_single_group_sweep(hip_ops, cfg, monkeypatch):
    """score_candidates_pair_kernel (PK_SCORE_PAIR=1, round 4: every loaded V fragment serves TWO groups of 32 users — six
    MFMAs per k-step on two accumulator chains, selection state per group; opt-in, it measured slower): ids and scores equal
    to the one-group-per-wave kernel's — pruned and full sweeps, tiny item chunks (both groups' state parked and resumed,
    one group of a wave pruned launches before the other), a last wave that owns only one group (odd group counts), with and
    without the threshold bootstrap, with and without seen-item filtering — and the same tiles scored."""
    from polara_amd import scoring
    n_users, n_items, K, topk = cfg['n_users'], cfg['n_items'], cfg['K'], cfg['topk']
    rng = np.random.RandomState(n_items + 2)
    decay = (1.0 + np.arange(n_items)) ** -0.6
    V = rng.randn(n_items, K) / np.sqrt(K) * decay[:, None]
    indptr, indices, values = rand_csr(rng, n_users, n_items, 40, long_rows=[(2, n_items // 2), (40, n_items - 3)], empty_rows=[7])
    T = hip_ops.csr(indptr, indices, values, (n_users, n_items))
    F = scoring.FactorImage(hip_ops, hip_ops.to_device(V))
    monkeypatch.setenv('PK_SCORE_HEAD_TILES', '0')
    out = {}
    try:
        hip_ops.score_tiles_per_chunk = cfg['chunk']
        hip_ops.score_splits_override = 1
        for pair in ('0', '1'):
            monkeypatch.setenv('PK_SCORE_PAIR', pair)
            res = []
            for boot in ('16', '0'):
                monkeypatch.setenv('PK_SCORE_BOOT_TILES', boot)
                st = {}
                res += [scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, stats=st),
                        scoring.recommend(hip_ops, F, T, topk, True, return_scores=True, prune=False),
                        scoring.recommend(hip_ops, F, T, topk, False, return_scores=True)]
                res.append(st['tiles_scored'])
            out[pair] = res
    finally:
        hip_ops.score_tiles_per_chunk = 0
        hip_ops.score_splits_override = 0
    for a, b in zip(out['0'], out['1']):
        if isinstance(a, tuple):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        else:
            assert a == b          # the same tiles scored: a group's exit does not depend on the wave it shares
