"""-m gpu: the ingest kernels (csrc/ingest.hip) through the C ABI against NumPy/SciPy: stable radix sort, exclusive
scan, COO -> CSR with duplicate sums, CSR -> CSC (plain and user-blocked), sorted column renaming, per-item counts, the
device-built wave-task plan, and the user-blocked transposed product.  Index results must be bit-exact."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sps
import torch

from polara_amd import _lib
from polara_amd.csr import coo_to_csr, csr_transpose, build_row_tasks
from polara_amd.ops import _ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [0, 1, 63, 4096, 4097, 100003, 3_000_000])
def test_exclusive_scan(hip_ops, n):
    ops = hip_ops
    rng = np.random.RandomState(n % 97)
    a = rng.randint(0, 1000, n).astype(np.int32)
    d = ops.to_device(a) if n else torch.empty(0, dtype=torch.int32, device=ops.device)
    out = torch.empty(n + 1, dtype=torch.int64, device=ops.device)
    work = ops._work(ops.lib.pk_scan_work_bytes(n))
    _lib.check(ops.lib.pk_exclusive_scan_i32(ops.stream(), n, _ptr(d), _ptr(out), _ptr(work)), 'scan')
    assert np.array_equal(ops.to_host(out), np.r_[0, np.cumsum(a.astype(np.int64))])


@pytest.mark.parametrize('key_bytes,bits,n', [(4, 17, 50001), (4, 32, 300000), (8, 37, 777777), (8, 64, 4099), (4, 3, 70000),
                                              (8, 9, 1)])
def test_radix_sort_pairs_is_a_stable_sort(hip_ops, key_bytes, bits, n):
    ops = hip_ops
    rng = np.random.RandomState(bits)
    if bits < 20:
        keys = rng.randint(0, 1 << bits, n).astype(np.uint64)       # many duplicates: stability matters
    else:
        keys = rng.randint(0, 1 << 62, n, dtype=np.int64).astype(np.uint64) & np.uint64((1 << bits) - 1 if bits < 64 else 2 ** 64 - 1)
    kt = np.uint32 if key_bytes == 4 else np.uint64
    keys = keys.astype(kt)
    vals = np.arange(n, dtype=np.uint32)
    kd = torch.from_numpy(keys.view(np.int32 if key_bytes == 4 else np.int64)).to(ops.device)
    vd = torch.from_numpy(vals.view(np.int32)).to(ops.device)
    kt_d, vt_d = torch.empty_like(kd), torch.empty_like(vd)
    work = ops._work(ops.lib.pk_radix_work_bytes(n))
    in_tmp = C.c_int32(0)
    _lib.check(ops.lib.pk_radix_sort_pairs(ops.stream(), n, key_bytes, _ptr(kd), _ptr(vd), _ptr(kt_d), _ptr(vt_d), bits,
                                           _ptr(work), C.byref(in_tmp)), 'radix')
    ks = (kt_d if in_tmp.value else kd).cpu().numpy().view(kt)
    vs = (vt_d if in_tmp.value else vd).cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind='stable')
    assert np.array_equal(ks, keys[order]) and np.array_equal(vs, vals[order])


@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
@pytest.mark.parametrize('interleaved', [False, True])
def test_coo_to_csr_matches_numpy_and_sums_duplicates(hip_ops, vdtype, interleaved):
    rng = np.random.RandomState(3)
    n_rows, n_cols, nnz = 5000, 1300, 400000
    rows = rng.randint(0, n_rows, nnz)
    rows[rows % 17 == 3] = 11                                   # a long row; rows 3, 20, ... stay empty
    cols = (rng.zipf(1.3, nnz) - 1) % n_cols                    # skewed: many duplicates
    vals = rng.randint(1, 6, nnz).astype(vdtype) if vdtype == np.float32 else rng.randn(nnz)
    if interleaved:
        idx = np.ascontiguousarray(np.stack([rows, cols], axis=1).astype(np.int64))
        A = hip_ops.csr_from_coo(idx[:, 0], idx[:, 1], vals, (n_rows, n_cols))
    else:
        A = hip_ops.csr_from_coo(rows.astype(np.int32), cols, vals, (n_rows, n_cols))
    indptr, indices, values = coo_to_csr(rows, cols, vals.astype(np.float64), (n_rows, n_cols))
    assert A.nnz == len(indices) < nnz
    assert np.array_equal(hip_ops.to_host(A.indptr), indptr)
    assert np.array_equal(hip_ops.to_host(A.indices), indices)
    got = hip_ops.to_host(A.values).astype(np.float64)
    if vdtype == np.float32:
        assert np.array_equal(got, values)                      # small integers: exact in any order
    else:
        assert np.allclose(got, values, rtol=1e-13, atol=1e-13)
    with pytest.raises(ValueError):
        hip_ops.csr_from_coo(np.r_[rows, n_rows], np.r_[cols, 0], np.r_[vals, vals[:1]], (n_rows, n_cols))
    E = hip_ops.csr_from_coo(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), (7, 5))
    assert E.nnz == 0 and not hip_ops.to_host(E.indptr).any()


def _rand_csr(seed, n_rows, n_cols, mean, vdtype=np.float32):
    rng = np.random.RandomState(seed)
    counts = rng.poisson(mean, n_rows).clip(0, n_cols)
    counts[5] = min(n_cols, 3000)
    counts[[0, 9, n_rows - 1]] = 0
    indptr = np.r_[0, np.cumsum(counts)].astype(np.int64)
    w = 1.0 / np.arange(1, n_cols + 1) ** 0.7
    indices = np.concatenate([np.sort(rng.choice(n_cols, c, replace=False, p=w / w.sum())) for c in counts]).astype(np.int32)
    values = rng.randint(1, 6, indptr[-1]).astype(vdtype) if vdtype == np.float32 else rng.randn(indptr[-1])
    return indptr, indices, values


@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
def test_csr_transpose_plain_and_blocked(hip_ops, vdtype):
    ops = hip_ops
    n_rows, n_cols = 9000, 4000
    indptr, indices, values = _rand_csr(1, n_rows, n_cols, 40, vdtype)
    A = ops.csr(indptr, indices, values, (n_rows, n_cols))
    tp, ti, tv = csr_transpose(indptr, indices, values, n_cols)
    T = A.T
    assert T.shape == (n_cols, n_rows)
    assert np.array_equal(ops.to_host(T.indptr), tp) and np.array_equal(ops.to_host(T.indices), ti)
    assert np.array_equal(ops.to_host(T.values), tv)
    # blocked image: row b * n_cols + c = column c restricted to rows [b * rpb, (b + 1) * rpb)
    rpb = 2048
    Tb = ops.csr_transpose(A, rows_per_block=rpb)
    n_blocks = -(-n_rows // rpb)
    assert Tb.shape == (n_blocks * n_cols, n_rows)
    bp, bi, bv = (ops.to_host(x) for x in (Tb.indptr, Tb.indices, Tb.values))
    S = sps.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n_rows, n_cols))
    for b in range(n_blocks):
        sub = S[b * rpb:(b + 1) * rpb].tocsc()
        sub.sort_indices()
        lo, hi = bp[b * n_cols], bp[(b + 1) * n_cols]
        assert np.array_equal(bp[b * n_cols:(b + 1) * n_cols + 1] - lo, sub.indptr)
        assert np.array_equal(bi[lo:hi], sub.indices + b * rpb) and np.array_equal(bv[lo:hi].astype(np.float64), sub.data)
    # the blocked product adds up to the plain one, whatever the block size
    rng = np.random.RandomState(2)
    for nc in (64, 51, 130):
        Y = ops.to_device(rng.randn(n_rows, nc))
        ref = S.T @ ops.to_host(Y)
        for rows_per_block in (1024, 4096, 8192):
            got = ops.to_host(ops.spmm(A.T_blocked(rows_per_block), Y))
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-13
        out = ops.empty(n_cols, nc + 3)
        out.fill_(7.0)
        ops.spmm(A.T_blocked(4096), Y, out=out[:, :nc])          # strided output, overwritten (not accumulated into)
        assert np.abs(ops.to_host(out)[:, :nc] - ref).max() / np.abs(ref).max() < 1e-13 and float(out[:, nc:].min()) == 7.0


def test_sorted_relabel_counts_and_device_plan(hip_ops):
    ops = hip_ops
    n_rows, n_cols = 7000, 2500
    indptr, indices, values = _rand_csr(4, n_rows, n_cols, 30)
    A = ops.csr(indptr, indices, values, (n_rows, n_cols), split=256)
    # per-item counts
    assert np.array_equal(ops.item_counts(A), np.bincount(indices, minlength=n_cols))
    # renaming with re-sorted rows
    perm = np.random.RandomState(5).permutation(n_cols).astype(np.int32)
    B = ops.csr_relabel_cols(A, perm)
    S = sps.csr_matrix((values.astype(np.float64), perm[indices], indptr), shape=(n_rows, n_cols))
    S.sort_indices()
    assert np.array_equal(ops.to_host(B.indices), S.indices) and np.array_equal(ops.to_host(B.values).astype(np.float64), S.data)
    assert B.sorted_cols and np.array_equal(ops.to_host(B.indptr), indptr)
    # the device-built plan equals the host restatement, array by array
    want = build_row_tasks(indptr, 256)
    assert (A.n_tasks, A.n_long, A.n_slots) == (len(want['task_row']), len(want['long_row']), want['n_slots'])
    for k in ('task_row', 'task_begin', 'task_end', 'task_slot', 'long_row', 'long_slot_begin', 'long_slot_end'):
        assert np.array_equal(ops.to_host(A.plan[k])[:len(want[k])], want[k]), k
    t0, nt, l0, nl = A.task_range(3, 4000)
    assert t0 == want['row_first_task'][3] and nt == want['row_first_task'][4000] - want['row_first_task'][3]
    assert (l0, nl) == tuple(int(x) for x in (np.searchsorted(want['long_row'], 3),
                                              np.searchsorted(want['long_row'], 4000) - np.searchsorted(want['long_row'], 3)))
    # a row range as its own launch still computes those rows
    X = ops.to_device(np.random.RandomState(6).randn(n_cols, 20))
    full = ops.spmm(A, X)
    part = torch.zeros_like(full)
    ops.spmm(A, X, out=part, rows=(3, 4000))
    assert torch.equal(part[3:4000], full[3:4000]) and float(part[4000:].abs().sum()) == 0.0


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_row_sort_of_the_renaming_at_every_row_length_class(hip_ops, dtype):
    """pk_csr_relabel_sorted sorts the rows where they are (round 4): empty rows, one and two entries, the one-wave
    workgroups' limit (1 024) and one beyond it, the LDS limit of the long-row workgroups (16 384) and one beyond it
    (global scratch), a 40 000-entry row — values follow their columns, bit for bit, in both value kinds."""
    ops = hip_ops
    rng = np.random.RandomState(3)
    n_cols = 70000
    lens = [0, 1, 2, 3, 63, 64, 65, 0, 1023, 1024, 1025, 2047, 2048, 5000, 16383, 16384, 16385, 40000, 0, 7] + list(rng.randint(0, 300, 500))
    indptr = np.r_[0, np.cumsum(lens)].astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, c, replace=False)) for c in lens]).astype(np.int32)
    values = rng.rand(int(indptr[-1])).astype(dtype) + 0.5
    A = ops.csr(indptr, indices, values, (len(lens), n_cols))
    perm = rng.permutation(n_cols).astype(np.int32)
    B = ops.csr_relabel_cols(A, perm)
    S = sps.csr_matrix((values.copy(), perm[indices], indptr), shape=(len(lens), n_cols))     # (sort_indices works in place)
    S.sort_indices()
    assert np.array_equal(ops.to_host(B.indices), S.indices)
    assert np.array_equal(ops.to_host(B.values), S.data.astype(dtype))
    assert np.array_equal(ops.to_host(B.indptr), indptr) and B.sorted_cols
    # a second renaming of the result (the serving order after the popularity order) — and back
    inv = np.empty_like(perm)
    inv[perm] = np.arange(n_cols, dtype=np.int32)
    C_ = ops.csr_relabel_cols(B, inv)
    assert np.array_equal(ops.to_host(C_.indices), indices) and np.array_equal(ops.to_host(C_.values), values)


def test_item_order_on_the_device_equals_the_host_order(hip_ops):
    """HipOps.item_order (counts -> own stable radix sort -> inverse, one copy back) against csr.popularity_order on the
    same counts: descending popularity, ties by id — including many ties (most items of a long tail share a count)."""
    from polara_amd.csr import popularity_order
    ops = hip_ops
    for seed, n_rows, n_cols, avg in ((1, 5000, 3000, 12), (2, 300, 20000, 3), (3, 12, 5, 2)):
        indptr, indices, values = _rand_csr(seed, n_rows, n_cols, avg)
        A = ops.csr(indptr, indices, values, (n_rows, n_cols))
        rank, inv, counts, rank_dev = ops.item_order(A)
        want_counts = np.bincount(indices, minlength=n_cols)
        want_rank, want_inv = popularity_order(None, n_cols, counts=want_counts)
        assert np.array_equal(counts, want_counts) and counts.dtype == np.int64
        assert np.array_equal(rank, want_rank) and np.array_equal(inv, want_inv) and rank.dtype == np.int32
        assert np.array_equal(ops.to_host(rank_dev), want_rank)
        B = ops.csr_relabel_cols(A, rank_dev)
        assert np.array_equal(ops.to_host(B.indices), ops.to_host(ops.csr_relabel_cols(A, want_rank).indices))


def test_unsorted_long_rows_get_sorted_for_the_seen_tiles(hip_ops):
    """rows longer than the in-LDS sort of the seen-tile builder after a bare renaming: re-sorted by the own kernels"""
    ops = hip_ops
    n_rows, n_cols = 40, 60000
    rng = np.random.RandomState(8)
    counts = np.full(n_rows, 50)
    counts[7] = ops.lib.pk_seen_tiles_max_unsorted_row() + 500
    indptr = np.r_[0, np.cumsum(counts)].astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, c, replace=False)) for c in counts]).astype(np.int32)
    A = ops.csr(indptr, indices, np.ones(indptr[-1], np.float32), (n_rows, n_cols))
    perm = rng.permutation(n_cols).astype(np.int32)
    B = ops.csr_relabel_cols(A, perm, sort=False)
    tiles, ntiles = B.seen_tiles()
    C_ = ops.csr_relabel_cols(A, perm, sort=True)
    tiles2, ntiles2 = C_.seen_tiles()
    assert torch.equal(ntiles, ntiles2)
    nt = ops.to_host(ntiles)
    t1, t2 = ops.to_host(tiles), ops.to_host(tiles2)
    for r in range(n_rows):
        assert np.array_equal(t1[indptr[r]:indptr[r] + nt[r]], t2[indptr[r]:indptr[r] + nt[r]])


def test_rows_by_length_is_a_stable_descending_order(hip_ops):
    ops = hip_ops
    n_rows, n_cols = 20000, 3000
    indptr, indices, values = _rand_csr(12, n_rows, n_cols, 25)
    A = ops.csr(indptr, indices, values, (n_rows, n_cols))
    P, perm = A.by_activity()
    perm = ops.to_host(perm)
    counts = np.diff(indptr)
    assert np.array_equal(perm, np.argsort(-counts, kind='stable'))
    assert np.array_equal(ops.to_host(P.indptr), np.r_[0, np.cumsum(counts[perm])])
    S = sps.csr_matrix((values, indices, indptr), shape=(n_rows, n_cols))[perm]
    assert np.array_equal(ops.to_host(P.indices), S.indices) and np.array_equal(ops.to_host(P.values), S.data)
    # the scoring pass gives the same lists with and without the grouping
    from polara_amd import scoring
    rng = np.random.RandomState(0)
    V = np.linalg.qr(rng.randn(n_cols, 16))[0] * np.linspace(3, 0.3, n_cols)[:, None]
    F = scoring.FactorImage(ops, ops.to_device(V))
    a, sa = scoring.recommend(ops, F, A, 10, True, return_scores=True)
    b, sb = scoring.recommend(ops, F, A, 10, True, return_scores=True, order_users=False)
    assert torch.equal(a, b) and torch.equal(sa, sb)
    assert torch.equal(scoring.recommend(ops, F, A, 10, True), a)


@pytest.mark.parametrize('vdtype', [np.float32, np.float64])
def test_diagonal_scaling_and_mode_plan_match_numpy(hip_ops, vdtype):
    """pk_csr_scale_f64 (ScaledMatrixMixin's D_r A D_c, bit-equal to the left-to-right NumPy product) and
    HipOps.mode_plan (a tensor's entries ordered by one mode on the device: same stable order, row pointers and index
    columns as the host's argsort(kind='stable'))."""
    ops = hip_ops
    n_rows, n_cols = 7000, 900
    indptr, indices, values = _rand_csr(5, n_rows, n_cols, 30)
    values = values.astype(vdtype)
    A = ops.csr(indptr, indices, values, (n_rows, n_cols))
    rng = np.random.RandomState(1)
    rs, cs = rng.rand(n_rows) + 0.5, rng.rand(n_cols) + 0.5
    B = ops.csr_scale(A, rs, cs)
    rows = np.repeat(np.arange(n_rows), np.diff(indptr))
    want = (rs[rows] * values.astype(np.float64)) * cs[indices]
    assert B.values.dtype == torch.float64 and np.array_equal(ops.to_host(B.values).view(np.int64), want.view(np.int64))
    assert torch.equal(B.indptr, A.indptr) and torch.equal(B.indices, A.indices)
    X = rng.randn(n_cols, 8)
    ref = sps.csr_matrix((want, indices, indptr), shape=(n_rows, n_cols)) @ X
    assert np.allclose(ops.to_host(ops.spmm(B, ops.to_device(X))), ref, rtol=1e-12, atol=1e-12)
    # mode plans
    nnz, shape = 50000, (300, 47, 5)
    idx = np.stack([rng.randint(0, s, nnz) for s in shape], 1).astype(np.int64)
    idx[:4000, 1] = 3                                              # a long row of mode 1
    idx_dev = torch.from_numpy(idx).to(ops.device)
    for mode0, mu, mv in ((0, 2, 1), (1, 2, 0), (2, 1, 0)):
        plan, iu, iv, order = ops.mode_plan(idx_dev, mode0, mu, mv, shape[mode0], split=256)
        want_order = np.argsort(idx[:, mode0], kind='stable')
        assert np.array_equal(ops.to_host(order), want_order)
        assert np.array_equal(ops.to_host(iu), idx[want_order, mu]) and np.array_equal(ops.to_host(iv), idx[want_order, mv])
        host = build_row_tasks(np.r_[0, np.cumsum(np.bincount(idx[:, mode0], minlength=shape[mode0]))].astype(np.int64), split=256)
        assert plan['n_tasks'] == len(host['task_row']) and plan['n_long'] == len(host['long_row'])
        for k in ('task_row', 'task_begin', 'task_end', 'task_slot', 'long_row', 'long_slot_begin', 'long_slot_end'):
            assert np.array_equal(ops.to_host(plan[k])[:len(host[k])], host[k]), (mode0, k)


@pytest.mark.parametrize('n_bins', [1, 5, 32, 33, 1000, 26744, 36864, 36865, 100000, 2_400_000])
def test_count_i32_matches_bincount(hip_ops, n_bins):
    """pk_count_i32 (item popularity, mode sizes): the LDS-histogram form (<= 36 864 bins; wave ballots for <= 32 bins), the
    direct form beyond and for short inputs; skewed keys (every key in one bin), keys outside [0, n_bins) ignored."""
    ops = hip_ops
    rng = np.random.RandomState(n_bins)
    for n in (1, 100, 5000, 3_000_001):
        keys = (rng.zipf(1.3, n) % n_bins).astype(np.int32)
        keys[::17] = n_bins - 1
        keys[5::1001] = -3                       # out of range: not counted
        keys[7::1003] = n_bins + 2
        kd = torch.from_numpy(keys).to(ops.device)
        out = torch.full((n_bins,), -1, dtype=torch.int32, device=ops.device)
        _lib.check(ops.lib.pk_count_i32(ops.stream(), n, _ptr(kd), n_bins, _ptr(out)), 'pk_count_i32')
        valid = keys[(keys >= 0) & (keys < n_bins)]
        assert np.array_equal(ops.to_host(out), np.bincount(valid, minlength=n_bins))
    one = torch.full((200000,), min(3, n_bins - 1), dtype=torch.int32, device=ops.device)
    out = torch.empty(n_bins, dtype=torch.int32, device=ops.device)
    _lib.check(ops.lib.pk_count_i32(ops.stream(), 200000, _ptr(one), n_bins, _ptr(out)), 'pk_count_i32')
    assert int(out[min(3, n_bins - 1)].item()) == 200000 and int(out.sum().item()) == 200000


@pytest.mark.gpu
@pytest.mark.parametrize('n,K', [(1, 3), (63, 8), (1000, 50), (26744, 50), (70001, 17)])
def test_norm_order_is_numpy_s_stable_argsort_of_the_row_norms(hip_ops, n, K):
    """pk_row_norm_order_f64 (the serving order of the catalogue, built on the device: own radix sort on the bit patterns of
    the fp64 norms + one gather) against `np.argsort(-np.linalg.norm(V, axis=1), kind='stable')` — the host statement of the
    plugin surface (polara_amd/models.py) — with tied norms (repeated rows, zero rows): order, its inverse, the gathered rows."""
    rng = np.random.RandomState(n + K)
    V = rng.standard_normal((n, K)) * np.exp(-3.0 * rng.rand(n))[:, None]
    if n > 10:
        V[rng.choice(n, n // 7, replace=False)] = V[0]          # ties: repeated rows ...
        V[rng.choice(n, n // 11, replace=False)] = 0.0          # ... and zero rows
    Vd = hip_ops.to_device(V)
    order, rank, Vs = hip_ops.norm_order(Vd)
    norms = hip_ops.to_host(torch.sqrt((Vd * Vd).sum(1)))      # the kernel accumulates with fma: compare orders on ITS arithmetic's norms, to rounding
    want = np.argsort(-np.linalg.norm(V, axis=1), kind='stable')
    got = hip_ops.to_host(order).astype(np.int64)
    assert sorted(got.tolist()) == list(range(n))
    nv = np.linalg.norm(V, axis=1)
    assert np.all(np.diff(nv[got]) <= 1e-15 * nv.max())           # descending to rounding
    same = nv[got] == nv[want]
    assert same.mean() > 0.999                                     # (norms within one rounding of each other may swap)
    ties_ok = all(np.all(np.diff(got[a:b]) > 0) for a, b in _runs(nv[got]) if b - a > 1 and nv[got][a] == 0.0)
    assert ties_ok                                                 # exact ties (the zero rows) keep ascending ids: stable
    r = hip_ops.to_host(rank).astype(np.int64)
    assert np.array_equal(r[got], np.arange(n))
    assert np.array_equal(hip_ops.to_host(Vs), V[got])
    _ = norms


def _runs(x):
    edges = np.flatnonzero(np.r_[True, x[1:] != x[:-1], True])
    return list(zip(edges[:-1], edges[1:]))
