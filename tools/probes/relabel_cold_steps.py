"""The first popularity relabelling of a process, statement by statement (bench.py's build_cold.relabel_popularity_s): every
statement of HipOps.item_order / csr_relabel_cols timed with a device synchronisation after it.
    python tools/probes/relabel_cold_steps.py"""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
sys.argv = ['bench.py']
import bench
from polara_amd import _lib
from polara_amd.ops import _ptr
args = bench.parse()
B = bench.Bench(args)
ops = B.ops
c = B.generate('ml20m')
n_users, n_items = c['shape']
rows = []


def lap(name, t0):
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rows.append((name, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))


for rep in range(2):
    rows.clear()
    torch.cuda.synchronize()
    t = time.perf_counter(); A = ops.csr(c['indptr'], c['indices'], c['values'], (n_users, n_items)); lap('upload', t)
    n = n_items
    t = time.perf_counter(); counts = torch.empty(n, dtype=torch.int32, device=ops.device); lap('empty counts', t)
    t = time.perf_counter(); _lib.check(ops.lib.pk_count_i32(ops.stream(), A.indices.numel(), _ptr(A.indices), n, _ptr(counts)), 'count'); lap('pk_count_i32', t)
    t = time.perf_counter(); mx = counts.max(); lap('counts.max()', t)
    t = time.perf_counter(); keys = (mx - counts).contiguous(); lap('max - counts', t)
    t = time.perf_counter(); pos = torch.arange(n, dtype=torch.int32, device=ops.device); lap('arange', t)
    t = time.perf_counter(); keys_tmp, pos_tmp = torch.empty_like(keys), torch.empty_like(pos); lap('empty_like x2', t)
    in_tmp = C.c_int32(0)
    t = time.perf_counter(); work = ops._work(ops.lib.pk_radix_work_bytes(n)); lap('_work', t)
    bits = int(A.indices.numel()).bit_length()
    t = time.perf_counter()
    _lib.check(ops.lib.pk_radix_sort_pairs(ops.stream(), n, 4, _ptr(keys), _ptr(pos), _ptr(keys_tmp), _ptr(pos_tmp), bits, _ptr(work), C.byref(in_tmp)), 'sort')
    lap('pk_radix_sort_pairs', t)
    inv = pos_tmp if in_tmp.value else pos
    t = time.perf_counter(); rank = torch.empty_like(inv); il = inv.long(); lap('inv.long()', t)
    t = time.perf_counter(); rank[il] = torch.arange(n, dtype=torch.int32, device=ops.device); lap('index_put', t)
    t = time.perf_counter(); cat = torch.cat([rank, inv, counts]); lap('cat', t)
    t = time.perf_counter(); host = cat.cpu().numpy(); lap('.cpu()', t)
    nnz = int(A.indices.numel())
    t = time.perf_counter(); idx = torch.empty_like(A.indices); lap('relabel: empty idx (%d MB)' % (nnz * 4 >> 20), t)
    t = time.perf_counter(); val = torch.empty_like(A.values); lap('relabel: empty val', t)
    wb = ops.lib.pk_csr_relabel_work_bytes(nnz)
    t = time.perf_counter(); work = ops._work(wb); lap('relabel: _work (%d MB)' % (wb >> 20), t)
    t = time.perf_counter()
    _lib.check(ops.lib.pk_csr_relabel_sorted(ops.stream(), A.shape[0], A.shape[1], nnz, _ptr(A.indptr), _ptr(A.indices), _ptr(A.values),
                                             A.val_kind, _ptr(rank), _ptr(idx), _ptr(val), _ptr(work)), 'relabel')
    lap('relabel: pk_csr_relabel_sorted', t)
    del idx, val
    print('pass %d: total %.2f ms' % (rep, sum(h + d for _, h, d in rows)))
    for name, h, d in rows:
        print('  %-22s host %8.3f  device tail %8.3f' % (name, h, d))
    del A
for mb in (64, 256, 512, 1024, 2048):
    torch.cuda.synchronize(); t = time.perf_counter(); x = torch.empty(mb << 20, dtype=torch.uint8, device=ops.device); torch.cuda.synchronize()
    print('fresh torch.empty of %d MB: %.2f ms' % (mb, 1e3 * (time.perf_counter() - t)))
