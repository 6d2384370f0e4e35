"""Two ranks (torchrun, gloo for the exchange, both contexts on the one visible GPU) driving the sharded coarse build
`pk_svd_build_sharded` through ctypes: every rank uploads ITS users, the Gramian-step all-reduce goes through the
`pk_comm` callback the host supplies (here: device -> host, gloo all-reduce, host -> device; a real host hands RCCL's
ncclAllReduce to the same slot), and the factors must equal the single-context build of the whole matrix; then every rank
scores its own users with pk_score_topk against the replicated V and the concatenated lists must equal the
single-context lists.  The library itself is used without torch (torch.distributed is only this test's transport)."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

lib = C.CDLL(os.path.join(ROOT, 'polara_amd', 'libpolarahip.so'))
hip = C.CDLL('libamdhip64.so')
vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double


class Stats(C.Structure):
    _fields_ = [('outer', i32), ('gramian_steps', i32), ('block', i32), ('converged', i32), ('final_rel_residual', f64)]


ALLREDUCE = C.CFUNCTYPE(C.c_int, vp, vp, i64, vp)


class Comm(C.Structure):
    _fields_ = [('rank', i32), ('world', i32), ('allreduce_sum_f64', ALLREDUCE), ('user', vp)]


lib.pk_ctx_create.argtypes, lib.pk_ctx_create.restype = [i32, C.POINTER(vp)], C.c_int
lib.pk_ctx_destroy.argtypes, lib.pk_ctx_destroy.restype = [vp], None
lib.pk_ctx_error.argtypes, lib.pk_ctx_error.restype = [vp], C.c_char_p
lib.pk_ctx_set_option.argtypes, lib.pk_ctx_set_option.restype = [vp, C.c_char_p, i32], C.c_int
lib.pk_ctx_stream.argtypes, lib.pk_ctx_stream.restype = [vp], vp
lib.pk_mat_from_csr.argtypes, lib.pk_mat_from_csr.restype = [vp, i64, i64, i64, vp, vp, vp, i32, C.POINTER(vp)], C.c_int
lib.pk_mat_free.argtypes, lib.pk_mat_free.restype = [vp, vp], None
lib.pk_svd_build.argtypes = [vp, vp, i32, i32, f64, i32, C.c_uint64, vp, vp, vp, C.POINTER(Stats)]
lib.pk_svd_build.restype = C.c_int
lib.pk_svd_build_sharded.argtypes = [vp, vp, C.POINTER(Comm), i32, i32, f64, i32, C.c_uint64, vp, vp, vp, C.POINTER(Stats)]
lib.pk_svd_build_sharded.restype = C.c_int
lib.pk_score_topk.argtypes, lib.pk_score_topk.restype = [vp, i64, i32, vp, vp, i32, i32, vp, vp], C.c_int
hip.hipMemcpy.argtypes, hip.hipMemcpy.restype = [vp, vp, C.c_size_t, C.c_int], C.c_int
hip.hipStreamSynchronize.argtypes, hip.hipStreamSynchronize.restype = [vp], C.c_int


def ptr(a):
    return a.ctypes.data_as(vp)


def check(ctx, rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, lib.pk_ctx_error(ctx).decode()))


def main():
    dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    rank, world = dist.get_rank(), dist.get_world_size()
    from coarse_abi_worker_data import planted      # same generator as the single-context worker
    n_users, n_items, k, topk = 3000, 700, 12, 10
    rows, cols, vals = planted(n_users, n_items, 30, 8, seed=3)
    import scipy.sparse as sps
    A = sps.csr_matrix((vals, (rows, cols)), shape=(n_users, n_items))
    A.sort_indices()
    # contiguous user blocks, nnz-balanced
    bounds = np.searchsorted(A.indptr, np.linspace(0, A.nnz, world + 1)).astype(np.int64)
    bounds[0], bounds[-1] = 0, n_users
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    Al = A[lo:hi]

    calls = {'n': 0, 'bytes': 0}

    def allreduce(user, buf, count, stream):
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                return -1
            host = np.empty(count, dtype=np.float64)
            if hip.hipMemcpy(ptr(host), buf, count * 8, 2) != 0:      # device -> host
                return -1
            t = torch.from_numpy(host)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if hip.hipMemcpy(buf, ptr(host), count * 8, 1) != 0:      # host -> device
                return -1
            calls['n'] += 1
            calls['bytes'] += count * 8
            return 0
        except Exception as exc:                                      # never let an exception cross the C boundary
            print('allreduce callback failed:', exc, flush=True)
            return -1
    cb = ALLREDUCE(allreduce)
    comm = Comm(rank, world, cb, None)

    ctx = vp()
    check(ctx, lib.pk_ctx_create(0, C.byref(ctx)), 'pk_ctx_create')
    assert lib.pk_ctx_stream(ctx)

    def upload(M):
        h = vp()
        ip, ix, vv = M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(np.float64)
        check(ctx, lib.pk_mat_from_csr(ctx, M.shape[0], M.shape[1], M.nnz, ptr(ip), ptr(ix), ptr(vv), 1, C.byref(h)), 'pk_mat_from_csr')
        return h
    Ml = upload(Al)
    check(ctx, lib.pk_ctx_set_option(ctx, b'dist_overlap', 0), 'pk_ctx_set_option')      # 0 = never split a product
    sigma, V = np.empty(k), np.empty((n_items, k), order='F')
    U = np.empty((hi - lo, k), order='F')
    st = Stats()
    check(ctx, lib.pk_svd_build_sharded(ctx, Ml, C.byref(comm), k, 0, 0.0, 0, 7, ptr(sigma), ptr(V), ptr(U), C.byref(st)),
          'pk_svd_build_sharded')
    # one all-reduce of Z per Gramian step + one of the Rayleigh-Ritz matrix per outer iteration (+ ONE scalar at the start:
    # the entry count of the whole matrix, from which every rank picks the same method): nothing else leaves the rank
    assert st.converged == 1 and calls['n'] == st.gramian_steps + st.outer + 1, (calls, st.gramian_steps, st.outer)
    # The same build with every Gramian product cut into TWO column panels (context option dist_overlap = 2; on its own the library
    # splits when the modelled exchange reaches 0.4 ms): the first panel's sum is handed to the callback on a SIDE stream
    # while the second panel's products are enqueued — twice the all-reduce calls for the steps, the same bytes, the same
    # factors.
    n0, b0 = calls['n'], calls['bytes']
    check(ctx, lib.pk_ctx_set_option(ctx, b'dist_overlap', 2), 'pk_ctx_set_option')      # 2 = whenever possible
    sigma2, V2 = np.empty(k), np.empty((n_items, k), order='F')
    st2 = Stats()
    streams = set()
    side_calls = [0]
    inner = allreduce
    main_stream = lib.pk_ctx_stream(ctx)

    def allreduce_seen(user, buf, count, stream):
        streams.add(stream)
        side_calls[0] += int(stream != main_stream)
        return inner(user, buf, count, stream)
    cb2 = ALLREDUCE(allreduce_seen)
    comm2 = Comm(rank, world, cb2, None)
    check(ctx, lib.pk_svd_build_sharded(ctx, Ml, C.byref(comm2), k, 0, 0.0, 0, 7, ptr(sigma2), ptr(V2), None, C.byref(st2)),
          'pk_svd_build_sharded (two panels)')
    check(ctx, lib.pk_ctx_set_option(ctx, b'dist_overlap', 0), 'pk_ctx_set_option')
    assert st2.converged == 1 and st2.gramian_steps == st.gramian_steps and st2.outer == st.outer
    # every product of a full-width block went in two panels (blocks narrowed by locking, and the rotations of the
    # Rayleigh-Ritz steps, go whole): one more call per split product, the first panel's on the side stream
    assert side_calls[0] >= 1 and calls['n'] - n0 == st2.gramian_steps + st2.outer + 1 + side_calls[0], (calls, n0, side_calls)
    assert calls['bytes'] - b0 == b0
    assert len(streams) == 2 and main_stream in streams, streams     # the context's stream and the side stream
    # (a half-width panel runs another instance of the product kernel — more lane groups per row, another summation order:
    # the factors agree to rounding, not bit for bit)
    assert np.allclose(sigma, sigma2, rtol=1e-11, atol=0.0), np.abs(sigma - sigma2).max()
    assert np.abs(V @ V.T - V2 @ V2.T).max() < 1e-9
    # the same on every rank, bit for bit (all-reduced inputs, identical arithmetic)
    both = [torch.empty(n_items * k + k, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, torch.from_numpy(np.r_[sigma, V.ravel(order='F')]))
    assert all(torch.equal(both[0], b) for b in both[1:])
    # this rank's user factors: U_local = A_local V / sigma
    assert np.allclose(U, (Al @ V) / sigma, rtol=1e-10, atol=1e-12)
    # lists of my users against the replicated V
    recs = np.empty((hi - lo, topk), dtype=np.int64)
    check(ctx, lib.pk_score_topk(ctx, n_items, k, ptr(V), Ml, topk, 1, ptr(recs), None), 'pk_score_topk')
    gathered = [None] * world
    dist.all_gather_object(gathered, recs)
    if rank == 0:
        Mf = upload(A)
        sigma1, V1 = np.empty(k), np.empty((n_items, k), order='F')
        st1 = Stats()
        check(ctx, lib.pk_svd_build(ctx, Mf, k, 0, 0.0, 0, 7, ptr(sigma1), ptr(V1), None, C.byref(st1)), 'pk_svd_build')
        assert np.allclose(sigma, sigma1, rtol=1e-10), (sigma, sigma1)
        assert np.allclose(sigma, np.linalg.svd(A.toarray(), compute_uv=False)[:k], rtol=1e-9)
        P, P1 = V @ V.T, V1 @ V1.T
        assert np.abs(P - P1).max() < 1e-8
        recs1 = np.empty((n_users, topk), dtype=np.int64)
        check(ctx, lib.pk_score_topk(ctx, n_items, k, ptr(V), Mf, topk, 1, ptr(recs1), None), 'pk_score_topk (whole)')
        assert np.array_equal(np.concatenate(gathered, axis=0), recs1)
        lib.pk_mat_free(ctx, Mf)
        print('COARSE_SHARDED_OK gramian_steps %d, all-reduce calls %d, %.1f MB exchanged per rank' % (
            st.gramian_steps, calls['n'], calls['bytes'] / 1e6), flush=True)
    lib.pk_mat_free(ctx, Ml)
    lib.pk_ctx_destroy(ctx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
