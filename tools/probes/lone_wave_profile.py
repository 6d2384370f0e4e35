"""Cycle counters of a lone-wave full sweep (library built with PK_SCORE_PROFILE2=1 PK_FAST_BUILD=1): 128 users, no pruning,
no pushes (PK_SCORE_ABLATE=2) — wave-cycles per tile in total, in the products (slot 1), in the mask walk (slot 2)."""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
sys.argv = sys.argv[:1]
bench = B.Bench(B.parse())
ops = bench.ops
c = bench.generate('ml20m')
st, _ = bench.build(c, 50, True)
from polara_amd import scoring
F, A = st['F'], st['A']
n_items = A.shape[1]
for n_users in (128, 32768, 138493):
    T = ops.csr_rows(A, 0, n_users)
    E = ops.spmm(T, F.V)
    Ep, ub = ops.pack_frag_bound(E)
    buf = (ctypes.c_ulonglong * 8)()
    ops.lib.pk_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for abl in ('2', '0'):
        os.environ['PK_SCORE_ABLATE'] = abl
        os.environ['PK_SCORE_BOOT_TILES'] = '0'
        kw = dict(seen_tiles=T.seen_tiles(), seen_dense=T.seen_dense())
        for _ in range(2):
            ops.score_candidates(F.Vp, Ep, n_users, n_items, 50, T.indptr, T.indices, 16, **kw)
        torch.cuda.synchronize()
        ops.lib.pk_debug_profile(None, 1)
        ops.score_candidates(F.Vp, Ep, n_users, n_items, 50, T.indptr, T.indices, 16, **kw)
        torch.cuda.synchronize()
        ops.lib.pk_debug_profile(buf, 0)
        k, prod, walk, push, prol, boot, nfl, tiles = [int(v) for v in buf]
        print(json.dumps(dict(n_users=n_users, ablate=int(abl), tiles=tiles, cycles_per_tile=round(k / tiles, 1),
                              products=round(prod / tiles, 1), walk=round(walk / tiles, 1), push=round(push / tiles, 1),
                              prologue_per_wave=round(prol / max(-(-n_users // 32), 1), 1))), flush=True)
