"""(round 5) The rank-200 / top-50 sweep of the configs[4] shard under its run-time switches: one wave per user group against the
two-waves-per-group instance (PK_SCORE_KHALF), item chunks of different lengths (L2 locality against launches), bootstrap
lengths.  Builds the shard's model once, then times `score_candidates` (all its chunk launches) per setting.
    python tools/probes/rank200_sweep_probe.py [--users 500000] [--settings khalf=0 khalf=1 khalf=0,chunk=48 ...]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import planted_csr, csr_to_numpy
from polara_amd.csr import popularity_order
from polara_amd.solver import svd_topk
from polara_amd import scoring


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=500_000)
    ap.add_argument('--items', type=int, default=500_000)
    ap.add_argument('--rank', type=int, default=200)
    ap.add_argument('--topk', type=int, default=50)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--profile', action='store_true', help='read the cycle counters of a PK_SCORE_PROFILE probe library')
    ap.add_argument('--settings', nargs='*', default=['khalf=0', 'khalf=1'])
    args = ap.parse_args()
    ops = HipOps('cuda:0')
    csr = planted_csr(args.users, args.items, 50, args.rank // 4, levels=5, seed=5, device='cuda:0', min_items=20,
                      max_items=2000, chunk_rows=1024)
    c = csr_to_numpy(csr); del csr
    n_users, n_items = c['shape']
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, _ = popularity_order(c['indices'], n_items)
    A = ops.csr_relabel_cols(A, rank_of)
    _, sigma, V, st = svd_topk(ops, A, args.rank)
    order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
    rank2 = torch.empty_like(order2); rank2[order2] = torch.arange(n_items, device=order2.device)
    V = V[order2].contiguous()
    A = ops.csr_relabel_cols(A, rank2, sort=False)
    F = scoring.FactorImage(ops, V)
    ref = None
    prof = hasattr(ops.lib, 'pk_debug_profile') if args.profile else False     # a PK_SCORE_PROFILE [+ PROFILE2] probe library
    if args.profile:
        import ctypes
        ops.lib.pk_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = (ctypes.c_ulonglong * 8)()
    for setting in args.settings:
        kv = dict(x.split('=') for x in setting.split(','))
        os.environ['PK_SCORE_KHALF'] = kv.get('khalf', '1')
        os.environ['PK_SCORE_BOOT_TILES'] = kv.get('boot', '16')
        os.environ['PK_SCORE_ABLATE'] = kv.get('ablate', '0')      # 1: no seen walk, 2: no pushes (lists wrong: only the time counts)
        ops.score_tiles_per_chunk = int(kv.get('chunk', 0))
        for _ in range(2):
            recs = scoring.recommend(ops, F, A, args.topk, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            recs = scoring.recommend(ops, F, A, args.topk, True)
        torch.cuda.synchronize()
        pass_ms = 1e3 * (time.perf_counter() - t0) / args.reps
        ops.timers = {}
        stats = {}
        recs = scoring.recommend(ops, F, A, args.topk, True, stats=stats)
        torch.cuda.synchronize()
        ms = {k: round(float(np.sum([a.elapsed_time(b) for a, b, _ in v])), 3) for k, v in ops.timers.items()}
        n_launch = len(ops.timers.get('score_candidates', []))
        ops.timers = None
        extra = {}
        if args.profile:
            ops.lib.pk_debug_profile(None, 1)
            scoring.recommend(ops, F, A, args.topk, True)
            torch.cuda.synchronize()
            ops.lib.pk_debug_profile(buf, 0)
            names = ('kernel', 'products_or_flush', 'walk', 'push_incl_flush', 'prologue', 'bootstrap', 'n_flush', 'tile_steps')
            d = {k: int(v) for k, v in zip(names, buf)}
            extra = dict(cycles_per_tile_step=round(d['kernel'] / max(d['tile_steps'], 1), 1),
                         shares={k: round(d[k] / max(d['kernel'], 1), 4) for k in names[1:6]}, n_flush=d['n_flush'], tile_steps=d['tile_steps'])
        if ref is None:
            ref = recs.clone()
        print(json.dumps(dict(setting=setting, pass_ms=round(pass_ms, 3), sweep_ms=ms.get('score_candidates'), sweep_timer_entries=n_launch,
                              swept=round(stats['tiles_scored'] / stats['tiles_total'], 5),
                              us_per_tile_step=round(1e3 * ms.get('score_candidates', 0.0) / max(stats['tiles_scored'] / 1024.0, 1e-9), 3), flagged=stats['flagged_users'],
                              same_lists=bool(torch.equal(ref, recs)), kernel_ms=ms, **extra)), flush=True)


if __name__ == '__main__':
    main()
