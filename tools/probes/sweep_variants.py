"""Sweep-shape sweep on one workload: per configuration of (threshold bootstrap tiles, two-phase head tiles, phase-2
splits) the time of the candidate sweep entry point, of the whole pass, the tiles scored and the longest chain; lists
compared with the first configuration's.  usage: python tools/probes/sweep_variants.py [ml20m|s1m] [rank] [topk]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench as B


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    topk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    catalogue = sys.argv[4] if len(sys.argv) > 4 else 'svd'
    prune = os.environ.get('PRUNE', '1') == '1'
    sys.argv = sys.argv[:1]
    args = B.parse()
    bench = B.Bench(args)
    ops = bench.ops
    c = bench.generate(wl)
    st, _ = bench.build(c, rank, True, catalogue)
    from polara_amd import scoring
    configs = [dict(boot=0, head=0, s2=3), dict(boot=16, head=0, s2=3), dict(boot=8, head=0, s2=3), dict(boot=32, head=0, s2=3),
               dict(boot=0, head=32, s2=3), dict(boot=16, head=32, s2=3), dict(boot=16, head=32, s2=7), dict(boot=16, head=64, s2=3),
               dict(boot=16, head=64, s2=7), dict(boot=16, head=16, s2=3), dict(boot=16, head=16, s2=7)]
    if os.environ.get('VARIANTS'):      # "boot,head,s2 boot,head,s2 ..."
        configs = [dict(zip(('boot', 'head', 's2', 'shared'), (int(x) for x in v.split(',')))) for v in os.environ['VARIANTS'].split()]
    n_users_cap = int(os.environ.get('USERS', '0'))
    if n_users_cap:                     # a shard: the first users of the activity-ordered matrix would not be typical; take every k-th
        A = st['A']
        step = max(1, A.shape[0] // n_users_cap)
        ip = ops.to_host(A.indptr)
        rows = np.arange(0, A.shape[0], step)[:n_users_cap]
        cnt = np.diff(ip)[rows]
        sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows])
        st['A'] = ops.csr(np.r_[0, np.cumsum(cnt)].astype(np.int64), ops.to_host(A.indices)[sel], ops.to_host(A.values)[sel], (len(rows), A.shape[1]))
    ref = None
    out = []
    for cfg in configs:
        os.environ['PK_SCORE_BOOT_TILES'] = str(cfg['boot'])
        os.environ['PK_SCORE_HEAD_TILES'] = str(cfg['head'])
        os.environ['PK_SCORE_PHASE2_SPLITS'] = str(cfg['s2'])
        os.environ['PK_SCORE_SHARED'] = str(cfg.get('shared', 0))
        for _ in range(3):
            recs = scoring.recommend(ops, st['F'], st['A'], topk, True, prune=prune)
        torch.cuda.synchronize()
        ops.timers = {}
        for _ in range(20):
            scoring.recommend(ops, st['F'], st['A'], topk, True, batches=1, prune=prune)
        torch.cuda.synchronize()
        ms = {k: float(np.mean(B.events_ms(v))) for k, v in ops.timers.items()}
        ops.timers = None
        t0 = time.perf_counter()
        for _ in range(20):
            recs = scoring.recommend(ops, st['F'], st['A'], topk, True, prune=prune)
        torch.cuda.synchronize()
        pass_ms = (time.perf_counter() - t0) / 20 * 1e3
        stats = {}
        scoring.recommend(ops, st['F'], st['A'], topk, True, stats=stats, batches=1, prune=prune)
        same = None
        if ref is None:
            ref = recs.clone()
        else:
            same = bool(torch.equal(ref, recs))
        row = dict(cfg, sweep_ms=round(ms['score_candidates'], 4), rescore_ms=round(ms['rescore_topk'], 4), pass_ms=round(pass_ms, 4),
                   swept=round(stats['tiles_scored'] / stats['tiles_total'], 4), flagged=stats['flagged_users'],
                   chain=stats.get('two_phase', {}).get('chain_quantiles', stats.get('exit_tile_quantiles')), same_lists=same)
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


if __name__ == '__main__':
    main()
