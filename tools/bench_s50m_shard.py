"""A 1-GPU shard of BASELINE.json configs[4] (synthetic 50M users x 500K items, rank 200, dense scoring + fused
top-50 on 8 GPUs): `--users` of the 50M users (default 1M = 1/50 of the job, 1/6 of one GPU's share), the full
500K-item catalogue, ~50 interactions per user.  Exercises the rank-200 / top-50 instances of the kernels
(100 MFMA steps per tile, 64 candidates per user, 256-wide solver block) at catalogue scale and reports the
same quantities as bench.py.  The item factors come from an SVD build on the shard itself (the config leaves
the build untimed: "V from a planted model"); parity is checked against the CPU oracle on a user sample.

    python tools/bench_s50m_shard.py [--users 1000000] [--steps 3] > profiles/r01_s50m_shard.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    """the measurement as a dict (bench.py's `sub.configs4_s50m_shard` calls this with its own argument object)"""
    from polara_amd.ops import HipOps
    from polara_amd.synth import planted_csr, csr_to_numpy
    from polara_amd.csr import popularity_order
    from polara_amd.solver import svd_topk
    from polara_amd import scoring
    dev = 'cuda:0'
    ops = HipOps(dev)
    t0 = time.perf_counter()
    csr = planted_csr(args.users, args.items, 50, args.rank // 4, levels=5, seed=5, device=dev, min_items=20,
                      max_items=2000, chunk_rows=1024)
    c = csr_to_numpy(csr)
    del csr
    torch.cuda.empty_cache()
    gen_s = time.perf_counter() - t0
    n_users, n_items = c['shape']
    nnz = int(c['indptr'][-1])
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, inv_order = popularity_order(c['indices'], n_items)
    A = ops.csr_relabel_cols(A, rank_of)
    _ = A.T
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, sigma, V, st = svd_topk(ops, A, args.rank)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    order2 = torch.argsort(torch.linalg.vector_norm(V, dim=1), descending=True, stable=True)
    rank2 = torch.empty_like(order2)
    rank2[order2] = torch.arange(n_items, device=order2.device)
    V = V[order2].contiguous()
    A = ops.csr_relabel_cols(A, rank2, sort=False)
    F = scoring.FactorImage(ops, V)
    for _ in range(2):                                           # warm-up: the first pass of a process on a fresh box has been
        recs = scoring.recommend(ops, F, A, args.topk, True)   # seen 7 ms slower than the following ones
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs = scoring.recommend(ops, F, A, args.topk, True)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / args.steps
    ops.timers = {}
    stats = {}
    scoring.recommend(ops, F, A, args.topk, True, stats=stats)
    torch.cuda.synchronize()
    ms = {k: float(np.mean([a.elapsed_time(b) for a, b, _ in v])) for k, v in ops.timers.items()}
    ops.timers = None
    swept = stats['tiles_scored'] / max(stats['tiles_total'], 1)
    flops = 2.0 * n_users * n_items * args.rank
    # CPU oracle on a sample (external item ids)
    n_chk = min(args.check_users, n_users)
    cpu_oracle = None
    if n_chk > 0:
        # CPU reference path on a user sample (external item ids): the only use of oracle/ here, as the checker
        from oracle import polara_oracle as orc
        p1 = int(c['indptr'][n_chk])
        td = (np.repeat(np.arange(n_chk), np.diff(c['indptr'][:n_chk + 1])), c['indices'][:p1].astype(np.int64),
              c['values'][:p1].astype(np.float64))
        o2 = ops.to_host(order2)
        back = np.empty_like(o2)
        back[o2] = np.arange(n_items)
        V_ext = np.ascontiguousarray(ops.to_host(V)[back][rank_of])
        t0 = time.perf_counter()
        ref = orc.svd_recommendations(V_ext, td, (n_chk, n_items), args.topk, filter_seen=True)
        cpu_s = time.perf_counter() - t0
        got = inv_order[o2[ops.to_host(recs[:n_chk])]]
        cpu_oracle = {'users': n_chk, 'seconds': cpu_s, 'users_per_s': n_chk / cpu_s,
                      'identical_rows': float((got == ref).all(axis=1).mean())}
    out = {
        'workload': 'shard of BASELINE.json configs[4]: %d of 50M users x %d items, rank %d, top-%d, ~50 nnz/user'
                    % (n_users, n_items, args.rank, args.topk),
        'n_users': n_users, 'n_items': n_items, 'nnz': nnz, 'gen_s': gen_s,
        'users_per_s': n_users / step_s, 'ms_per_pass': 1e3 * step_s,
        'projected_8gpu_50M_users_s': 50e6 / 8 / (n_users / step_s),
        'build_s': build_s, 'build': {k: st[k] for k in ('gramian_steps', 'outer', 'block', 'converged')},
        'kernel_ms': ms, 'swept_fraction': swept, 'exit_tile_quantiles': stats.get('exit_tile_quantiles'),
        'flagged_users': stats['flagged_users'], 'refolded_users': stats.get('refolded_users'), 'candidate_capacity': stats['candidate_capacity'],
        'sweep_TFLOPs_executed': flops * swept / (ms['score_candidates'] * 1e-3) / 1e12,
        'sweep_TFLOPs_dense_equivalent': flops / (ms['score_candidates'] * 1e-3) / 1e12,
        'cpu_oracle': cpu_oracle,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--users', type=int, default=1_000_000)
    ap.add_argument('--items', type=int, default=500_000)
    ap.add_argument('--rank', type=int, default=200)
    ap.add_argument('--topk', type=int, default=50)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--check-users', type=int, default=300)
    print(json.dumps(run(ap.parse_args())))


if __name__ == '__main__':
    main()
