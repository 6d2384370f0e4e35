"""bench.py — users scored/sec (+ SVD build time) of the PureSVD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the largest single-GPU config): synthetic planted 1M users x
100K items, ~1e8 nnz CSR, PureSVD rank 50, top-10, every user scored.  A "step" is one full
`get_recommendations` pass over all users (polara_amd.scoring.recommend, the path the model classes
run): fold-in SpMM (against the fp32 image of V, certified) -> fused MFMA score/mask/top-k sweep with
exact norm-bound pruning -> exact fp64 re-scoring + certification -> exact re-do of the uncertified
users.  The SVD build (the other half of the metric) runs once before the timed region (after one
untimed warm-up build) and is reported as `build_s` — including the re-indexing of the catalogue by
factor norm that the scoring passes use — with its own roofline.
Users are sharded over ranks (strong scaling: the total work is fixed); the build's only
collective is the all-reduce of the Gramian-step block, scoring has none.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel of the timed region
(score_candidates, MFMA-bound): `achieved`/`frac` on the MFMA flops actually executed (the pruned share of
the reference's 2*n_users*n_items*rank), the dense-equivalent rate next to it; per-kernel durations come
from five extra untimed passes with HIP events.  `roofline_build` describes the SpMM (HBM-bound; algorithmic
bytes per launch = nnz*(4+val_bytes) + 8*(n_rows+1) + nc*(x_bytes*n_cols + 8*n_rows)).
`cpu_baseline` is the oracle (= the reference's SciPy/NumPy path restated) timed on this box's host
cores on a bounded sample; the GPU lists of that sample are compared with it row by row.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='s1m', choices=['s1m', 'ml20m', 'ml1m'])
    ap.add_argument('--no-prune', action='store_true',
                    help='score every item tile for every user (disables the exact norm-bound pruning of the sweep)')
    ap.add_argument('--no-norm-order', action='store_true',
                    help='keep the popularity item order for scoring (default: re-index the catalogue by descending '
                         'factor norm after the build; the re-indexing time is part of build_s)')
    ap.add_argument('--batches', type=int, default=0,
                    help='user batches per scoring pass, round-robin on two HIP streams (0 = auto: one batch per 4M users)')
    ap.add_argument('--scale', type=float, default=1.0, help='shrink users/items (debug only; invalidates the number)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--item-order', default='popularity', choices=['popularity', 'natural'],
                    help='internal item order of the device path (models.py does the same relabelling)')
    ap.add_argument('--cpu-users', type=int, default=0, help='users in the CPU scoring sample (0 = auto)')
    return ap.parse_args()


def spmm_alg_bytes(meta):
    n_rows, n_cols, nnz, nc, vbytes, xbytes = meta     # xbytes: element size of the dense input block
    return nnz * (4 + vbytes) + 8 * (n_rows + 1) + nc * (xbytes * n_cols + 8 * n_rows)


def pmc_traffic():
    """HBM/fabric bytes per launch of the two dominant kernels, from the committed rocprofv3 --pmc
    passes of this same command (profiles/r01_bench_pmc_{fetch,write}_size.txt; separate passes, as
    the tool requires).  FETCH_SIZE is doubled: both kernels read with 16 B/lane loads, which gfx950
    tallies at half their size (128 B requests counted as 64 B, MI355X_MICROARCH.md §HBM; calibrated
    there for streaming reads, assumed for the 256 B gather pieces of the SpMM).  Returns {} when the
    summaries are not there."""
    out = {}
    try:
        def per_launch(fn, kernel):
            for line in open(os.path.join(ROOT, 'profiles', fn)):
                if kernel in line and ('FETCH_SIZE' in line or 'WRITE_SIZE' in line):
                    return float(line.split()[-1]) * 1024.0     # KB -> bytes
            return None
        f_s, w_s = per_launch('r01_bench_pmc_fetch_size.txt', 'score_candidates_kernel'), per_launch('r01_bench_pmc_write_size.txt', 'score_candidates_kernel')
        f_m, w_m = per_launch('r01_bench_pmc_fetch_size.txt', 'spmm_csr_groups_kernel'), per_launch('r01_bench_pmc_write_size.txt', 'spmm_csr_groups_kernel')
        if f_s is not None and w_s is not None:
            out['score'] = 2.0 * f_s + w_s
        if f_m is not None and w_m is not None:
            out['spmm'] = 2.0 * f_m + w_m
    except OSError:
        pass
    return out


def events_ms(pairs):
    return [e0.elapsed_time(e1) for e0, e1, _ in pairs]


def cpu_baseline(c, V_host, rank, topk, n_score_users, build_rows):
    """The oracle on host cores: (a) reference scoring path on the first `n_score_users` users
    (chunked exactly like utils.py:16-53), (b) scipy svds on the first `build_rows` users."""
    import scipy.sparse as sps
    from oracle import polara_oracle as orc     # checker / CPU baseline only
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    indptr, indices, values = c['indptr'], c['indices'], c['values']
    n_items = c['shape'][1]
    # (a) scoring
    hi = int(indptr[n_score_users])
    users = np.repeat(np.arange(n_score_users, dtype=np.int64), np.diff(indptr[:n_score_users + 1]))
    td = (users, indices[:hi].astype(np.int64), values[:hi].astype(np.float64))
    t0 = time.perf_counter()
    recs = orc.svd_recommendations(V_host, td, (n_score_users, n_items), topk, True)
    t_score = time.perf_counter() - t0
    # (b) build on a row sample
    hb = int(indptr[build_rows])
    A = sps.csr_matrix((values[:hb].astype(np.float64), indices[:hb], indptr[:build_rows + 1]),
                       shape=(build_rows, n_items))
    np.random.seed(0)
    t0 = time.perf_counter()
    orc.svd_build(A, rank)
    t_build = time.perf_counter() - t0
    return dict(value=n_score_users / t_score, unit='users/s', cores=int(blas_threads), kind='port',
                sample='reference scoring path (chunked GEMM + downvote + per-row argpartition, fp64) on the first '
                       '%d users of the same matrix with the GPU-built V; svds build timed on the first %d users '
                       '(%d nnz)' % (n_score_users, build_rows, hb),
                score_sample_s=t_score, build_sample_s=t_build, build_sample_users=build_rows,
                build_sample_nnz=hb, host_cpus=os.cpu_count()), recs


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    from polara_amd.dist import init_from_env
    from polara_amd.solver import NoComm
    debug_backend = os.environ.get('PK_BENCH_DEBUG_BACKEND')   # e.g. 'gloo': N ranks sharing ONE GPU (path check only)
    if world > 1:
        if debug_backend:
            os.environ['LOCAL_RANK'] = '0'
        comm = init_from_env(backend=debug_backend)
    else:
        torch.cuda.set_device(0)
        comm = NoComm()
    rank_id = comm.rank
    from polara_amd.ops import HipOps
    from polara_amd.synth import make_workload, csr_to_numpy
    from polara_amd.csr import nnz_balanced_row_partition
    from polara_amd.solver import svd_topk
    from polara_amd import scoring

    dev = 'cuda:%d' % torch.cuda.current_device()
    ops = HipOps(dev)
    t_gen = time.perf_counter()
    csr, cfg = make_workload(args.workload, device=dev, scale=args.scale)
    c = csr_to_numpy(csr)
    del csr
    torch.cuda.empty_cache()
    t_gen = time.perf_counter() - t_gen
    n_users, n_items = c['shape']
    nnz = int(c['indptr'][-1])
    rank, topk = cfg['rank'], cfg['topk']

    # ---- shard users (nnz-balanced contiguous blocks) -------------------------------------------------
    bounds = nnz_balanced_row_partition(c['indptr'], comm.world)
    lo, hi = int(bounds[rank_id]), int(bounds[rank_id + 1])
    sub = slice(int(c['indptr'][lo]), int(c['indptr'][hi]))
    A = ops.csr(c['indptr'][lo:hi + 1] - c['indptr'][lo], c['indices'][sub], c['values'][sub], (hi - lo, n_items))
    inv_order = None
    if args.item_order == 'popularity':
        # internal item order = descending popularity over the WHOLE matrix (identical on every rank);
        # results are mapped back to the external ids below.  Part of data ingest, like the CSC image.
        from polara_amd.csr import popularity_order
        rank_of, inv_order = popularity_order(c['indices'], n_items)
        A = ops.csr_relabel_cols(A, rank_of)
    _ = A.T   # CSC image built once (device transpose), outside any timing

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
        torch.cuda.synchronize()

    # ---- SVD build (once) -------------------------------------------------------------------------------
    if args.warmup > 0:
        # one untimed build: the first heavy GPU work of a process also pays one-off allocator / page-table
        # set-up costs (~80 ms on a fresh box) that are not part of the solver
        _, _, Vw, _ = svd_topk(ops, A, rank, comm=comm)
        ow = torch.argsort(torch.linalg.vector_norm(Vw, dim=1), descending=True, stable=True)   # same for the
        Vw = Vw[ow].contiguous()                                        # re-indexing helpers (first-use loads)
        del Vw, ow
    ops.timers = {}
    import gc
    gc.collect()                      # a pending collection of the data-generation garbage otherwise lands in
    gc_was = gc.isenabled()           # the timed build now and then (+35 ms, bimodal build times)
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    _, sigma, V, bstats = svd_topk(ops, A, rank, comm=comm)
    barrier()
    build_s = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    spmm_ev = ops.timers.get('spmm', [])
    spmm_ms = events_ms(spmm_ev)
    spmm_bytes = [spmm_alg_bytes(m) for _, _, m in spmm_ev]
    ops.timers = None
    # ---- serving index: catalogue re-indexed by descending factor norm -------------------------------------
    # The pruning bound of the sweep is a suffix maximum of the item-factor norms: it is tightest when the
    # items are visited in descending norm.  Norms are only known after the build, so the factors and the
    # matrix of the users to score are relabelled once here (a device sort of the nnz; row pointers and
    # task plan unchanged); this is model-dependent preparation and is charged to build_s.
    order2 = None
    reindex_s = 0.0
    A_score = A
    if not args.no_norm_order:
        barrier()
        t0 = time.perf_counter()
        vn = torch.linalg.vector_norm(V, dim=1)
        order2 = torch.argsort(vn, descending=True, stable=True)          # new internal id -> old internal id
        rank2 = torch.empty_like(order2)
        rank2[order2] = torch.arange(n_items, device=order2.device)
        V = V[order2].contiguous()
        A_score = ops.csr_relabel_cols(A, rank2, sort=False)   # renaming only: nothing downstream needs ordered rows
        del A
        barrier()
        reindex_s = time.perf_counter() - t0
        build_s += reindex_s
    F = scoring.FactorImage(ops, V)
    A = A_score

    # ---- timed region: K full scoring passes ------------------------------------------------------------
    kw = dict(prune=not args.no_prune, batches=args.batches or None)
    for _ in range(args.warmup):
        scoring.recommend(ops, F, A, topk, True, **kw)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    if not os.environ.get('PK_BENCH_KEEP_GC'):
        gc.disable()                      # no collector pauses inside the timed region
    barrier()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        recs = scoring.recommend(ops, F, A, topk, True, **kw)
        if os.environ.get('PK_BENCH_STEP_TIMES'):   # debugging aid: recommend() ends with a host sync anyway
            step_marks.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if step_marks:
        print('step ms:', ' '.join('%.2f' % (1e3 * (b_ - a_)) for a_, b_ in zip([t0] + step_marks[:-1], step_marks)),
              file=sys.stderr)
    # ---- untimed instrumented passes: one batch, HIP events around every kernel (the durations the roofline
    # is computed from are those of the kernels running ALONE, not overlapped with another batch's fold-in).
    # Five passes back to back in the rhythm of the timed loop — no statistics in between: the host work of
    # reading them leaves the GPU idle long enough for its clocks to drop, and the next pass's kernels then time
    # 4-8 % slower than in the timed loop — then one more pass for the sweep statistics.
    ops.timers = {}
    prof = None
    if os.environ.get('PK_SCORE_PROFILE'):   # tuning builds only (polara_amd/build_native.py)
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        ops.lib.pk_debug_profile(None, 1)
    for _ in range(5):
        scoring.recommend(ops, F, A, topk, True, prune=not args.no_prune, batches=1)
    torch.cuda.synchronize()
    if os.environ.get('PK_SCORE_PROFILE'):
        ops.lib.pk_debug_profile(buf, 0)
        prof = dict(zip(('kernel', 'flush', 'walk', 'push_incl_flush', 'prologue', 'epilogue', 'n_flush', 'tiles'), list(buf)))
        print('PK_SCORE_PROFILE', prof, file=sys.stderr)
    cand_ms = events_ms(ops.timers.get('score_candidates', []))
    fold_ms = events_ms(ops.timers.get('spmm', []))
    ops.timers = None
    stats = {}
    scoring.recommend(ops, F, A, topk, True, stats=stats, prune=not args.no_prune, batches=1)
    torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if debug_backend == 'gloo' else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_users / (elapsed / args.steps)

    if rank_id != 0:
        return
    cand_avg_ms = float(np.mean(cand_ms))
    traffic = pmc_traffic() if (args.workload == 's1m' and args.scale == 1.0 and comm.world == 1) else {}
    n_chunk_launches = int(ops.lib.pk_score_chunk_launches(n_items, rank, stats.get('item_splits', 1), 0, 0 if args.no_prune else 1))
    flops = 2.0 * (hi - lo) * n_items * rank                     # the reference's dense contraction (models.py:860)
    swept = stats['tiles_scored'] / max(stats['tiles_total'], 1)    # share of the (user group x item tile) grid scored
    flops_exec = flops * swept
    achieved_tf = flops_exec / (cand_avg_ms * 1e-3) / 1e12
    out = {
        'metric': 'users scored/sec + SVD build time', 'value': value, 'unit': 'users/s',
        'n_gpus': comm.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'dtype_detail': 'f32 MFMA candidate scoring; fold-in gathers fl32(V) with f64 accumulation, its rounding is part of the '
                        'certification (uncertified users are re-folded in f64); f64 exact re-scoring and SVD build',
        'data': 'synthetic (planted low-rank + Zipf popularity, seeded; generated on GPU)',
        'config': {'workload': {'s1m': 'Synthetic 1M users x 100K items, ~0.1% density CSR, PureSVD rank=50, top-10, all users scored (BASELINE.json configs[1])',
                                'ml20m': 'ML-20M-shaped synthetic 138493 x 26744, PureSVD rank=100, top-20 (BASELINE.json configs[2])',
                                'ml1m': 'ML-1M-shaped synthetic 6040 x 3706, PureSVD rank=10, top-10 (BASELINE.json configs[0])'}[args.workload],
                   'n_users': n_users, 'n_items': n_items, 'nnz': nnz, 'rank': rank, 'topk': topk,
                   'parallelism': 'users sharded over %d GPU(s); Gramian all-reduce in build only' % comm.world,
                   'scale': args.scale, 'item_order': args.item_order, 'prune': not args.no_prune, 'score_order': 'popularity' if args.no_norm_order else 'factor norm',
                   'batches': args.batches or 'auto'},
        'build_s': build_s, 'reindex_s': reindex_s,
        'build': {'gramian_steps': bstats['gramian_steps'], 'outer_iterations': bstats['outer'],
                  'block': bstats['block'], 'converged': bstats['converged'], 'spmm_launches': len(spmm_ms),
                  'sigma_max': float(sigma[0].item()), 'sigma_min': float(sigma[-1].item())},
        'score': {'fold_in_ms': float(np.mean(fold_ms)) if fold_ms else None, 'candidates_ms': cand_avg_ms,
                  'flagged_users_last_step': stats.get('flagged_users'), 'refolded_users_last_step': stats.get('refolded_users'),
                  'candidate_capacity': stats.get('candidate_capacity')},
        'roofline': {'kernel': 'score_candidates_kernel', 'bound': 'mfma', 'achieved': achieved_tf,
                     'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved_tf / PEAK_FP32_MFMA_TFLOPS,
                     'traffic': (traffic['score'] * n_chunk_launches if 'score' in traffic else None),
                     'traffic_note': 'HBM/fabric bytes per scoring pass = item-chunk launches x (2*FETCH_SIZE + WRITE_SIZE) '
                                     'of a separate rocprofv3 --pmc run (profiles/r01_bench_pmc_*.txt); algorithmic minimum ~1 GB',
                     'launches': len(cand_ms), 'kernel_launches_per_pass': n_chunk_launches, 'avg_ms': cand_avg_ms, 'flop_per_launch': flops_exec,
                     'swept_fraction': swept, 'exit_tile_quantiles': stats.get('exit_tile_quantiles'),
                     'n_tiles': -(-n_items // 32),
                     'note': 'achieved/frac count only the MFMA tiles actually scored: the sweep is pruned exactly '
                             '(Cauchy-Schwarz bound, identical results; --no-prune scores every tile)',
                     'dense_equivalent': {'flop_per_launch': flops, 'TFLOP/s': flops / (cand_avg_ms * 1e-3) / 1e12,
                                          'frac_of_peak': flops / (cand_avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}},
        'roofline_build': {'kernel': 'spmm_csr_groups_kernel', 'bound': 'hbm',
                           'achieved': float(sum(spmm_bytes) / (sum(spmm_ms) * 1e-3) / 1e9) if spmm_ms else None,
                           'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                           'frac': float(sum(spmm_bytes) / (sum(spmm_ms) * 1e-3) / 1e9 / PEAK_HBM_GBPS) if spmm_ms else None,
                           'traffic': traffic.get('spmm'),
                           'traffic_note': '2*FETCH_SIZE + WRITE_SIZE per launch (profiles/r01_bench_pmc_*.txt): ~20x the '
                                           'algorithmic bytes - the 400-512 B row gathers of the dense block are not '
                                           'read once but once per nnz, and half of them miss L2/MALL',
                           'gather_GBps': (float(sum(m[2] * m[3] * float(m[5]) for _, _, m in spmm_ev) / (sum(spmm_ms) * 1e-3) / 1e9)
                                           if spmm_ms else None),
                           'gather_note': 'nnz*nc*8 bytes of dense-row gathers per launch / time: the traffic that actually '
                                          'bounds this kernel (profiles/r01_spmm_probe.json)',
                           'launches': len(spmm_ms), 'total_ms': float(sum(spmm_ms)),
                           'bytes_total': float(sum(spmm_bytes))},
        'gen_s': t_gen,
    }
    if not args.no_cpu_baseline and comm.world == 1:
        n_score = args.cpu_users or min(n_users, 20000)   # ~16 chunks of the reference's 1 GB rule on S-1M, ~10 s
        build_rows = min(n_users, max(1000, int(5e6 / max(nnz / n_users, 1))))
        V_ext = ops.to_host(V)
        if order2 is not None:
            o2 = ops.to_host(order2)
            back = np.empty_like(o2)
            back[o2] = np.arange(n_items)
            V_ext = V_ext[back]                                      # norm order -> popularity (build) order
        if inv_order is not None:
            V_ext = V_ext[rank_of]                                   # external item j = internal row rank_of[j]
        base, cpu_recs = cpu_baseline(c, np.ascontiguousarray(V_ext), rank, topk, n_score, build_rows)
        gpu_recs = ops.to_host(recs[:n_score])
        if order2 is not None:
            gpu_recs = ops.to_host(order2)[gpu_recs]                 # norm order -> popularity (build) order
        if inv_order is not None:
            gpu_recs = inv_order[gpu_recs].astype(np.int64)          # internal -> external item ids
        same = float((gpu_recs == cpu_recs).all(axis=1).mean())
        base['gpu_vs_cpu_identical_rows'] = same
        base['speedup_scoring'] = value / base['value']
        out['cpu_baseline'] = base
    print(json.dumps(out))


if __name__ == '__main__':
    main()
