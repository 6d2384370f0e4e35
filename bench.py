"""bench.py — users scored/sec + SVD build time of the PureSVD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

HEADLINE (top-level `value`, `config`, `roofline`, `cpu_baseline`): the configuration BASELINE.json's `metric`
is quoted on — "ML-20M rank-50 PureSVD" — as an ML-20M-shaped synthetic matrix (138 493 x 26 744, ~2e7 nnz, 10
rating levels), rank 50, top-10, every user scored.  A "step" is one full `get_recommendations` pass over all
users (polara_amd.scoring.recommend, the path the model classes run): fold-in SpMM (against the fp32 image of V,
certified) -> fused MFMA score / seen-mask / top-k sweep with exact norm-bound pruning -> exact fp64 re-scoring
+ certification -> exact re-do of the uncertified users -> the [n_users x topk] int64 result copied to a pinned
HOST buffer (the reference returns a host array).  The copy runs on a side stream, double-buffered, so that pass
i+1 computes while pass i's result travels; the timed region ends when the last result is on the host
(`latency_ms_per_pass` is the un-pipelined figure: pass + copy + sync).

The other half of the metric, `build_s`, is the WHOLE warm build as the model layer does it — upload of the host
CSR, relabelling into popularity order, CSC image, eigensolver, re-indexing of catalogue and test rows by factor
norm, factor images — itemised in `build`, with the first (cold) build of the process next to it.

Sub-blocks (N = 1 only, `--only-headline` skips them), each with its own ms_per_step / swept fraction / rows
identical to the CPU path:
  configs2_ml20m_rank100_top20   BASELINE.json configs[2]
  configs1_s1m_rank50_top10      BASELINE.json configs[1] (1M x 100K, 1e8 nnz), pruned and --no-prune sweeps
  flat_norm_catalogue            the headline matrix with every item-factor row scaled to unit norm: the pruning
                                 bound never fires, the sweep scores 100 % of the tiles (adversarial catalogue)
  model_path                     SVDModel(ArrayData).build() + .get_recommendations() through the plugin surface
                                 (host triplets in, host int64 array out), end to end

Users are sharded over ranks (strong scaling: the total work is fixed); the build's only collective is the
all-reduce of the Gramian-step block, scoring has none.

`roofline` describes the dominant kernel of the timed region (score_candidates, MFMA-bound): achieved/frac on the
MFMA flops actually executed (the pruned share of the reference's 2*n_users*n_items*rank), dense-equivalent rate
next to it; kernel durations from five extra untimed passes with HIP events on the launch stream.
`roofline_build` describes the SpMM (HBM-bound; algorithmic bytes per launch = nnz*(4+val_bytes) + 8*(n_rows+1) +
nc*(x_bytes*n_cols + 8*n_rows)).  `traffic` comes from the rocprofv3 --pmc summaries of this command committed
under profiles/ (separate passes; the summary names the commit it was taken at).
`cpu_baseline` is the oracle (= the reference's SciPy/NumPy path restated) timed on this box's host cores on a
bounded sample; the GPU lists of that sample are compared with it row by row.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The timed loop alternates consecutive passes between two HIP streams next to a copy stream (and the build runs its
# convergence monitors on a fourth).  The HIP runtime maps streams round-robin onto GPU_MAX_HW_QUEUES hardware queues —
# four by default — and two streams that share a queue do not overlap at all: with the default the pipelined loop measured
# 0.81 ms per step on this round's boxes, with eight queues 0.68 (165 -> 203 M users/s; the serial loop 0.845 either way).
# Must be in the environment before the runtime initialises; a caller's own setting wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense (= the fp32 vector rate)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (2495 measured)
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec

WORKLOAD_TEXT = {
    'ml20m': 'ML-20M-shaped synthetic 138493 x 26744, ~2e7 nnz, 10 levels (BASELINE.json metric / configs[2])',
    's1m': 'Synthetic 1M users x 100K items, ~0.1% density CSR (BASELINE.json configs[1])',
    'ml1m': 'ML-1M-shaped synthetic 6040 x 3706 (BASELINE.json configs[0])',
}


def log(msg):
    print('[bench] ' + msg, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='ml20m', choices=['s1m', 'ml20m', 'ml1m'])
    ap.add_argument('--rank', type=int, default=0, help='0 = the headline rank of the workload (ml20m: 50, the metric line)')
    ap.add_argument('--topk', type=int, default=0)
    ap.add_argument('--only-headline', action='store_true',
                    help='skip the three adversarial-catalogue rows that ride in the compact line (always skipped for N > 1)')
    ap.add_argument('--detail', action='store_true',
                    help='also measure the other BASELINE.json configurations, the plugin surface and the coarse C ABI (N = 1); '
                         'they go to bench_detail.json, never to the stdout line')
    ap.add_argument('--no-prune', action='store_true',
                    help='score every item tile for every user (disables the exact norm-bound pruning of the sweep)')
    ap.add_argument('--no-norm-order', action='store_true',
                    help='keep the popularity item order for scoring (default: re-index the catalogue by descending '
                         'factor norm after the build; the re-indexing time is part of build_s)')
    ap.add_argument('--batches', type=int, default=0,
                    help='user batches per scoring pass, round-robin on two HIP streams (0 = auto: one batch per 4M users)')
    ap.add_argument('--scale', type=float, default=1.0, help='shrink users/items (debug only; invalidates the number)')
    ap.add_argument('--no-graph', dest='graph', action='store_false',
                    help='launch every pass kernel by kernel from Python, no questions asked.  Default: the pass over this '
                         'fixed user set is also captured once in a hipGraph (scoring.CapturedPass; the replay is checked '
                         'against a launched pass), both ways of launching are timed during warm-up and the timed region runs '
                         'the faster one (config.launch says which; warmup_calibration_ms_per_step holds both).  Since the '
                         'split-bf16 sweep an ML-20M-shaped pass is ~0.9 ms of kernels: ~14 Python launches cost 0.9-1.3 ms '
                         'of host time depending on the box, a replay ~10 us per graph node: 1.03-1.17 ms')
    ap.set_defaults(graph=True)
    ap.add_argument('--replay-streams', type=int, default=4,
                    help='streams the replayed form of a short pass alternates between (scoring.RecordedPass; default 4)')
    ap.add_argument('--pass-streams', type=int, default=2,
                    help='HIP streams consecutive scoring passes alternate between (python launches; 1 = strictly serial passes)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--warm', default='code', choices=['code', 'pipeline', 'none'],
                    help="what HipOps() does once per process: load the library's code objects (default), also run the miniature pipeline, or nothing")
    ap.add_argument('--cpu-users', type=int, default=0, help='users in the CPU scoring sample (0 = auto)')
    return ap.parse_args()


MAX_PASS_STREAMS = 6      # pass streams + copy stream + the solver's monitor stream stay within the 8 hardware queues asked for


def ensure_world(args):
    """`--gpus N` means N ranks, one per GPU, over RCCL — whoever starts the script.  Under torchrun (WORLD_SIZE set) the
    flag must agree with the launcher; started bare with N > 1 this process becomes the launcher: it checks that N GPUs
    are visible (fails loudly otherwise — a silent one-GPU run would hand the driver a flat scaling curve) and re-executes
    itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 …`, passing every
    flag through; its exit code is the job's.  PK_BENCH_DEBUG_BACKEND=gloo (N ranks sharing ONE GPU, a path check, never
    a number) lifts the GPU-count requirement."""
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is not None:
        if int(env_world) != int(args.gpus):
            raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; they must agree'
                             % (args.gpus, env_world))
        return
    if args.gpus <= 1:
        return
    debug = os.environ.get('PK_BENCH_DEBUG_BACKEND')
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < args.gpus and not debug:
        raise SystemExit('bench.py: --gpus %d needs %d visible GPUs (one rank per GPU over RCCL), this box shows %d; '
                         'nothing was measured' % (args.gpus, args.gpus, visible))
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('--gpus %d without a launcher: re-executing as %s' % (args.gpus, ' '.join(cmd[1:9])))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def spmm_alg_bytes(meta):
    n_rows, n_cols, nnz, nc, vbytes, xbytes = meta     # xbytes: element size of the dense input block
    return nnz * (4 + vbytes) + 8 * (n_rows + 1) + nc * (xbytes * n_cols + 8 * n_rows)


def pmc_traffic(tag):
    """HBM/fabric bytes of the two dominant kernels from the committed rocprofv3 --pmc passes of this command
    (profiles/r03_<tag>_pmc_{fetch,write}_size.txt, else round 2's; separate passes, as the tool requires; the files carry the
    commit they were taken at).  FETCH_SIZE is doubled: both kernels read with 16 B/lane loads, which gfx950 tallies
    at half their size (MI355X_MICROARCH.md §HBM).  Returns {} when the summaries are not there.
      score       bytes per launch of score_candidates_kernel
      spmm_total  bytes of ALL build SpMM launches of the profiled run (fp64 dense block: every instance of
                  spmm_csr_groups_kernel<.., double, ..> and of the column-per-lane spmm_csr_kernel); the profiled
                  command builds twice (cold + warm), the caller divides by its product count"""
    out = {}
    hashes = {}
    try:
        def rows(fn):
            commit, found = None, []
            for line in open(os.path.join(ROOT, 'profiles', fn)):
                if line.startswith('# commit'):
                    commit = line.split()[-1]
                if line.startswith('# sha256'):
                    hashes[line.split()[2]] = line.split()[3]
                if 'FETCH_SIZE' in line or 'WRITE_SIZE' in line:
                    parts = line.split()
                    found.append((line, int(parts[-3]), float(parts[-2]) * 1024.0, float(parts[-1]) * 1024.0))   # launches, sum, per launch (KB -> B)
            return commit, found
        rnd = next((r for r in ('r06', 'r05', 'r04', 'r03', 'r02') if os.path.exists(os.path.join(ROOT, 'profiles', '%s_%s_pmc_fetch_size.txt' % (r, tag)))), 'r02')
        out['round'] = rnd
        cf, fetch = rows('%s_%s_pmc_fetch_size.txt' % (rnd, tag))
        _, write = rows('%s_%s_pmc_write_size.txt' % (rnd, tag))
        pick = lambda table, pred: [r for r in table if pred(r[0])]
        # several instances of a kernel family appear in a run (the operator set's warm-up launches rank-4 ones): the one that
        # carries the workload is the one with the largest total
        top = lambda rows_: max(rows_, key=lambda r: r[2])
        sc_f, sc_w = pick(fetch, lambda l: 'score_candidates_kernel' in l), pick(write, lambda l: 'score_candidates_kernel' in l)
        if sc_f and sc_w:
            out['score'] = 2.0 * top(sc_f)[3] + top(sc_w)[3]
        is_build_spmm = lambda l: ('spmm_csr_kernel<' in l) or ('spmm_csr_groups_kernel<' in l and 'double' in l)
        bf, bw = pick(fetch, is_build_spmm), pick(write, is_build_spmm)
        if bf and bw:
            out['spmm_total'] = 2.0 * sum(r[2] for r in bf) + sum(r[2] for r in bw)
        is_fold = lambda l: 'fold_q20_kernel<' in l
        ff, fw = pick(fetch, is_fold), pick(write, is_fold)
        if ff and fw:
            out['fold'] = 2.0 * top(ff)[3] + top(fw)[3]      # per launch
        out['commit'] = cf
        # a profile describes the kernels of the tree it was taken in: compare the hashes it carries with the sources
        # here.  No hashes (profiles of rounds 2-3) or a different score.hip / spmm.hip: STALE — the counters then do not
        # go into the record (a kernel change without re-profiling must not ship old counters in a
        # fresh-looking line)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from summarize_rocprof import kernel_source_hashes
        now = kernel_source_hashes(ROOT)
        out['stale_score'] = hashes.get('score.hip') != now.get('score.hip')
        out['stale_spmm'] = hashes.get('spmm.hip') != now.get('spmm.hip')
        out['stale_fold'] = hashes.get('foldq.hip') != now.get('foldq.hip')
    except (OSError, ValueError, IndexError):
        pass
    return out


def events_ms(pairs):
    return [e0.elapsed_time(e1) for e0, e1, _ in pairs]


class Bench:
    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        from polara_amd.dist import init_from_env
        from polara_amd.solver import NoComm
        self.debug_backend = os.environ.get('PK_BENCH_DEBUG_BACKEND')   # 'gloo': N ranks sharing ONE GPU (path check only)
        if self.world > 1:
            if self.debug_backend:
                os.environ['LOCAL_RANK'] = '0'
            self.comm = init_from_env(backend=self.debug_backend)
        else:
            torch.cuda.set_device(0)
            self.comm = NoComm()
        from polara_amd.ops import HipOps
        self.dev = 'cuda:%d' % torch.cuda.current_device()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.ops = HipOps(self.dev, warm=False if getattr(args, 'warm', 'code') == 'none' else getattr(args, 'warm', 'code'))     # default: the library's code objects only (pk_warm_up); the first build pays its own first calls
        torch.cuda.synchronize()
        self.ops_create_s = time.perf_counter() - t0
        self.dist_info = {'world': self.world, 'backend': 'none (one process)', 'device': torch.cuda.get_device_name(self.dev)}
        if self.world > 1:
            import torch.distributed as tdist
            be = tdist.get_backend()
            self.dist_info.update(world=tdist.get_world_size(), backend=be)
            if be == 'nccl':
                try:
                    self.dist_info['rccl'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
                except Exception as exc:       # the version call is decoration; the collectives are not
                    self.dist_info['rccl'] = 'unknown (%s)' % type(exc).__name__
        # The streams of the timed loop are taken NOW, one after the other: torch hands out streams from a pool of 32 that
        # HIP maps round-robin onto a handful of hardware queues (4 by default), so two streams created at unrelated
        # moments can share a queue — and then they do not overlap at all (a pass stream on the copy stream's queue cost
        # the pipelined loop its whole gain: 0.83 instead of 0.69 ms per pass in the same process).  Taken back to back the
        # copy stream and the pass streams sit on different queues.
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.pass_streams_all = [torch.cuda.Stream(device=self.dev) for _ in range(max(0, min(int(getattr(args, 'pass_streams', 2)), MAX_PASS_STREAMS)))]
        # the replayed form of a SHORT pass (scoring.RecordedPass) is bound by the host's launches, not by the chip: four streams
        # keep it fed (tools/probes/shard_pass_modes.py: 2 / 4 / 6 streams at a rank's shard of N = 8: 0.180 / 0.147 / 0.275 ms)
        n_rec = max(len(self.pass_streams_all), min(int(getattr(args, 'replay_streams', 4)), MAX_PASS_STREAMS)) if self.pass_streams_all else 0
        self.rec_streams_all = self.pass_streams_all + [torch.cuda.Stream(device=self.dev) for _ in range(n_rec - len(self.pass_streams_all))]

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            self.comm.barrier()
        torch.cuda.synchronize()

    # ---- data ---------------------------------------------------------------------------------------------
    def generate(self, workload, scale=1.0):
        from polara_amd.synth import make_workload, csr_to_numpy, WORKLOADS
        from polara_amd.datasets import find_movielens, load_movielens
        t0 = time.perf_counter()
        real = find_movielens(workload, ROOT) if scale == 1.0 and not os.environ.get('PK_BENCH_SYNTHETIC') else None
        if real:
            # the REAL matrix when somebody put it there (data/ml-20m.zip, data/ml-20m/ratings.csv, ...; there is no
            # network here): same metric, same code path, `data: "real"` in the line
            c = load_movielens(real)
            c = {k: c[k] for k in ('indptr', 'indices', 'values', 'shape')}
            cfg = dict(rank=WORKLOADS[workload]['rank'], topk=WORKLOADS[workload]['topk'], levels=int(len(np.unique(c['values']))))
            c['data'] = 'real (%s)' % os.path.relpath(real, ROOT)
            log('real data: %s  %d x %d, %d ratings' % (real, c['shape'][0], c['shape'][1], len(c['values'])))
        else:
            csr, cfg = make_workload(workload, device=self.dev, scale=scale)
            c = csr_to_numpy(csr)
            del csr
            torch.cuda.empty_cache()
            c['data'] = 'synthetic'
        c['gen_s'] = time.perf_counter() - t0
        c['cfg'] = cfg
        if os.environ.get('PK_BENCH_SORT_USERS'):       # experiment: users ordered by activity before they are grouped by 32
            cnt = np.diff(c['indptr'])
            order = np.argsort(-cnt if os.environ['PK_BENCH_SORT_USERS'] == 'desc' else cnt, kind='stable')
            pos = np.concatenate([np.arange(c['indptr'][r], c['indptr'][r + 1]) for r in order])
            c['indices'], c['values'] = c['indices'][pos], c['values'][pos]
            c['indptr'] = np.r_[0, np.cumsum(cnt[order])].astype(np.int64)
        return c

    # ---- build: everything between "host CSR of my users" and "ready to score" -------------------------------
    def build(self, c, rank, norm_order=True, catalogue='svd'):
        """One complete build as the model layer does it; returns (state, timings).  Every stage is bracketed by a
        device synchronisation (and a barrier across ranks) so that the items add up to the total.
        `catalogue` rescales the rows of the item factors AFTER the solver (adversarial inputs for the sweep's norm-bound
        pruning; the lists are then those of the rescaled factors, checked against the CPU path like any other):
          'svd'    the factors as built (row norms follow the generator's popularity / factor-scale decay)
          'flat'   every row scaled to unit norm: the pruning bound never fires, 100 % of the tiles are scored
          'pop25'  row norms proportional to (item count)^0.25: a slow, real-data-like decay"""
        from polara_amd.csr import nnz_balanced_row_partition
        from polara_amd.solver import svd_topk, prepare_operator
        from polara_amd import scoring
        ops, comm = self.ops, self.comm
        n_users, n_items = c['shape']
        t = {}
        marks = [time.perf_counter()]

        def lap(name):
            self.barrier()
            marks.append(time.perf_counter())
            t[name] = marks[-1] - marks[-2]
        self.barrier()
        marks[0] = time.perf_counter()
        bounds = nnz_balanced_row_partition(c['indptr'], comm.world)
        lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
        sub = slice(int(c['indptr'][lo]), int(c['indptr'][hi]))
        A = ops.csr(c['indptr'][lo:hi + 1] - c['indptr'][lo], c['indices'][sub], c['values'][sub], (hi - lo, n_items))
        lap('upload_s')
        # internal item order = descending popularity over the WHOLE matrix (identical on every rank): per-item counts
        # of the local rows on the device (pk_count_i32), summed over ranks; the renaming itself is one gather
        rank_of, inv_order, counts, rank_dev = ops.item_order(A, comm)     # counts, order and its inverse on the device: one copy back
        A = ops.csr_relabel_cols(A, rank_dev)    # rows re-sorted (pk_csr_relabel_sorted): ascending gathers in every SpMM
        lap('relabel_popularity_s')
        prepare_operator(ops, A, rank, comm=comm)   # the image the products of the build run on: CSC (user-blocked), built on the device (pk_csr_transpose)
        _ = A.plan
        lap('transpose_and_plans_s')
        ops.timers = {}
        counters = ('n_allgather', 'bytes_gathered', 'n_reduce_scatter', 'bytes_scattered', 'n_allreduce', 'bytes_reduced', 'n_panel_exchanges')
        before = {k: getattr(comm, k, 0) for k in counters}
        _, sigma, V, bstats = svd_topk(ops, A, rank, comm=comm)
        lap('solver_s')
        coll = {k: int(getattr(comm, k, 0) - before[k]) for k in counters}
        shard = np.array([[comm.rank, hi - lo, int(c['indptr'][hi] - c['indptr'][lo]), torch.cuda.current_device()]], dtype=np.int64)
        shards = comm.gather_rows(shard, comm.world, 4) if comm.world > 1 else shard
        spmm_ev = ops.timers.get('spmm', [])
        ops.timers = None
        if catalogue != 'svd':
            unit = V / torch.linalg.vector_norm(V, dim=1, keepdim=True).clamp_min(1e-300)
            if catalogue == 'pop25':
                cnt = torch.as_tensor(np.asarray(counts, dtype=np.float64)[inv_order], device=V.device)   # internal order
                unit = unit * cnt.clamp_min(1.0).pow(0.25).unsqueeze(1)
            V = unit.contiguous()
        order2 = None
        A_score = A
        if norm_order:
            # serving index: catalogue re-indexed by descending factor norm (the pruning bound of the sweep is a
            # suffix maximum of these norms); factors and the rows to score are relabelled once
            order32, rank32, V = ops.norm_order(V)          # own radix sort on the norms' bits + one gather (pk_row_norm_order_f64)
            order2 = order32.long()                         # new internal id -> old internal id
            A_score = ops.csr_relabel_cols(A, rank32, sort=True)
        F = scoring.FactorImage(ops, V)
        # the test rows grouped by activity (what the scoring pass sweeps) and their seen-tile streams
        (A_score.by_activity()[0] if A_score.shape[0] >= scoring.ORDER_USERS_MIN else A_score).seen_tiles()
        lap('reindex_and_images_s')
        t['total_s'] = marks[-1] - marks[0]
        state = dict(A=A_score, F=F, V=V, sigma=sigma, order2=order2, rank_of=rank_of, inv_order=inv_order, lo=lo, hi=hi,
                     bstats=bstats, spmm_ev=spmm_ev, collectives=coll, shards=shards)
        del A
        return state, t

    # ---- scoring ---------------------------------------------------------------------------------------------
    def score_passes(self, st, topk, steps, warmup, prune=True, batches=None):
        """Timed region: `steps` full passes, result of every pass copied to a pinned host buffer (double-buffered on a
        side stream).  Returns (seconds max over ranks, last host result, extras)."""
        from polara_amd import scoring
        ops = self.ops
        F, A = st['F'], st['A']
        n_local = A.shape[0]
        DEPTH = 4        # result buffers: the host may run this many passes ahead of the copy engine
        host = [torch.empty((n_local, topk), dtype=torch.int64).pin_memory() for _ in range(DEPTH)]
        done = [torch.cuda.Event() for _ in range(DEPTH)]
        kw = dict(prune=prune, batches=batches)
        main = torch.cuda.current_stream(self.dev)
        cap = stage = None
        self.launch_mode = 'python, kernel by kernel'

        def capture():
            # The pass in a hipGraph.  Only worth trying for SHORT passes: a replay costs the host ~70 us per kernel node on
            # this runtime (1.15-1.3 ms for the 18 nodes of a pass whatever the kernels take), and an instantiated graph
            # was seen to cost the pipelined loop its overlap (0.72 -> 0.80-0.85 ms per pass in the same process)
            nonlocal cap, stage
            try:
                cap = scoring.CapturedPass(ops, F, A, topk, True, prune=prune)
                check = scoring.recommend(ops, F, A, topk, True, prune=prune)
                if not bool((cap.replay() == check).all()):
                    raise RuntimeError('the replayed graph and the launched pass disagree')
                stage = [torch.empty_like(cap.out) for _ in range(DEPTH)]    # the graph rewrites its output buffer every replay
            except Exception as exc:      # a capture problem must not cost the run its number
                log('graph capture failed (%s: %s): launching kernel by kernel' % (type(exc).__name__, exc))
                cap = stage = None
                torch.cuda.synchronize()

        # Consecutive passes alternate between TWO HIP streams (python launches only): pass i + 1's fold-in fills the SIMDs
        # that the tail of pass i's candidate sweep leaves idle (the sweep ends with its longest chains: a handful of
        # waves on an otherwise empty chip).  Every pass still runs completely — its own operands, scratch state per
        # stream, its own result buffer — and the K passes of the timed region all finish inside it; what overlaps is the
        # END of one pass with the BEGINNING of the next, like the result copy already does.  0.79 -> 0.61-0.64 ms per
        # pass on ML-20M-shaped, 4.34 -> 4.13 ms on S-1M (profiles/r03_two_stream_passes_ml20m.txt); `--pass-streams 1`
        # is the strictly serial form, and `latency_ms_per_pass` stays the un-pipelined figure.
        n_ps = max(1, min(int(getattr(self.args, 'pass_streams', 2)), len(self.pass_streams_all)))
        self.pass_streams = self.pass_streams_all[:n_ps] if n_ps > 1 else []

        recorded = []                # scoring.RecordedPass per pass stream (mode 'recorded')

        def one(i, use_cap=True):
            b = i % DEPTH
            src = main
            if use_cap == 'recorded':
                # the pass as its recorded library calls, on the stream it was recorded on; its output buffer is rewritten by
                # the next replay of the same recording, which therefore waits for the copy of this one
                # (its lists go to the host from the pass's LAST KERNEL, which writes the recording's own pinned buffer — mapped
                # memory, ops.scatter_rows — : no copy call, no cross-stream event, no stream switch per pass; DEPTH >= the
                # number of recordings, so the throttle below keeps a buffer from being rewritten before its pass was waited for)
                r = i % len(recorded)
                recs = recorded[r].replay()
                done[b].record(rec_streams[r])
                return recs
            elif cap is not None and use_cap:
                recs = stage[b]
                recs.copy_(cap.replay())       # device copy (microseconds): the D2H of pass i overlaps replay i + 1
            elif self.pass_streams:
                src = self.pass_streams[i % len(self.pass_streams)]
                with torch.cuda.stream(src):
                    recs = scoring.recommend(ops, F, A, topk, True, **kw)
            else:
                recs = scoring.recommend(ops, F, A, topk, True, **kw)
            ready = torch.cuda.Event()
            ready.record(src)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                host[b].copy_(recs, non_blocking=True)
                recs.record_stream(self.copy_stream)
                done[b].record(self.copy_stream)
            return recs
        def loop(n, use_cap):
            r = None
            for ps in (rec_streams if use_cap == 'recorded' else self.pass_streams):
                ps.wait_stream(main)            # whatever the main stream set up is visible to the pass streams
            for i in range(n):
                if i >= DEPTH:
                    done[i % DEPTH].synchronize()   # the buffer about to be overwritten has been consumed (DEPTH passes ago)
                r = one(i, use_cap)
            return r

        # warm-up, untimed as far as `value` goes.  Up to three ways of running the loop are calibrated here and the timed
        # region below runs the fastest, all K steps: 'graph' (replay of the captured pass: ~70 us of host time per kernel
        # node on this runtime), 'serial' (Python launches, passes strictly one after the other on one stream) and
        # 'pipelined' (Python launches, consecutive passes alternating between the pass streams).  Each mode is settled
        # first (the caching allocator's pools are per stream and the result copies defer the reuse of their blocks, so the
        # first dozen passes of a mode still grow the pools: hipMalloc inside a pass).
        all_streams = self.pass_streams

        def set_mode(mode):
            self.pass_streams = all_streams if mode == 'pipelined' else []
            return 'recorded' if mode == 'recorded' else mode == 'graph'

        def record():
            # The pass as a list of recorded library calls per pass stream (scoring.RecordedPass): for SHORT passes, where
            # two passes in flight are bound by the 170 us of host time `recommend` takes to enqueue one (a rank's shard at
            # N >= 4).  Checked against the launched pass like the graph.
            nonlocal recorded, rec_streams
            try:
                rec_streams = list(self.rec_streams_all)[:DEPTH] or [main]      # (one result buffer per recording, DEPTH of them)
                check_host = scoring.recommend(ops, F, A, topk, True, prune=prune).cpu()
                for s_ in rec_streams:
                    s_.wait_stream(main)
                    with torch.cuda.stream(s_):
                        rp = scoring.RecordedPass(ops, F, A, topk, True, prune=prune,
                                                  host_out=torch.empty((n_local, topk), dtype=torch.int64).pin_memory())
                        for _ in range(2):
                            rp.host_out.fill_(-7)
                            got = rp.replay()             # the pinned buffer: the pass's last kernel writes it
                            s_.synchronize()
                            if got is not rp.host_out or not bool((got == check_host).all()):
                                raise RuntimeError('the lists handed to the host by the replayed pass differ from the launched pass')
                    recorded.append(rp)
                torch.cuda.synchronize()
            except Exception as exc:      # a recording problem must not cost the run its number
                log('recording the pass failed (%s: %s): launching through recommend()' % (type(exc).__name__, exc))
                recorded = []
                torch.cuda.synchronize()

        cal = {}
        n_cal = max(10, warmup)

        def calibrate(mode):
            uc = set_mode(mode)
            loop(3 * DEPTH, uc)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loop(n_cal, uc)
            torch.cuda.synchronize()
            cal[mode] = 1e3 * (time.perf_counter() - t1) / n_cal

        set_mode('serial')
        loop(max(warmup, 24), False)        # first passes of the process on this matrix: allocator growth, code objects
        torch.cuda.synchronize()
        modes = ['serial'] + (['pipelined'] if all_streams else [])
        for mode in modes:
            calibrate(mode)
        rec_streams = []
        try_record = self.args.graph and not batches and min(cal.values()) < 0.4     # host-bound territory: ~0.2 ms per pass
        if self.world > 1:
            flag = torch.tensor([1.0 if try_record else 0.0], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            try_record = bool(flag.item() > 0.5)
        if try_record:
            record()
            ok = torch.tensor([1.0 if recorded else 0.0], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            if self.world > 1:
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if ok.item() > 0.5:
                modes.append('recorded')
                calibrate('recorded')
        try_graph = self.args.graph and not batches and cal['serial'] < 0.5
        if self.world > 1:
            flag = torch.tensor([1.0 if try_graph else 0.0], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            try_graph = bool(flag.item() > 0.5)
        if try_graph:
            capture()
            ok = torch.tensor([1.0 if cap is not None else 0.0], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            if self.world > 1:
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if ok.item() > 0.5:
                modes.append('graph')
                calibrate('graph')
        if self.world > 1:            # every rank runs the same mode: the slowest rank's calibration decides
            tt = torch.tensor([cal[m] for m in modes], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            cal = dict(zip(modes, [float(v) for v in tt.tolist()]))
        best = min(modes, key=lambda m: cal[m])
        # the overlap of consecutive passes (0.79 -> 0.61-0.65 ms per pass with the host far ahead of the GPU, 0.72-0.75 ms
        # inside this loop with its result copies and DEPTH-deep throttle; profiles/r03_two_stream_passes_ml20m.txt) needs
        # the pass streams and the copy stream on different hardware queues and no instantiated graph around; it is only
        # taken when the warm-up shows a clear gain over the serial loop
        if best == 'pipelined' and cal['pipelined'] > 0.97 * cal['serial']:
            best = min([m for m in modes if m != 'pipelined'], key=lambda m: cal[m])
        use_cap = set_mode(best)
        self.launch_mode = {'graph': 'hipGraph replay of the captured pass',
                            'recorded': 'recorded library calls of the pass (scoring.RecordedPass), replayed on %d HIP streams' % max(len(rec_streams), 1),
                            'serial': 'python, kernel by kernel, passes one after the other',
                            'pipelined': 'python, kernel by kernel; consecutive passes alternate between %d HIP streams' % len(all_streams)}[best]
        serial_ms = cal.get('serial')
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()                      # no collector pauses inside the timed region
        # the settle passes run LAST before the timed region: a collector run (tens of ms of host time with an idle
        # GPU) between them and the region would hand the first timed passes a chip that has clocked down
        loop(max(warmup, 3 * DEPTH), use_cap)
        torch.cuda.synchronize()
        self.barrier()
        t0 = time.perf_counter()
        recs = loop(steps, use_cap)
        self.barrier()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        # the same loop over a region 10x as long (not `value`): what a pass costs once the fill of the first passes and the
        # drain of the last (its 0.8 ms latency + the 0.4 ms copy of its result) are spread over 200 instead of 20 steps
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loop(10 * steps, use_cap)
        torch.cuda.synchronize()
        long_ms = 1e3 * (time.perf_counter() - t1) / max(10 * steps, 1)
        # un-pipelined latency of one pass: compute + copy + sync
        lat = []
        for i in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            one(i, use_cap)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        if self.world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if self.debug_backend == 'gloo' else self.dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(tt.item())
        last = host[(steps - 1) % DEPTH if steps else 0]
        if use_cap == 'recorded' and steps:
            last = recorded[(steps - 1) % len(recorded)].host_out
        extras = dict(latency_ms_per_pass=1e3 * float(np.median(lat)), d2h_bytes_per_pass=int(n_local * topk * 8),
                      host_result=last, launch=self.launch_mode)
        extras['serial_ms_per_step'] = serial_ms
        extras['long_region_ms_per_step'] = long_ms
        extras['python_launch_ms_per_step'] = cal.get('serial')
        extras['graph_replay_ms_per_step'] = cal.get('graph')
        extras['pipelined_ms_per_step'] = cal.get('pipelined')
        extras['recorded_ms_per_step'] = cal.get('recorded')
        self.pass_streams = all_streams
        return elapsed, recs, extras

    def kernel_times(self, st, topk, prune=True):
        """five untimed instrumented passes back to back (HIP events around every kernel, on the launch stream), then
        one more for the sweep statistics"""
        from polara_amd import scoring
        ops = self.ops
        ops.timers = {}
        for _ in range(5):
            scoring.recommend(ops, st['F'], st['A'], topk, True, prune=prune, batches=1)
        torch.cuda.synchronize()
        ms = {k: float(np.mean(events_ms(v))) for k, v in ops.timers.items()}
        n = {k: len(v) for k, v in ops.timers.items()}
        ops.timers = None
        stats = {}
        scoring.recommend(ops, st['F'], st['A'], topk, True, stats=stats, prune=prune, batches=1)
        torch.cuda.synchronize()
        return ms, n, stats

    # ---- CPU legs ----------------------------------------------------------------------------------------------
    def external_V(self, st):
        ops = self.ops
        V_ext = ops.to_host(st['V'])
        n_items = V_ext.shape[0]
        if st['order2'] is not None:
            o2 = ops.to_host(st['order2'])
            back = np.empty_like(o2)
            back[o2] = np.arange(n_items)
            V_ext = V_ext[back]                                      # norm order -> popularity (build) order
        return np.ascontiguousarray(V_ext[st['rank_of']])            # external item j = internal row rank_of[j]

    def external_ids(self, st, recs_host):
        r = recs_host
        if st['order2'] is not None:
            r = self.ops.to_host(st['order2'])[r]
        return st['inv_order'][r].astype(np.int64)

    def cpu_scoring(self, c, V_ext, topk, n_score_users):
        from oracle import polara_oracle as orc     # checker / CPU baseline only
        indptr, indices, values = c['indptr'], c['indices'], c['values']
        n_items = c['shape'][1]
        hi = int(indptr[n_score_users])
        users = np.repeat(np.arange(n_score_users, dtype=np.int64), np.diff(indptr[:n_score_users + 1]))
        td = (users, indices[:hi].astype(np.int64), values[:hi].astype(np.float64))
        t0 = time.perf_counter()
        recs = orc.svd_recommendations(V_ext, td, (n_score_users, n_items), topk, True)
        return time.perf_counter() - t0, recs

    def cpu_build(self, c, rank, build_rows):
        import scipy.sparse as sps
        from oracle import polara_oracle as orc
        indptr, indices, values = c['indptr'], c['indices'], c['values']
        hb = int(indptr[build_rows])
        A = sps.csr_matrix((values[:hb].astype(np.float64), indices[:hb], indptr[:build_rows + 1]),
                           shape=(build_rows, c['shape'][1]))
        np.random.seed(0)
        t0 = time.perf_counter()
        orc.svd_build(A, rank)
        return time.perf_counter() - t0, hb

    @staticmethod
    def blas_threads():
        try:
            from threadpoolctl import threadpool_info
            return int(max([p.get('num_threads', 1) for p in threadpool_info()] or [1]))
        except Exception:
            return int(os.cpu_count())

    # ---- one complete measurement of (workload, rank, topk) -------------------------------------------------
    def measure(self, c, workload, rank, topk, steps, warmup, prune=True, norm_order=True, cpu=True, cpu_users=0,
                cpu_build=True, catalogue='svd', cold_build=None, cpu_build_whole=False):
        ops, comm = self.ops, self.comm
        n_users, n_items = c['shape']
        nnz = int(c['indptr'][-1])
        st, tb = self.build(c, rank, norm_order, catalogue)           # warm (the process has built before) or cold
        elapsed, recs, extra = self.score_passes(st, topk, steps, warmup, prune=prune, batches=self.args.batches or None)
        ms, n_launch, stats = self.kernel_times(st, topk, prune=prune)
        out = None
        if comm.rank == 0:
            lo, hi = st['lo'], st['hi']
            cand_ms = ms.get('score_candidates')
            flops = 2.0 * (hi - lo) * n_items * rank
            swept = stats['tiles_scored'] / max(stats['tiles_total'], 1)
            tp = stats.get('two_phase')
            if tp:      # head launch + the chunk launches of the seeded splits over the rest of the catalogue
                n_chunk = 1 + int(ops.lib.pk_score_chunk_launches(n_items - 32 * tp['head_tiles'], rank, tp['splits'], 0, 1))
            else:
                n_chunk = int(ops.lib.pk_score_chunk_launches(n_items, rank, stats.get('item_splits', 1), 0, 1 if prune else 0))
            spmm_ms = events_ms(st['spmm_ev'])
            spmm_bytes = [spmm_alg_bytes(m) for _, _, m in st['spmm_ev']]
            bstats = st['bstats']
            out = {
                'value': n_users / (elapsed / steps), 'ms_per_step': 1e3 * elapsed / steps,
                'latency_ms_per_pass': extra['latency_ms_per_pass'], 'd2h_bytes_per_pass': extra['d2h_bytes_per_pass'],
                'ms_per_step_serial': extra.get('serial_ms_per_step'),
                'ms_per_step_long_region': extra.get('long_region_ms_per_step'),
                'workload': '%s, PureSVD rank=%d, top-%d, all users scored' % (
                    WORKLOAD_TEXT[workload] if c.get('data', 'synthetic') == 'synthetic' else
                    'MovieLens ratings from %s, %d x %d' % (c['data'], n_users, n_items), rank, topk),
                'data': c.get('data', 'synthetic'),
                'n_users': n_users, 'n_items': n_items, 'nnz': nnz, 'rank': rank, 'topk': topk, 'prune': prune,
                'score_order': 'factor norm' if norm_order else 'popularity',
                'launch': extra.get('launch', 'python, kernel by kernel'),
                'launch_short': 'hipGraph' if extra.get('launch', '').startswith('hipGraph') else (
                    'recorded calls, %d streams' % len(self.rec_streams_all or [0]) if extra.get('launch', '').startswith('recorded') else (
                        'python, passes on %d streams' % len(self.pass_streams) if 'alternate' in extra.get('launch', '') else 'python')),
                'launches_per_pass': int(sum(n_launch.values()) // 5) if n_launch else None,
                'warmup_calibration_ms_per_step': {'python_launch': extra.get('python_launch_ms_per_step'),
                                                   'python_pipelined': extra.get('pipelined_ms_per_step'),
                                                   'graph_replay': extra.get('graph_replay_ms_per_step'),
                                                   'recorded_replay': extra.get('recorded_ms_per_step')},
                'build_s': tb['total_s'], 'build': dict(tb, gramian_steps=bstats['gramian_steps'],
                                                         outer_iterations=bstats['outer'], block=bstats['block'],
                                                         krylov_block=bstats.get('krylov_block'), method=bstats.get('method'),
                                                         recurrence=bstats.get('recurrence'), looks=[list(c) for c in bstats.get('checks', [])],
                                                         nested=bstats.get('nested'), monitor_lag=bstats.get('monitor_lag'),
                                                         converged=bstats['converged'],
                                                         final_rel_residual=bstats.get('final_rel_residual'),
                                                         spmm_launches=len(spmm_ms), spmm_ms=float(sum(spmm_ms)),
                                                         sigma_max=float(st['sigma'][0].item()),
                                                         sigma_min=float(st['sigma'][-1].item())),
                'score': {'kernel_ms': ms, 'kernel_launches_5_passes': n_launch,
                          'flagged_users': stats.get('flagged_users'), 'refolded_users': stats.get('refolded_users'),
                          'candidate_capacity': stats.get('candidate_capacity'), 'item_splits': stats.get('item_splits'),
                          'swept_fraction': swept, 'exit_tile_quantiles': stats.get('exit_tile_quantiles'),
                          'two_phase': stats.get('two_phase'),
                          'n_tiles': -(-n_items // 32)},
            }
            sh = np.asarray(st['shards'])
            cl = st['collectives']
            out['dist'] = dict(self.dist_info, users_per_rank=[int(x) for x in sh[:, 1]], nnz_per_rank=[int(x) for x in sh[:, 2]],
                               device_of_rank=[int(x) for x in sh[:, 3]],
                               build_collectives={'all_gather': cl['n_allgather'], 'reduce_scatter': cl['n_reduce_scatter'],
                                                  'all_reduce': cl['n_allreduce'], 'overlapped_panels': cl.get('n_panel_exchanges', 0),
                                                  'MB': (cl['bytes_gathered'] + cl['bytes_scattered'] + cl['bytes_reduced']) / 1e6},
                               scoring_collectives=0)
            if catalogue == 'flat':
                out['workload'] += '; item-factor rows scaled to unit norm (flat-norm catalogue: the pruning bound never fires)'
            elif catalogue == 'pop25':
                out['workload'] += '; item-factor rows rescaled to norm = (item count)^0.25 (slow, real-data-like norm decay)'
            if cand_ms:
                # the sweep computes every fp32-accurate product as THREE bf16 MFMAs (hi.hi + hi.lo + lo.hi) over the rank
                # padded to a multiple of 16: executed bf16 flops = 3 * (16 * k_steps / rank) * the algorithmic flops swept
                k_steps = int(ops.lib.pk_pack_kq(rank)) // 2
                # USEFUL work: the algorithmic flops of the tiles actually scored (2 n_users n_items k,
                # times the swept fraction), times 3 for the split product — no credit for the zero columns of the rank padded
                # to a multiple of 16.  `issued_*` is what the matrix cores were actually given (padding included).
                useful_flops = flops * swept * 3.0
                bf16_flops = useful_flops * (16.0 * k_steps / rank)
                ach = useful_flops / (cand_ms * 1e-3) / 1e12
                issued = bf16_flops / (cand_ms * 1e-3) / 1e12
                f32_eq = flops * swept / (cand_ms * 1e-3) / 1e12
                # what the sweep asks of the L2: every wave (32 users) reads the fragments of every tile it scores — 2 k_steps KB
                # per tile and wave — through 16-byte-per-lane loads.  (The UNPRUNED sweep asks 17-21 TB/s, about what the SpMM
                # gathers reach; round 4's diagnostic build prices those loads at 15 % of the loop at most: DESIGN K3)
                frag_bytes = swept * ((int(st['A'].shape[0]) + 31) // 32) * ((n_items + 31) // 32) * 2 * k_steps * 1024.0     # (this rank's users)
                out['roofline'] = {
                    'kernel': 'score_candidates_kernel', 'bound': 'mfma', 'dtype': 'bf16 (split product: 3 bf16 MFMAs per fp32-class product)',
                    'achieved': ach, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_MFMA_TFLOPS, 'traffic': None,
                    'issued': issued, 'issued_frac': issued / PEAK_BF16_MFMA_TFLOPS,
                    'avg_ms': cand_ms / max(n_chunk, 1), 'sweep_ms_per_pass': cand_ms, 'launches_per_pass': n_chunk,
                    'flop_per_launch': useful_flops / max(n_chunk, 1), 'issued_flop_per_launch': bf16_flops / max(n_chunk, 1),
                    'swept_fraction': swept, 'l2_delivery_TBps': frag_bytes / (cand_ms * 1e-3) / 1e12,
                    'note': 'achieved/frac: ALGORITHMIC flops of the tiles actually scored x 3 (split product), no padding; issued/issued_frac: '
                            'the bf16 MFMA flops actually issued (rank padded to a multiple of 16). The sweep is pruned '
                            'exactly (Cauchy-Schwarz bound, identical results; --no-prune scores every tile). At rank 50 the kernel is '
                            'not bound by the matrix cores but by its selection epilogue (VALU): see DESIGN.md K3',
                    'f32_equivalent': {'flop_per_launch': flops * swept, 'TFLOP/s': f32_eq, 'of_f32_mfma_peak': f32_eq / PEAK_FP32_MFMA_TFLOPS,
                                       'note': 'the same products on v_mfma_f32_32x32x2_f32 (round 1) are capped at 157.3 TFLOP/s'},
                    'dense_equivalent': {'flop_per_launch': flops, 'TFLOP/s': flops / (cand_ms * 1e-3) / 1e12,
                                         'of_f32_mfma_peak': flops / (cand_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}}
            if spmm_ms:
                gbps = float(sum(spmm_bytes) / (sum(spmm_ms) * 1e-3) / 1e9)
                out['roofline_build'] = {
                    'kernel': 'spmm_csr_groups_kernel', 'bound': 'hbm', 'achieved': gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                    'frac': gbps / PEAK_HBM_GBPS, 'traffic': None, 'launches': len(spmm_ms), 'total_ms': float(sum(spmm_ms)),
                    'bytes_total': float(sum(spmm_bytes)),
                    'gather_GBps': float(sum(m[2] * m[3] * float(m[5]) for _, _, m in st['spmm_ev']) / (sum(spmm_ms) * 1e-3) / 1e9),
                    'gather_note': 'nnz*nc*8 bytes of dense-row gathers per launch / time: the traffic that actually bounds this kernel'}
            fold_name = 'fold_q20' if ms.get('fold_q20') else 'spmm'
            fold_ms = ms.get(fold_name)
            if fold_ms:
                # the fold-in E' = A_test * image(V): HBM-bound by SURVEY §8(d)'s reckoning — the CSR stream of the test rows,
                # the image once, the E rows out — and in practice bound by the latency of its row gathers
                T = st['A']
                vb = T.values.element_size()
                packed = fold_name == 'fold_q20'
                row_bytes = int(st['F'].Q20[0].shape[1]) if packed else int(st['F'].V32x.stride(0) * 4)
                alg = float(T.nnz * (4 + vb) + 8 * (T.shape[0] + 1) + n_items * row_bytes + T.shape[0] * st['F'].Kx * 8)
                gbps = alg / (fold_ms * 1e-3) / 1e9
                out['roofline_foldin'] = {
                    'kernel': 'fold_q20_kernel' if packed else 'spmm_csr_groups_kernel<float, 4, float>', 'bound': 'hbm',
                    'achieved': gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': gbps / PEAK_HBM_GBPS, 'traffic': None,
                    'avg_ms': fold_ms, 'algorithmic_bytes': alg, 'image_row_bytes': row_bytes,
                    'gather_GBps': float(T.nnz) * row_bytes / (fold_ms * 1e-3) / 1e9,
                    'refolded_users': stats.get('refolded_users'),
                    'refold_ms': (ms.get('spmm_rows_list') or ms.get('spmm_flagged') or ms.get('fold_rows') or 0.0) + (ms.get('rescore_topk_refolded') or 0.0)}
            if cold_build is not None:
                out['build_cold'] = cold_build
            if cpu and comm.world == 1:
                n_score = cpu_users or min(n_users, 20000)
                V_ext = self.external_V(st)
                t_score, cpu_recs = self.cpu_scoring(c, V_ext, topk, n_score)
                gpu_recs = self.external_ids(st, extra['host_result'][:n_score].numpy())
                same = float((gpu_recs == cpu_recs).all(axis=1).mean())
                base = dict(value=n_score / t_score, unit='users/s', cores=self.blas_threads(), kind='port',
                            sample='reference scoring path (chunked GEMM + downvote + per-row argpartition, fp64) on the '
                                   'first %d users of the same matrix with the GPU-built V' % n_score,
                            score_sample_s=t_score, host_cpus=os.cpu_count(), gpu_vs_cpu_identical_rows=same,
                            speedup_scoring=out['value'] / (n_score / t_score),
                            sample_short='scoring: first %d users of the matrix, GPU-built V' % n_score)
                if cpu_build:
                    # north_star's ">= 10x on build + score" is defined on the WHOLE build: the headline matrix is
                    # factorised once by the reference's own call (scipy svds, ARPACK, tol 0 — models.py:844) on every
                    # row; larger workloads (S-1M: 1e8 nnz) keep a 5e6-nnz sample and say so
                    build_rows = n_users if cpu_build_whole else min(n_users, max(1000, int(5e6 / max(nnz / n_users, 1))))
                    t_build, hb = self.cpu_build(c, rank, build_rows)
                    base.update(build_sample_s=t_build, build_sample_users=build_rows, build_sample_nnz=hb,
                                build_whole_matrix=bool(build_rows == n_users))
                    base['sample'] += '; svds build timed on %s (%d users, %d nnz)' % (
                        'the WHOLE matrix' if build_rows == n_users else 'the first rows', build_rows, hb)
                    base['sample_short'] = 'scoring: first %d users of the matrix, GPU-built V; scipy svds (ARPACK tol 0): %s (%d users, %d nnz)' % (
                        n_score, 'WHOLE matrix' if build_rows == n_users else 'first rows', build_rows, hb)
                    if build_rows == n_users:
                        cpu_total = t_build + n_users / (n_score / t_score)      # scoring extrapolated linearly from the sample
                        base['build_s'] = t_build
                        base['speedup_build'] = t_build / out['build_s']
                        base['speedup_build_plus_score'] = cpu_total / (out['build_s'] + 1e-3 * out['ms_per_step'])
                out['cpu_baseline'] = base
        del st
        torch.cuda.empty_cache()
        return out

    def model_path(self, c, rank, topk):
        """The plugin surface end to end: SVDModel(data).build() + .get_recommendations() — host triplets in, host int64
        array out — and its lists against the kernel-level path's."""
        from polara_amd.data import ArrayData
        from polara_amd.models import SVDModel
        n_users, n_items = c['shape']
        u = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(c['indptr']))
        t0 = time.perf_counter()
        d = ArrayData((u, c['indices'], c['values']), n_users=n_users, n_items=n_items, test=(u, c['indices'], c['values']))
        t_data = time.perf_counter() - t0
        m = SVDModel(d, ops=self.ops)
        m.verbose = False
        m.rank, m.topk = rank, topk
        res = {}
        for tag in ('cold', 'warm'):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.build()
            torch.cuda.synchronize()
            t_build = time.perf_counter() - t0
            t0 = time.perf_counter()
            recs = m.get_recommendations()
            t_rec = time.perf_counter() - t0
            t0 = time.perf_counter()
            recs2 = m.get_recommendations()
            t_rec2 = time.perf_counter() - t0
            res[tag] = dict(build_s=t_build, solver_s=m.training_time[-1], get_recommendations_s=t_rec,
                            get_recommendations_again_s=t_rec2, users_per_s=n_users / t_rec2)
        res['array_data_s'] = t_data
        # the lists of the plugin surface (external ids) against the reference path on a sample
        n_chk = min(n_users, 5000)
        t_cpu, cpu_recs = self.cpu_scoring(c, np.ascontiguousarray(m.factors[d.fields.itemid]), topk, n_chk)
        res['gpu_vs_cpu_identical_rows'] = float((recs[:n_chk] == cpu_recs).all(axis=1).mean())
        res['identical_between_calls'] = bool(np.array_equal(recs, recs2))
        res['note'] = ('SVDModel(ArrayData).build() incl. COO -> device CSR, popularity relabel, CSC image, solver, factors to '
                       'host (F-ordered, external ids), serving index; get_recommendations() incl. test triplets -> device '
                       'CSR, seen-tile streams, scoring pass, D2H, internal -> external ids')
        return res, recs


def coarse_c_path(B, c, rank, topk):
    """The coarse C entry points (include/polara_hip.h: pk_mat_from_csr / pk_svd_build / pk_score_topk) on the same
    matrix: host CSR in, host factors and lists out, the library's own device memory — what a non-Python host gets."""
    import ctypes as C
    from polara_amd import _lib
    lib = _lib.load()
    n_users, n_items = c['shape']
    nnz = int(c['indptr'][-1])
    vp = C.c_void_p

    class Stats(C.Structure):
        _fields_ = [('outer', C.c_int32), ('gramian_steps', C.c_int32), ('block', C.c_int32), ('converged', C.c_int32),
                    ('final_rel_residual', C.c_double)]
    ctx, A = vp(), vp()
    _lib.check(lib.pk_ctx_create(torch.cuda.current_device(), C.byref(ctx)), 'pk_ctx_create')
    ptr = lambda a: a.ctypes.data_as(vp)
    indptr, indices, values = (np.ascontiguousarray(c['indptr'], dtype=np.int64), np.ascontiguousarray(c['indices'], dtype=np.int32),
                               np.ascontiguousarray(c['values'], dtype=np.float32))
    out = {}
    t0 = time.perf_counter()
    rc = lib.pk_mat_from_csr(ctx, n_users, n_items, nnz, ptr(indptr), ptr(indices), ptr(values), 0, C.byref(A))
    assert rc == 0, lib.pk_ctx_error(ctx)
    out['mat_from_csr_s'] = time.perf_counter() - t0
    sigma = np.empty(rank)
    V = np.empty((n_items, rank), order='F')
    st = Stats()
    for tag in ('cold', 'warm'):
        t0 = time.perf_counter()
        rc = lib.pk_svd_build(ctx, A, rank, 0, 0.0, 0, 0, ptr(sigma), ptr(V), None, C.byref(st))
        assert rc == 0, lib.pk_ctx_error(ctx)
        out['svd_build_%s_s' % tag] = time.perf_counter() - t0
    out.update(gramian_steps=st.gramian_steps, outer=st.outer, final_rel_residual=st.final_rel_residual)
    recs = np.empty((n_users, topk), dtype=np.int64)
    for tag in ('first', 'again'):
        t0 = time.perf_counter()
        rc = lib.pk_score_topk(ctx, n_items, rank, ptr(V), A, topk, 1, ptr(recs), None)
        assert rc == 0, lib.pk_ctx_error(ctx)
        out['score_topk_%s_s' % tag] = time.perf_counter() - t0
    out['users_per_s'] = n_users / out['score_topk_again_s']
    # the serving handle: what pk_score_topk sets up per call stays on the device between calls
    sv = vp()
    t0 = time.perf_counter()
    rc = lib.pk_serving_create(ctx, n_items, rank, ptr(V), A, C.byref(sv))
    assert rc == 0, lib.pk_ctx_error(ctx)
    out['serving_create_s'] = time.perf_counter() - t0
    recs_sv = np.empty_like(recs)
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        rc = lib.pk_serving_score(ctx, sv, topk, 1, ptr(recs_sv), None)
        assert rc == 0, lib.pk_ctx_error(ctx)
        times.append(time.perf_counter() - t0)
    lib.pk_serving_free(ctx, sv)
    out['serving_score_first_s'] = times[0]
    out['serving_score_again_s'] = float(np.median(times[1:]))
    out['serving_users_per_s'] = n_users / out['serving_score_again_s']
    out['serving_identical_to_score_topk'] = bool(np.array_equal(recs_sv, recs))
    n_chk = min(n_users, 5000)
    t_cpu, cpu_recs = B.cpu_scoring(c, np.ascontiguousarray(V), topk, n_chk)
    out['gpu_vs_cpu_identical_rows'] = float((recs[:n_chk] == cpu_recs).all(axis=1).mean())
    out['note'] = ('pk_svd_build / pk_score_topk: every call takes host arrays, allocates its device buffers (hipMalloc) and returns host '
                   'arrays; pk_score_topk also re-orders the catalogue by factor norm, re-sorts the test rows and builds the factor images and '
                   'seen-tile streams inside the call; pk_serving_create / pk_serving_score keep that state on the device between calls '
                   '(pageable host result: the [n_users x topk] int64 array is copied into the caller\'s memory inside the call)')
    lib.pk_mat_free(ctx, A)
    lib.pk_ctx_destroy(ctx)
    return out


def coffee_block(B, cpu=True):
    """BASELINE.json configs[3]: CoffeeModel (HOOI of the user x item x rating tensor) on the ML-1M-shaped data at the
    multilinear ranks (30,30,4) — the largest the reference itself accepts, `lib/tensor.py:79` — and (30,30,5); warm build,
    then `get_recommendations` for all users.  Parity lives in tests/test_gpu_configs.py (golden fixture from the pinned
    oracle); the un-jitted CPU oracle needs 114 s for this build and is not re-timed here."""
    from polara_amd.data import ArrayData
    from polara_amd.models import CoffeeModel
    from polara_amd.synth import make_workload, csr_to_coo_triplets
    csr, cfg = make_workload('ml1m')
    u, i, v = csr_to_coo_triplets(csr)
    n_users, n_items = csr['shape']
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))
    d = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
    out = {'workload': 'ML-1M-shaped tensor %d x %d x %d, %d entries (BASELINE.json configs[3])' % (n_users, n_items, len(np.unique(v)), len(v))}
    for mlrank in ((30, 30, 4), (30, 30, 5)):
        m = CoffeeModel(d, ops=B.ops)
        m.verbose = False
        m.mlrank, m.seed, m.topk = mlrank, 0, 10
        m.build()
        builds = []
        for _ in range(3):                # warm builds: the median of three (a lone second build now and then takes 2x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.build()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            builds.append(t1 - t0)
        recs = m.get_recommendations()
        t2 = time.perf_counter()
        recs2 = m.get_recommendations()
        t3 = time.perf_counter()
        f = d.fields
        orth = max(float(np.abs(m.factors[k].T @ m.factors[k] - np.eye(m.factors[k].shape[1])).max())
                   for k in (f.userid, f.itemid, f.feedback))
        out['mlrank_%d_%d_%d' % mlrank] = dict(build_s=sorted(builds)[1], builds_s=builds, iterations=len(m.core_norm_trace),
                                               core_norm=float(m.core_norm_trace[-1]), factors_orthonormal_to=orth,
                                               get_recommendations_s=t2 - t1, get_recommendations_again_s=t3 - t2,
                                               users_per_s=n_users / (t3 - t2), identical_between_calls=bool(np.array_equal(recs, recs2)))
    return out


def s50m_block(B, cpu=True):
    """BASELINE.json configs[4] as the share one call can hold: 1M of the 50M users against the full 500K-item catalogue,
    rank 200, top-50 (tools/bench_s50m_shard.py; rows checked against the CPU path on a 300-user sample)."""
    import argparse
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    import bench_s50m_shard
    return bench_s50m_shard.run(argparse.Namespace(users=1_000_000, items=500_000, rank=200, topk=50, steps=3,
                                                   check_users=300 if cpu else 0))


MAX_LINE_BYTES = 3000      # the driver's capture of the final stdout line; round 2's 25 KB line came back unparsed


def _r(x, nd=4):
    """numbers of the compact line: 4 significant digits are what a reader compares"""
    if x is None:
        return None
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    x = float(x)
    if x == 0.0 or not np.isfinite(x):
        return x
    from math import floor, log10
    return round(x, nd - 1 - int(floor(log10(abs(x)))))


def compact_line(head, n_gpus, steps, warmup, adversarial=None, scale=1.0):
    """The ONE stdout line of the run: headline + roofline + cpu_baseline, flat numbers only (no notes, no nested
    sub-blocks); everything else lives in bench_detail.json.  Pure function of the measurement record, so that the
    CPU suite can check its size and shape on a canned record (tests/test_host_logic.py)."""
    sc = head.get('score', {})
    cfg = {'workload': head['workload'], 'n_users': head['n_users'], 'n_items': head['n_items'], 'nnz': head['nnz'],
           'rank': head['rank'], 'topk': head['topk'], 'prune': head['prune'],
           'swept_fraction': _r(sc.get('swept_fraction')), 'launch': head.get('launch_short', 'python'),
           'launches_per_pass': head.get('launches_per_pass'),
           'parallelism': 'users sharded over %d GPU(s); build: one all-reduce of the Krylov block (n_items x 16 fp64) per Gramian '
                          'step; scoring: no collective' % n_gpus}
    if scale != 1.0:
        cfg['scale'] = scale
    if adversarial:
        cfg['adversarial_users_per_s'] = {k: _r(v) for k, v in adversarial.items()}
    r100 = head.get('rank100_top20')
    if r100:
        cfg['configs2_rank100_top20'] = {k: _r(v) for k, v in r100.items()}
    out = {'metric': 'users scored/sec + SVD build time, ML-20M rank-50 PureSVD', 'value': _r(head['value'], 6), 'unit': 'users/s',
           'n_gpus': n_gpus, 'steps': steps, 'warmup': warmup, 'ms_per_step': _r(head['ms_per_step'], 5),
           'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16x3',
           'data': head.get('data', 'synthetic'), 'config': cfg, 'build_s': _r(head['build_s']),
           'latency_ms_per_pass': _r(head.get('latency_ms_per_pass')), 'ms_per_step_serial': _r(head.get('ms_per_step_serial')),
           'ms_per_step_long_region': _r(head.get('ms_per_step_long_region'))}
    b = head.get('build', {})
    out['build'] = {k: _r(b.get(k)) for k in ('solver_s', 'gramian_steps', 'krylov_block', 'converged', 'spmm_ms') if k in b}
    di = head.get('dist')
    if di:
        bc = di.get('build_collectives', {})
        out['dist'] = {'world': di.get('world'), 'backend': di.get('backend'), 'rccl': di.get('rccl'),
                       'users_per_rank': di.get('users_per_rank'), 'nnz_per_rank': di.get('nnz_per_rank'),
                       'build_collectives': {k: _r(v) for k, v in bc.items()}, 'scoring_collectives': di.get('scoring_collectives')}
    rf = head.get('roofline')
    if rf:
        rf = dict(rf, dtype='bf16x3 (split product)') if str(rf.get('dtype', '')).startswith('bf16 (split') else rf     # (the long form stays in the detail record)
        out['roofline'] = {k: (_r(rf.get(k)) if not isinstance(rf.get(k), str) else rf.get(k)) for k in
                           ('kernel', 'bound', 'dtype', 'achieved', 'peak', 'unit', 'frac', 'issued_frac', 'l2_delivery_TBps', 'avg_ms', 'launches_per_pass',
                            'swept_fraction', 'traffic', 'traffic_commit', 'stale') if k in rf or k == 'traffic'}
    rfo = head.get('roofline_foldin')
    if rfo:
        out['roofline_foldin'] = {k: (_r(rfo.get(k)) if not isinstance(rfo.get(k), str) else rfo.get(k)) for k in
                                  ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_ms', 'traffic', 'refolded_users',
                                   'refold_ms', 'traffic_commit', 'stale') if k in rfo or k == 'traffic'}
    cold = head.get('cold')
    if cold:
        out['cold'] = {k: _r(v) for k, v in cold.items()}
    rb = head.get('roofline_build')
    if rb:
        out['roofline_build'] = {k: (_r(rb.get(k)) if not isinstance(rb.get(k), str) else rb.get(k)) for k in
                                 ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launches', 'total_ms', 'traffic',
                                  'algorithmic_bytes_per_product', 'traffic_commit', 'stale') if k in rb or k == 'traffic'}
    cb = head.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {
            'value': _r(cb['value']), 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'],
            'sample': cb.get('sample_short', cb.get('sample', ''))[:200],
            'identical_rows': _r(cb.get('gpu_vs_cpu_identical_rows')), 'build_s': _r(cb.get('build_s')),
            'build_whole_matrix': cb.get('build_whole_matrix'), 'speedup_scoring': _r(cb.get('speedup_scoring')),
            'speedup_build': _r(cb.get('speedup_build')), 'speedup': _r(cb.get('speedup_build_plus_score'))}
    line = json.dumps(out, separators=(',', ':'))
    if len(line) > MAX_LINE_BYTES:        # never again an unparseable record: drop the optional blocks, loudest last
        for k in ('build', 'roofline_build', 'latency_ms_per_pass', 'dist', 'roofline_foldin'):
            out.pop(k, None)
            line = json.dumps(out, separators=(',', ':'))
            if len(line) <= MAX_LINE_BYTES:
                break
    assert len(line) <= MAX_LINE_BYTES, len(line)
    return line


def write_detail(record):
    """bench_detail.json next to bench.py (and under gpurun_out/ when that directory exists, so that it comes back from
    the GPU box); failures to write are reported on stderr and never cost the run its line."""
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        if d != ROOT and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, 'bench_detail.json'), 'w') as f:
                json.dump(record, f, indent=1, default=lambda o: o.tolist() if hasattr(o, 'tolist') else str(o))
        except OSError as exc:
            log('could not write %s/bench_detail.json: %s' % (d, exc))


def main():
    args = parse()
    ensure_world(args)
    import polara_amd
    # the imports are done: no full cycle collection over torch's heap (34 ms each) inside the cold figures below
    gc_frozen = polara_amd.freeze_imports()
    B = Bench(args)
    comm = B.comm
    headline_rank = args.rank or {'ml20m': 50, 's1m': 50, 'ml1m': 10}[args.workload]
    headline_topk = args.topk or {'ml20m': 10, 's1m': 10, 'ml1m': 10}[args.workload]
    prune = not args.no_prune
    c = B.generate(args.workload, args.scale)
    gen_s = c['gen_s']
    # cold build: the first heavy GPU work of the process (allocator growth, page tables, code objects), itemised
    st_cold, cold = B.build(c, headline_rank, not args.no_norm_order)
    # ... and the first scoring pass of the process on it (host wall time, device idle before and after)
    from polara_amd import scoring as _scoring, runtime_info as _runtime_info
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _scoring.recommend(B.ops, st_cold['F'], st_cold['A'], headline_topk, True, prune=prune)
    torch.cuda.synchronize()
    first_pass_ms = 1e3 * (time.perf_counter() - t0)
    del st_cold
    torch.cuda.empty_cache()
    full_size = comm.world == 1 and args.scale == 1.0
    head = B.measure(c, args.workload, headline_rank, headline_topk, args.steps, args.warmup, prune=prune,
                     norm_order=not args.no_norm_order, cpu=not args.no_cpu_baseline, cpu_users=args.cpu_users,
                     cold_build=cold, cpu_build_whole=full_size and args.workload in ('ml20m', 'ml1m'))
    if comm.rank == 0:
        # what a process pays ONCE: creating the operator set (the library's code objects), its first build and its first
        # pass — next to the warm figures of the line (`build_s`, `ms_per_step`); `time_to_first_model_s` is their sum up to
        # the first model (the reference pays no warm-up, tools/timing.py:20-34)
        rt = _runtime_info()
        head['cold'] = {'time_to_first_model_s': B.ops_create_s + cold['total_s'],
                        'ops_create_s': B.ops_create_s, 'warm_up_s': B.ops.warm_up_s, 'build_cold_s': cold['total_s'],
                        'solver_cold_s': cold['solver_s'], 'first_pass_ms': first_pass_ms, 'gc_frozen_objects': gc_frozen, 'hw_queues': rt['hw_queues'],
                        'hw_queues_in_time': rt['in_time']}
    subs, adversarial = {}, {}
    if args.scale == 1.0 and not args.only_headline and args.workload == 'ml20m' and not args.rank:
        # BASELINE.json configs[2] as written (rank 100, top-20) rides in the line at every N next to the metric's rank 50
        s = B.measure(c, 'ml20m', 100, 20, max(5, min(args.steps, 20)), 2, cpu=not args.no_cpu_baseline, cpu_users=5000,
                      cpu_build=False)
        if comm.rank == 0:
            subs['configs2_ml20m_rank100_top20'] = s
            head['rank100_top20'] = {'users_per_s': s['value'], 'ms_per_step': s['ms_per_step'], 'build_s': s['build_s'],
                                     'gramian_steps': s['build']['gramian_steps']}
            if 'cpu_baseline' in s:
                head['rank100_top20']['identical_rows'] = s['cpu_baseline'].get('gpu_vs_cpu_identical_rows')
    if full_size and not args.only_headline and args.workload == 'ml20m':
        # the headline depends on how fast the item-factor norms decay (the sweep is pruned by a norm bound): the same
        # matrix with three less friendly catalogues ALWAYS runs next to it and rides in the compact line
        sub_steps = max(5, min(args.steps, 20))
        for name, kw in (('flat_norm', dict(catalogue='flat')), ('pop25_norm', dict(catalogue='pop25')),
                         ('no_prune', dict(prune=False))):
            s = B.measure(c, 'ml20m', headline_rank, headline_topk, sub_steps, 2, cpu=not args.no_cpu_baseline, cpu_users=5000,
                          cpu_build=False, **kw)
            subs[name] = s
            adversarial[name] = s['value']
    if full_size and args.detail and args.workload == 'ml20m':
        sub_steps = max(5, min(args.steps, 20))
        # the plugin surface, end to end
        mp, _ = B.model_path(c, headline_rank, headline_topk)
        subs['model_path'] = mp
        subs['coarse_c_abi'] = coarse_c_path(B, c, headline_rank, headline_topk)
        del c
        gc.collect()
        # BASELINE.json configs[1]
        c1 = B.generate('s1m')
        s = B.measure(c1, 's1m', 50, 10, sub_steps, 2, cpu=not args.no_cpu_baseline, cpu_users=5000, cpu_build=False)
        s['gen_s'] = c1['gen_s']
        subs['configs1_s1m_rank50_top10'] = s
        s = B.measure(c1, 's1m', 50, 10, 2, 1, prune=False, cpu=False)
        subs['configs1_s1m_no_prune'] = {k: s[k] for k in ('value', 'ms_per_step', 'latency_ms_per_pass', 'score', 'roofline')}
        del c1
        gc.collect()
        torch.cuda.empty_cache()
        # BASELINE.json configs[3] and configs[4]: not on the metric's path; a failure in one of these blocks is
        # reported in its place and does not cost the run its line
        for name, fn in (('configs3_coffee_ml1m', coffee_block), ('configs4_s50m_shard', s50m_block)):
            try:
                subs[name] = fn(B, cpu=not args.no_cpu_baseline)
            except Exception as exc:
                subs[name] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            gc.collect()
            torch.cuda.empty_cache()
    if comm.rank != 0:
        return
    tag = {'ml20m': 'ml20m', 's1m': 's1m'}.get(args.workload)
    traffic = pmc_traffic(tag) if (tag and full_size and headline_rank == 50) else {}
    if 'roofline' in head and traffic:
        head['roofline']['traffic_commit'] = traffic.get('commit')
        head['roofline']['stale'] = bool(traffic.get('stale_score', True))
    if 'roofline_build' in head and traffic:
        head['roofline_build']['traffic_commit'] = traffic.get('commit')
        head['roofline_build']['stale'] = bool(traffic.get('stale_spmm', True))
    if 'roofline_foldin' in head and traffic and head['roofline_foldin']['kernel'] == 'fold_q20_kernel':
        head['roofline_foldin']['traffic_commit'] = traffic.get('commit')
        head['roofline_foldin']['stale'] = bool(traffic.get('stale_fold', True))
        if 'fold' in traffic and not traffic.get('stale_fold', True):
            head['roofline_foldin']['traffic'] = traffic['fold']
    if 'roofline' in head and 'score' in traffic and not traffic.get('stale_score', True):
        head['roofline']['traffic'] = traffic['score']       # per LAUNCH, like `achieved`
        head['roofline']['traffic_note'] = ('HBM/fabric bytes per launch = 2*FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc run of '
                                            'this command at commit %s (profiles/%s_%s_pmc_*.txt)' % (traffic.get('commit'), traffic.get('round'), tag))
    if 'roofline_build' in head and 'spmm_total' in traffic and not traffic.get('stale_spmm', True):
        rb = head['roofline_build']
        products = 2 * head['build']['gramian_steps']          # A.X and A^T.Y of every Gramian step
        rb['traffic'] = traffic['spmm_total'] / (2 * products)   # the profiled command builds twice (cold + warm)
        rb['algorithmic_bytes_per_product'] = rb['bytes_total'] / products
        rb['traffic_note'] = ('HBM/fabric bytes per SpMM PRODUCT (A.X, or A^T.Y = the sum of its user-block launches): '
                              '(2*FETCH_SIZE + WRITE_SIZE) summed over every build SpMM launch of the profiled run / its %d '
                              'products; same source' % (2 * products))
    detail = dict(head)
    detail.pop('host_result', None)
    detail.update(metric='users scored/sec + SVD build time, ML-20M rank-50 PureSVD', unit='users/s', n_gpus=comm.world,
                  steps=args.steps, warmup=args.warmup, gen_s=gen_s, scale=args.scale,
                  dtype_detail='candidate scoring on bf16 MFMA with every operand split into two bf16 (3 MFMAs per product, fp32 '
                               'accumulate, error <= 3 * 2^-16 + (4 K + 10) * 2^-23 relative to ||e|| ||v||, certified); fold-in gathers '
                               'fl32(V) with f64 accumulation, its rounding is part of the certification (uncertified users are '
                               're-folded in f64); EXACT f64 re-scoring of the candidates decides every list; f64 SVD build',
                  data=('synthetic (planted low-rank + Zipf popularity, seeded; generated on GPU)' if head.get('data', 'synthetic') == 'synthetic'
                        else head['data']),
                  result='int64 [n_users x topk] copied to pinned host memory inside the timed region (double-buffered)',
                  sub=subs)
    write_detail(detail)
    log('detail record: bench_detail.json (%d sub-blocks: %s)' % (len(subs), ', '.join(subs)))
    print(compact_line(head, comm.world, args.steps, args.warmup, adversarial, args.scale), flush=True)


if __name__ == '__main__':
    main()
