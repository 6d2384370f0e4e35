"""Strong-scaling proxies on ONE GPU (round 2): the work rank 0 would own at N = 1, 2, 4, 8 GPUs.
  scoring: see tools/probes/shard_pass_modes.py (bench.py's own loop on rank 0's nnz-balanced user shard);
  build:   the eigensolver on rank 0's row shard with the exchange stubbed (NoComm), per-kernel-class times from HIP
           events, plus the MODELLED exchange: per Gramian step one all-gather of X and one reduce-scatter of Z
           [n_items x l] fp64 (= the volume of a sum all-reduce) over a ring of N GPUs at 100 GB/s per direction
           (xGMI: 153 GB/s nominal per link; `busbw_300` re-prices it at the 300 GB/s bus bandwidth RCCL reaches when it
           drives all seven links) and the l x l all-reduces.  The item-side dense kernels (Gram, tall-skinny GEMM,
           recurrence, residual: solver.ItemRows shards their rows) are RECORDED from that run and REPLAYED at
           n_items and at ceil(n_items / N) rows: `non_spmm_sharded_ms` = measured non-SpMM time - replay(n_items) +
           replay(n_items / N).
usage: python tools/probes/scale_proxy2.py [ml20m|s1m] [rank] [topk]"""
import sys, time, json, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, numpy as np
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk, NoComm, choose_krylov_block, default_block
from polara_amd.csr import popularity_order, nnz_balanced_row_partition
from polara_amd import scoring
ops = HipOps('cuda:0')
WL = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
csr, cfg = make_workload(WL, device='cuda:0')
rank = int(sys.argv[2]) if len(sys.argv) > 2 else {'ml20m': 50}.get(WL, cfg['rank'])
topk = int(sys.argv[3]) if len(sys.argv) > 3 else {'ml20m': 10}.get(WL, cfg['topk'])
LINK = 100e9
c = csr_to_numpy(csr); del csr
n_users, n_items = c['shape']
A0 = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
rank_of, inv = popularity_order(None, n_items, counts=ops.item_counts(A0))
A0 = ops.csr_relabel_cols(A0, rank_of)
out = {'workload': WL, 'rank': rank, 'topk': topk, 'build': {}, 'scoring': {}}
V = None
ITEM_OPS = ('gram', 'tsmm', 'tsmm_sub', 'axpbypcz', 'resid_colnorm2', 'scale_cols')


class Recorder:
    """passes every call through to the device ops and logs the dense ones with their operand shapes"""

    def __init__(self, inner):
        self.inner, self.log = inner, []

    def __getattr__(self, name):
        f = getattr(self.inner, name)
        if name not in ITEM_OPS:
            return f

        def call(*a, **k):
            self.log.append((name, [tuple(x.shape) if torch.is_tensor(x) else x for x in a]))
            return f(*a, **k)
        return call


def replay(log, rows):
    """the logged item-side calls (first dimension n_items) at `rows` rows, on scratch operands; ms in total"""
    calls = []
    for name, args in log:
        if not (isinstance(args[0 if name != 'axpbypcz' else 1], tuple) and args[0 if name != 'axpbypcz' else 1][0] == n_items):
            continue
        mk = lambda shp: torch.randn((rows,) + tuple(shp[1:]), dtype=torch.float64, device='cuda:0') if shp[0] == n_items \
            else torch.randn(shp, dtype=torch.float64, device='cuda:0')
        calls.append((name, [mk(a) if isinstance(a, tuple) else a for a in args]))
    def run():
        for name, a in calls:
            getattr(ops, name)(*a)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), len(calls)


class ShardComm(NoComm):
    """rank 0 of an N-rank job with the exchange stubbed (the sum over the ranks is the identity): since round 6 a sharded
    build runs the library's recurrence too — this rank's products, ONE all-reduce of the Krylov block per step (modelled
    below), the orthogonalisation replicated on every rank"""

    def __init__(self, world):
        self.world = world


METHOD = None
NNZ_TOTAL = int(c['indptr'][-1])
for N in (1, 2, 4, 8):
    bounds = nnz_balanced_row_partition(c['indptr'], N)
    A = A0 if N == 1 else ops.csr_rows(A0, 0, int(bounds[1]))
    KW = {}
    if N > 1:      # the JOB's decisions (the stub cannot sum the entry count): block width, look schedule
        from polara_amd.solver import _lanczos_model
        import math
        lw = default_block(rank, n_items)
        kb = choose_krylov_block(NNZ_TOTAL, n_items, lw, N)
        steps_m, t_m = _lanczos_model(NNZ_TOTAL, n_items, lw, kb, N)
        lag = int(min(10, max(3, math.ceil(5e-3 * max(1.0, lw / 64.0) ** 1.5 / t_m))))
        KW = dict(comm=ShardComm(N), krylov_block=kb, monitor_lag=lag, first_look=int(math.ceil((0.7 if lag >= 5 else 0.5) * steps_m)))
    _ = A.plan
    _, _, _, st0 = svd_topk(ops, A, rank, method=METHOD, **KW)                     # warm-up (allocations)
    if N == 1:
        METHOD = st0['method'].split()[0]      # the JOB's choice (the all-reduced entry count decides, not the shard's)
    for _rep in range(2):          # (the first timed build creates the library's timing events: tens of microseconds each)
        torch.cuda.synchronize()
        ops.timers = {}
        t0 = time.perf_counter()
        _, s, Vn, st = svd_topk(ops, A, rank, method=METHOD, **KW)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    rec = Recorder(ops)
    timers, ops.timers = ops.timers, None
    svd_topk(rec, A, rank, method=METHOD, **KW)
    torch.cuda.synchronize()
    ops.timers = timers
    full_ms, n_calls = replay(rec.log, n_items)
    shard_ms, _ = replay(rec.log, -(-n_items // N))
    spmm_ms = sum(a.elapsed_time(b) for a, b, _ in ops.timers.get('spmm', []))
    ops.timers = None
    l = st.get('krylov_block') or st['block']          # the exchanged block is a Krylov block (round 6: narrower than the nested width)
    z_bytes = n_items * l * 8
    # small all-reduces per step: block Lanczos reduces the block column of T and the l x l Gram matrices of its three
    # orthogonalisation passes (5 per step, <= 0.5 MB each); the subspace iteration about as many per filter step
    small = (5 if st.get('method') == 'lanczos' else 3) * st['gramian_steps'] + 4 * st['outer']
    if st.get('recurrence') == 'library':
        small = 0          # the replicated item side exchanges nothing but the block itself
    ring = 0.0 if N == 1 else st['gramian_steps'] * (2.0 * (N - 1) / N * z_bytes / LINK + 2 * (N - 1) * 5e-6) \
        + small * (2 * (N - 1) * 5e-6)
    bus = 0.0 if N == 1 else st['gramian_steps'] * (2.0 * (N - 1) / N * z_bytes / 300e9 + 2 * (N - 1) * 5e-6) \
        + small * (2 * (N - 1) * 5e-6)
    sharded_wall = wall - 1e-3 * (full_ms - shard_ms)
    out['build']['N=%d' % N] = dict(rows_on_rank0=A.shape[0], solver_wall_s=wall, spmm_ms=spmm_ms, non_spmm_ms=1e3 * wall - spmm_ms,
                                    item_side_calls=n_calls, item_side_ms_replicated=full_ms, item_side_ms_sharded=shard_ms,
                                    non_spmm_sharded_ms=1e3 * sharded_wall - spmm_ms,
                                    gramian_steps=st['gramian_steps'], method=st.get('method'), krylov_block=st.get('krylov_block'), recurrence=st.get('recurrence'), small_allreduces=small,
                                    modelled_exchange_ms=1e3 * ring,
                                    modelled_exchange_ms_busbw_300=1e3 * bus,
                                    modelled_total_s_replicated=wall + ring,
                                    modelled_total_s=sharded_wall + ring, modelled_total_s_busbw_300=sharded_wall + bus)
    if N == 1:
        V = Vn
b1 = out['build']['N=1']['modelled_total_s']
for k, v in out['build'].items():
    v['speedup_vs_N1'] = b1 / v['modelled_total_s']
    v['speedup_vs_N1_replicated'] = b1 / v['modelled_total_s_replicated']
    v['speedup_vs_N1_busbw_300'] = b1 / v['modelled_total_s_busbw_300']
# The scoring leg of rounds 2-5 (a hipGraph replay, one pass at a time) left this file at the end of round 6: a shard's pass is
# bound by how the HOST launches it, so the loop that matters is bench.py's own — tools/probes/shard_pass_modes.py runs that loop,
# with its calibration of the launch forms, on the same shards (profiles/r06_shard_pass_modes_*.txt).
out['scoring'] = 'tools/probes/shard_pass_modes.py'
print(json.dumps(out))
