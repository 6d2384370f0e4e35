"""What bench.py's scoring loop does with the user shard rank 0 owns at N = 1, 2, 4, 8 (one GPU: the pass has no collective,
so N x users of the shard / time is the job's rate): the loop's own calibration (serial / pipelined / graph) with 2, 3, 4 and
6 pass streams.
    python tools/probes/shard_pass_modes.py [ml20m|s1m] [max streams ...]"""
import json, sys
sys.path.insert(0, '.')
import numpy as np
import torch
WL = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
STREAMS = [int(a) for a in sys.argv[2:]] or [2, 3, 4, 6]
REC = 4      # streams of the replayed form whatever the launched form uses
sys.argv = ['bench.py', '--workload', WL, '--pass-streams', str(max(max(STREAMS), REC))]
import bench
import polara_amd
from polara_amd.csr import nnz_balanced_row_partition
polara_amd.freeze_imports()
args = bench.parse()
B = bench.Bench(args)
B.pass_streams_all = [torch.cuda.Stream(device=B.dev) for _ in range(max(max(STREAMS), REC))]
B.rec_streams_all = B.pass_streams_all
c = B.generate(WL)
rank, topk = 50, 10
st, _ = B.build(c, rank)
As = st['A']
n_users = As.shape[0]
for N in (1, 2, 4, 8):
    bounds = nnz_balanced_row_partition(c['indptr'], N)
    T = As if N == 1 else B.ops.csr_rows(As, 0, int(bounds[1]))
    st_n = dict(st, A=T)
    row = {'N': N, 'users_on_rank0': int(T.shape[0])}
    for ns in STREAMS:
        B.args.pass_streams = ns
        B.rec_streams_all = B.pass_streams_all[:max(ns, REC)]
        elapsed, recs, ex = B.score_passes(st_n, topk, 40, 5)
        ms = 1e3 * elapsed / 40
        row['streams_%d' % ns] = {'ms_per_step': round(ms, 4), 'job_users_per_s': round(n_users / (ms * 1e-3) / 1e6, 1), 'mode': ex['launch'][:28],
                                  'serial': round(ex['serial_ms_per_step'], 4), 'pipelined': round(ex['pipelined_ms_per_step'] or 0, 4),
                                  'graph': None if ex['graph_replay_ms_per_step'] is None else round(ex['graph_replay_ms_per_step'], 4),
                                  'recorded': None if ex.get('recorded_ms_per_step') is None else round(ex['recorded_ms_per_step'], 4)}
    print(json.dumps(row), flush=True)
