"""Host time of ONE scoring pass (enqueue only, nothing waited for) on the shard rank 0 owns at N = 8: what bounds the job's
rate once a shard's pass is shorter on the GPU than on the host.  cProfile over 300 passes + the plain loop.
    python tools/probes/pass_host_cost.py [ml20m] [N]"""
import cProfile, io, pstats, sys, time
sys.path.insert(0, '.')
import torch
WL = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sys.argv = ['bench.py', '--workload', WL]
import bench, polara_amd
from polara_amd import scoring
from polara_amd.csr import nnz_balanced_row_partition
polara_amd.freeze_imports()
B = bench.Bench(bench.parse())
c = B.generate(WL)
st, _ = B.build(c, 50)
bounds = nnz_balanced_row_partition(c['indptr'], N)
T = st['A'] if N == 1 else B.ops.csr_rows(st['A'], 0, int(bounds[1]))
F, ops = st['F'], B.ops
for _ in range(30):
    scoring.recommend(ops, F, T, 10, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    scoring.recommend(ops, F, T, 10, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%d users: enqueue %.1f us per pass; with the GPU drained %.1f us per pass' % (T.shape[0], 1e6 * (t1 - t0) / 300, 1e6 * (t2 - t0) / 300))
rp = scoring.RecordedPass(ops, F, T, 10, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    rp.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('recorded: %d calls (%s); replay enqueue %.1f us per pass; with the GPU drained %.1f us' % (len(rp.calls), ' '.join(n for n, _, _ in rp.calls), 1e6 * (t1 - t0) / 300, 1e6 * (time.perf_counter() - t0) / 300))
import collections
per = collections.OrderedDict()
for name, fn, args in rp.calls:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        fn(*args)
    per[name + '/%d args' % len(args)] = per.get(name + '/%d args' % len(args), 0.0) + 1e6 * (time.perf_counter() - t0) / 100
    torch.cuda.synchronize()
print('host us per call (100 back-to-back enqueues each):', {k: round(v, 1) for k, v in per.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    scoring.recommend(ops, F, T, 10, True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print(s.getvalue()[:7000])
