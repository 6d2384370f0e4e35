"""The FIRST scoring pass of a process (bench.py's cold.first_pass_ms: 36-38 ms against 0.6-0.9 ms warm): cProfile of the one
call plus the caching allocator's device allocations, after a build in the same fresh process.
    python tools/probes/cold_pass_profile.py [ml20m]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, '.')
import torch
MODE = sys.argv[2] if len(sys.argv) > 2 else 'profile'      # 'plain': no allocator statistics, no profiler
sys.argv = ['bench.py', '--workload', sys.argv[1] if len(sys.argv) > 1 else 'ml20m']
import bench
from polara_amd import scoring
args = bench.parse()
B = bench.Bench(args)
c = B.generate(args.workload)
st, tb = B.build(c, 50)
torch.cuda.synchronize()
m0 = torch.cuda.memory_stats() if MODE == 'profile' else {}
pr = cProfile.Profile()
t0 = time.perf_counter()
if MODE == 'profile':
    pr.enable()
r = scoring.recommend(B.ops, st['F'], st['A'], 10, True)
torch.cuda.synchronize()
t1 = time.perf_counter()
if MODE == 'profile':
    pr.disable()
m1 = torch.cuda.memory_stats()
print('first pass %.2f ms; device allocs %d (%.1f MB reserved)' % (1e3 * (t1 - t0), m1.get('num_device_alloc', 0) - m0.get('num_device_alloc', 0),
      (m1.get('reserved_bytes.all.current', 0) - m0.get('reserved_bytes.all.current', 0)) / 2**20))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
t0 = time.perf_counter()
r = scoring.recommend(B.ops, st['F'], st['A'], 10, True)
torch.cuda.synchronize()
print('second pass %.2f ms' % (1e3 * (time.perf_counter() - t0)))
