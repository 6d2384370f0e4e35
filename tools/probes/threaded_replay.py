"""Can TWO host threads enqueue recorded passes faster than one?  (ctypes releases the GIL inside a library call; whether the HIP
runtime's launch path runs in parallel for different streams is the question.)  Rank 0's shard of N = 8, recordings on 4 streams.
    python tools/probes/threaded_replay.py [ml20m] [N]"""
import sys, threading, time
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
streams = [torch.cuda.Stream() for _ in range(4)]
main = torch.cuda.current_stream()
recs = []
for s in streams:
    s.wait_stream(main)
    with torch.cuda.stream(s):
        recs.append(scoring.RecordedPass(ops, F, T, 10, True))
torch.cuda.synchronize()
want = scoring.recommend(ops, F, T, 10, True)
for r in recs:
    assert torch.equal(r.replay(), want)
torch.cuda.synchronize()
K = 400


def run(mine, n):
    for i in range(n):
        mine[i % len(mine)].replay()


for threads in (1, 2, 4):
    for rep in range(2):
        groups = [recs[t::threads] for t in range(threads)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if threads == 1:
            run(recs, K)
        else:
            th = [threading.Thread(target=run, args=(g, K // threads)) for g in groups]
            for t in th:
                t.start()
            for t in th:
                t.join()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%d thread(s): enqueue %.1f us per pass, drained %.1f us per pass (%d users: %.0f M users/s for the job)'
              % (threads, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K, T.shape[0], c['shape'][0] / ((t2 - t0) / K) / 1e6), flush=True)
for r in recs:
    assert torch.equal(r.replay(), want)
