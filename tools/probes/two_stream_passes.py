"""Consecutive scoring passes on ONE stream against passes alternating between TWO streams (pass i + 1's fold-in can fill
the SIMDs the tail of pass i's sweep leaves idle).  usage: python tools/probes/two_stream_passes.py [ml20m|s1m]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
sys.argv = sys.argv[:1]
bench = B.Bench(B.parse())
ops = bench.ops
c = bench.generate(wl)
st, _ = bench.build(c, 50, True)
from polara_amd import scoring
F, A = st['F'], st['A']
main = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in range(3)]


def run(n_streams, n=40, batches=None):
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if n_streams == 1:
            outs.append(scoring.recommend(ops, F, A, 10, True, batches=batches))
        else:
            s = streams[i % n_streams]
            with torch.cuda.stream(s):
                outs.append(scoring.recommend(ops, F, A, 10, True, batches=batches))
        if len(outs) > 4:
            outs.pop(0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, outs[-1]


ref = scoring.recommend(ops, F, A, 10, True)
for k, b in ((1, None), (2, None), (1, 2), (1, 3), (1, 4), (2, 2), (1, None), (2, None)):
    run(k, 6, b)
    ms, last = run(k, 40, b)
    print(json.dumps(dict(workload=wl, streams=k, batches=b, ms_per_pass=round(ms, 4), users_per_s=round(A.shape[0] / ms * 1e3), same=bool(torch.equal(ref, last)))), flush=True)
