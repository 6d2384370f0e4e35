"""bench.py's timed loop (result of every pass copied to pinned host memory on a copy stream, host throttled to `depth`
passes ahead) with consecutive passes on 1 or 2 compute streams: which part limits the pipelined rate?"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
sys.argv = sys.argv[:1]
bench = B.Bench(B.parse())
ops = bench.ops
c = bench.generate('ml20m')
st, _ = bench.build(c, 50, True)
from polara_amd import scoring
F, A = st['F'], st['A']
main = torch.cuda.current_stream()
copy_stream = torch.cuda.Stream()
ALL_STREAMS = [torch.cuda.Stream() for _ in range(3)]
HOST = [torch.empty((A.shape[0], 10), dtype=torch.int64).pin_memory() for _ in range(8)]


def run(n_streams, depth, d2h, n=40):
    streams = ALL_STREAMS[:n_streams] if n_streams > 1 else [main]
    host = HOST[:depth]
    done = [torch.cuda.Event() for _ in range(depth)]
    for s in streams:
        s.wait_stream(main)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        b = i % depth
        if i >= depth:
            done[b].synchronize()
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            recs = scoring.recommend(ops, F, A, 10, True)
        ready = torch.cuda.Event()
        ready.record(s)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            if d2h:
                host[b].copy_(recs, non_blocking=True)
            recs.record_stream(copy_stream)
            done[b].record(copy_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for ns, depth, d2h in ((1, 2, True), (2, 2, True), (2, 4, True), (2, 8, True), (2, 2, False), (2, 4, False), (3, 6, True), (1, 2, True), (2, 4, True)):
    run(ns, depth, d2h, 12)
    print(json.dumps(dict(streams=ns, depth=depth, d2h=d2h, ms_per_pass=round(run(ns, depth, d2h), 4))), flush=True)
