"""bench.Bench.score_passes against the stand-alone pipelined loop of two_stream_bench_loop.py in ONE process (same box)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B
sys.argv = sys.argv[:1]
args = B.parse()
bench = B.Bench(args)
ops = bench.ops
c = bench.generate('ml20m')
st, _ = bench.build(c, 50, True)
from polara_amd import scoring
F, A = st['F'], st['A']
main = torch.cuda.current_stream()
copy_stream = torch.cuda.Stream()
ALL = [torch.cuda.Stream() for _ in range(2)]
HOST = [torch.empty((A.shape[0], 10), dtype=torch.int64).pin_memory() for _ in range(4)]


def standalone(n_streams, depth, n=20):
    streams = ALL[:n_streams] if n_streams > 1 else [main]
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
            HOST[b].copy_(recs, non_blocking=True)
            recs.record_stream(copy_stream)
            done[b].record(copy_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    for ns in (1, 2):
        standalone(ns, 4, 12)
        print(json.dumps(dict(loop='standalone', streams=ns, ms=round(standalone(ns, 4, 20), 4), ms40=round(standalone(ns, 4, 40), 4))), flush=True)
    args.graph = False
    for ps in (1, 2):
        args.pass_streams = ps
        el, recs, ex = bench.score_passes(st, 10, 20, 5)
        print(json.dumps(dict(loop='bench.score_passes', pass_streams=ps, ms_per_step=round(el / 20 * 1e3, 4), launch=ex['launch'][:50],
                              cal=[ex.get('python_launch_ms_per_step'), ex.get('pipelined_ms_per_step')])), flush=True)
