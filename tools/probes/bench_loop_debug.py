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
for graph in (False, True):
    for ps in (2, 1, 2):
        args.graph = graph
        args.pass_streams = ps
        el, recs, ex = bench.score_passes(st, 10, 20, 5)
        print(json.dumps(dict(graph=graph, pass_streams=ps, ms_per_step=round(el / 20 * 1e3, 4), serial=ex.get('serial_ms_per_step'), launch=ex['launch'][:60],
                              cal=[ex.get('python_launch_ms_per_step'), ex.get('graph_replay_ms_per_step')])), flush=True)
