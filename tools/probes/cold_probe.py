"""What a process pays before its first build and pass run like its second: one child process per setting
   (--warm none: code objects loaded at first launch; code: HipOps() loads them — the default since round 6; pipeline: it also
   runs the miniature build + passes of round 5), each timing  HipOps()  ->  first build (itemised)  ->  first pass  ->  second build  ->  a warm pass,
   with the allocator's device allocations counted per stage.
   usage: python tools/probes/cold_probe.py [ml20m|s1m] [rank]"""
import json
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    t_start = time.perf_counter()
    sys.path.insert(0, '.')
    import numpy as np
    import torch
    t_torch = time.perf_counter() - t_start
    wl, rank = sys.argv[2], int(sys.argv[3])
    argv, sys.argv = sys.argv, ['bench.py', '--workload', wl, '--warm', sys.argv[4]]
    import bench
    args = bench.parse()
    sys.argv = argv
    from polara_amd import scoring
    torch.cuda.init()
    torch.zeros(1, device='cuda:0')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    B = bench.Bench(args)
    torch.cuda.synchronize()
    t_ops = time.perf_counter() - t0
    c = B.generate(wl)
    torch.cuda.synchronize()
    topk = 10

    def mallocs():
        return torch.cuda.memory_stats().get('num_device_alloc', 0)
    rec = dict(warm=argv[4], import_torch_s=t_torch, ops_create_s=t_ops, warm_up_s=B.ops.warm_up_s)
    m0 = mallocs()
    st, tb = B.build(c, rank)
    rec['build_first'] = dict(tb, device_allocs=mallocs() - m0)
    rec['time_to_first_model_s'] = t_ops + tb['total_s']
    m0 = mallocs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = scoring.recommend(B.ops, st['F'], st['A'], topk, True)
    torch.cuda.synchronize()
    rec['pass_first_ms'] = 1e3 * (time.perf_counter() - t0)
    rec['pass_first_device_allocs'] = mallocs() - m0
    t0 = time.perf_counter()
    r = scoring.recommend(B.ops, st['F'], st['A'], topk, True)
    torch.cuda.synchronize()
    rec['pass_second_ms'] = 1e3 * (time.perf_counter() - t0)
    del st, r
    m0 = mallocs()
    st, tb = B.build(c, rank)
    rec['build_second'] = dict(tb, device_allocs=mallocs() - m0)
    for _ in range(10):
        scoring.recommend(B.ops, st['F'], st['A'], topk, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        scoring.recommend(B.ops, st['F'], st['A'], topk, True)
    torch.cuda.synchronize()
    rec['pass_warm_ms'] = 1e3 * (time.perf_counter() - t0) / 20
    print('RESULT ' + json.dumps(rec))
else:
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    rank = sys.argv[2] if len(sys.argv) > 2 else '50'
    out = []
    for warm in ('none', 'code', 'pipeline'):
        r = subprocess.run([sys.executable, __file__, 'child', wl, rank, warm], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        if not line:
            print('child failed:', r.stderr[-1500:])
            continue
        rec = json.loads(line[-1][7:])
        out.append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/cold_probe_%s_r%s.json' % (wl, rank), 'w') as f:
        json.dump(out, f, indent=1)
