"""How should a replayed pass hand its lists to the host?  Rank 0's shard of N, four recordings on four streams, K passes in
bench.py's pattern (event per pass, throttle DEPTH = 4), three hand-overs A/B/C in ONE process:
  torch   host.copy_(dev, non_blocking) under the pass's stream (torch's copy dispatch)
  copy    pk_copy_to_host_async as the recording's last call (hipMemcpyAsync)
  mapped  the pass's last kernel writes the pinned buffer itself (recommend(out=pinned))
    python tools/probes/recorded_handover.py [ml20m] [N]"""
import sys, time
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
want = scoring.recommend(ops, F, T, 10, True).cpu()
K = 400
for rep in range(2):
    for mode in ('torch', 'copy', 'mapped'):
        recs, hosts = [], []
        for s in streams:
            s.wait_stream(main)
            h = torch.empty(tuple(want.shape), dtype=torch.int64).pin_memory()
            with torch.cuda.stream(s):
                recs.append(scoring.RecordedPass(ops, F, T, 10, True, host_out=None if mode == 'torch' else h, hand_over=mode if mode != 'torch' else 'copy'))
            hosts.append(h)
        done = [torch.cuda.Event() for _ in range(4)]
        torch.cuda.synchronize()

        def loop(n):
            for i in range(n):
                r = i % 4
                if i >= 4:
                    done[r].synchronize()
                out = recs[r].replay()
                if mode == 'torch':
                    with torch.cuda.stream(streams[r]):
                        hosts[r].copy_(out, non_blocking=True)
                done[r].record(streams[r])
        loop(40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(K)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / K
        ok = all(bool((h == want).all()) for h in hosts)
        print('%-7s %.4f ms per pass (%d users; job %.0f M users/s)  lists %s' % (mode, ms, T.shape[0], c['shape'][0] / ms / 1e3, 'ok' if ok else 'WRONG'), flush=True)
        del recs, hosts
