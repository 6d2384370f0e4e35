"""Timeline of ONE warm svd_topk call from a rocprofv3 kernel trace: per phase (marked by the SpMM launches of the Gramian
steps) GPU-busy time against wall time, and the kernels by total time.
  step 1 (on the box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $R/tools/probes/solver_timeline.py run [method]
  step 2:               python tools/probes/solver_timeline.py report /tmp/st > gpurun_out/solver_timeline.txt"""
import sys, os, glob, csv, collections
if sys.argv[1] == 'run':
    import numpy as np, torch
    sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
    from polara_amd.ops import HipOps
    from polara_amd.synth import make_workload, csr_to_numpy
    from polara_amd.solver import svd_topk
    from polara_amd.csr import popularity_order
    ops = HipOps('cuda:0')
    csr, cfg = make_workload(sys.argv[4] if len(sys.argv) > 4 else 'ml20m', device='cuda:0')
    c = csr_to_numpy(csr); del csr
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, inv = popularity_order(None, c['shape'][1], counts=ops.item_counts(A))
    A = ops.csr_relabel_cols(A, rank_of); A.transpose_operator(); _ = A.plan
    meth = sys.argv[2] if len(sys.argv) > 2 else 'lanczos'
    kw = dict(krylov_block=int(sys.argv[3])) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else {}
    if len(sys.argv) > 5:
        kw['monitor_lag'] = int(sys.argv[5])          # 0: every look on the calling thread (what a look costs alone)
    if len(sys.argv) > 6:
        kw['first_look'] = int(sys.argv[6])
    wl = sys.argv[4] if len(sys.argv) > 4 else 'ml20m'
    for _ in range(2):
        svd_topk(ops, A, 50, method=meth, **kw)
    torch.cuda.synchronize()
    # marker: a recognisable kernel (randn of an odd size) brackets the traced solve
    ops.randn(777, 3, 1); torch.cuda.synchronize()
    _, s, V, st = svd_topk(ops, A, 50, method=meth, **kw)
    torch.cuda.synchronize()
    ops.randn(777, 3, 2); torch.cuda.synchronize()
    print(st['gramian_steps'], st.get('nested'), st.get('krylov_block'), st.get('monitor_lag'), st.get('monitor_wait_ms'), st.get('look_ms'))
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True):
        with open(path, newline='') as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    name = lambda r: r['Kernel_Name'].split('(')[0][:60]
    # the bracket: the last two launches of the marker grid (777 * 3 elements)
    marks = [i for i, r in enumerate(rows) if 'randn' in r['Kernel_Name'].lower() or 'philox' in r['Kernel_Name'].lower() or 'distribution' in r['Kernel_Name'].lower()]
    lo, hi = marks[-2], marks[-1]
    seg = rows[lo + 1:hi]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print('solve: %d kernels, wall %.2f ms, sum of kernel durations %.2f ms' % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        a = agg[name(r)]; a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%-62s %5d %9.3f ms %8.1f us' % (k, n, t / 1e6, t / n / 1e3))
    # phases between SpMM groups: a "gap" > 60 us between consecutive kernels is host time
    gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) for a, b in zip(seg, seg[1:])]
    big = [g for g in gaps if g > 20000]
    print('gaps > 20 us: %d, total %.2f ms; gaps > 100 us: %d, total %.2f ms; all positive gaps %.2f ms' % (
        len(big), sum(big) / 1e6, len([g for g in gaps if g > 100000]), sum(g for g in gaps if g > 100000) / 1e6, sum(g for g in gaps if g > 0) / 1e6))
    # time line in 1 ms buckets: busy fraction and dominant kernel
    nb = (t1 - t0) // 1000000 + 1
    bucket = [collections.defaultdict(int) for _ in range(nb)]
    for r in seg:
        s_, e_ = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        bucket[min(nb - 1, s_ // 1000000)][name(r)] += e_ - s_
    for i, bk in enumerate(bucket):
        tot = sum(bk.values())
        top = sorted(bk.items(), key=lambda kv: -kv[1])[:3]
        print('ms %2d busy %3d%%  %s' % (i, 100 * tot // 1000000, ', '.join('%s %.0fus' % (k[:34], v / 1e3) for k, v in top)))
