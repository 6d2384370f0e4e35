"""(round 2) per-call durations of the kernels of a scoring pass out of a rocprofv3 --kernel-trace CSV, in launch order:
shows whether a kernel runs longer inside the pipelined timed loop (result copy of the previous pass in flight) than in
the instrumented passes.   usage: python tools/probes/trace_fold_calls.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
from collections import defaultdict
rows = []
for p in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
want = ('spmm_csr_groups_kernel<float, 4, float', 'score_candidates_kernel', 'rescore_topk_kernel', 'copyBuffer')
per = defaultdict(list)
for s, e, n in rows:
    for w in want:
        if w in n:
            per[w].append((e - s) / 1e3)
for w, v in per.items():
    print(w, len(v), 'calls; us in launch order:', ' '.join('%.0f' % x for x in v[:200]))
