"""Per-dispatch view of a scoring pass from a rocprofv3 kernel trace.

  on the GPU box:  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- \
                       python $REPO/bench.py --only-headline --no-cpu-baseline --steps 10 --warmup 3
                   python $REPO/tools/probes/trace_pass.py /tmp/tp > gpurun_out/trace_pass.txt

Prints, for the LAST complete pass of the run (a pass = the dispatches from one fold-in spmm to the next), every
dispatch in launch order with its start offset, duration and grid, and the mean duration of each position over the last
passes — which shows what a C entry point that launches several kernels (the two-phase sweep: head, seeded splits, merge)
is made of.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(src, n_last=8):
    path = sorted(glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True))[-1]
    rows = list(csv.DictReader(open(path, newline='')))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    name = lambda r: r['Kernel_Name'].split('(')[0][:90]
    # passes: split at the fold-in (an spmm whose dense block is float: the fp32 image of V)
    starts = [i for i, r in enumerate(rows) if 'spmm_csr' in r['Kernel_Name'] and 'float' in r['Kernel_Name'].split('(')[0].split('<')[-1]]
    if len(starts) < 3:
        print('no passes found in', path)
        return
    passes = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    passes = [p for p in passes if len(p) == len(passes[-1])][-n_last:]
    acc = defaultdict(list)
    for p in passes:
        t0 = int(p[0]['Start_Timestamp'])
        for j, r in enumerate(p):
            acc[j].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    print('# %s: %d passes of %d dispatches averaged' % (path, len(passes), len(passes[-1])))
    print('%3s %10s %10s %12s  %s' % ('#', 'start_us', 'dur_us', 'grid', 'kernel'))
    tot = 0.0
    for j, r in enumerate(passes[-1]):
        st = sum(a for a, _ in acc[j]) / len(acc[j]) / 1e3
        du = sum(b for _, b in acc[j]) / len(acc[j]) / 1e3
        tot += du
        grid = 'x'.join(r.get(k, '?') for k in ('Grid_Size_X', 'Grid_Size_Y')) if 'Grid_Size_X' in r else r.get('Grid_Size', '?')
        print('%3d %10.1f %10.1f %12s  %s' % (j, st, du, grid, name(r)))
    last = passes[-1][-1]
    span = (int(last['End_Timestamp']) - int(passes[-1][0]['Start_Timestamp'])) / 1e3
    print('# sum of durations %.1f us, span of the last pass %.1f us' % (tot, span))


if __name__ == '__main__':
    main(sys.argv[1])
