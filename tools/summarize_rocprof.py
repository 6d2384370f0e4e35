"""Condenses rocprofv3 CSV output into small text summaries that fit in the repo (profiles/).

usage: python tools/summarize_rocprof.py <rocprof_output_dir> <summary_out.txt>
Handles --kernel-trace --stats (per-kernel table) and --pmc (per-kernel counter sums / launch).
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.split('(')[0]
    return name if len(name) <= n else name[:n - 3] + '...'


def kernel_source_hashes(root=None):
    """sha256 (first 16 hex digits) of the kernel sources a profile describes — computed from the tree the profiled
    command ran in, so a profile can be matched against the sources of any later checkout (bench.py: `stale`)."""
    import hashlib
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name in ('score.hip', 'spmm.hip', 'rescore.hip', 'dense.hip', 'foldq.hip'):
        try:
            with open(os.path.join(root, 'polara_amd', 'csrc', name), 'rb') as f:
                out[name] = hashlib.sha256(f.read()).hexdigest()[:16]
        except OSError:
            pass
    return out


def commit_line():
    """the commit the profiled tree was snapshotted from (written to gpurun_in/COMMIT before the gpurun call: the
    GPU box has no .git), and the hashes of the kernel sources of that tree"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        line = '# commit ' + open(os.path.join(root, 'gpurun_in', 'COMMIT')).read().strip()
    except OSError:
        line = '# commit unknown'
    return line + ''.join('\n# sha256 %s %s' % kv for kv in sorted(kernel_source_hashes(root).items()))


def main(src, dst):
    lines = [commit_line()]
    for path in sorted(glob.glob(os.path.join(src, '**', '*.csv'), recursive=True)):
        base = os.path.basename(path)
        with open(path, newline='') as f:
            rows = list(csv.DictReader(f))
        if not rows:
            continue
        cols = list(rows[0].keys())
        lines.append('## %s  (%d rows)  columns: %s' % (base, len(rows), ', '.join(cols)))
        if 'kernel_stats' in base or ('Calls' in cols and 'AverageNs' in cols):
            lines.append('%-72s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'pct'))
            for r in rows[:40]:
                lines.append('%-72s %8s %14.3f %12.2f %7s' % (short(r['Name']), r['Calls'],
                             float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r.get('Percentage', '')))
        elif 'counter_collection' in base or 'Counter_Name' in cols:
            agg = defaultdict(lambda: [0.0, set()])
            for r in rows:
                k = (short(r.get('Kernel_Name', '?')), r['Counter_Name'])
                agg[k][0] += float(r['Counter_Value'])
                agg[k][1].add(r.get('Dispatch_Id', len(agg[k][1])))
            lines.append('%-72s %-22s %10s %18s %18s' % ('kernel', 'counter', 'launches', 'sum', 'per_launch'))
            ours = ('score_', 'spmm_', 'rescore_', 'eigh_', 'gram_', 'tsmm_', 'ttm_', 'pack_', 'axpby', 'resid', 'exact', 'fold_', 'q20_', 'lanczos', 'chol')
            mine = {k: v for k, v in agg.items() if any(o in k[0] for o in ours)}
            for (kn, cn), (tot, disp) in sorted(mine.items()):
                n = max(len(disp), 1)
                lines.append('%-72s %-22s %10d %18.1f %18.1f' % (kn, cn, n, tot, tot / n))
            lines.append('-- other kernels (top by sum) --')
            rest = {k: v for k, v in agg.items() if k not in mine}
            for (kn, cn), (tot, disp) in sorted(rest.items(), key=lambda kv: -kv[1][0])[:25]:
                n = max(len(disp), 1)
                lines.append('%-72s %-22s %10d %18.1f %18.1f' % (kn, cn, n, tot, tot / n))
        elif 'kernel_trace' in base:
            agg = defaultdict(lambda: [0, 0.0, None])
            for r in rows:
                d = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                a = agg[short(r['Kernel_Name'])]
                a[0] += 1
                a[1] += d
                a[2] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'), r.get('LDS_Block_Size'),
                        r.get('Workgroup_Size'), r.get('Grid_Size'))
            lines.append('%-72s %8s %14s %12s  %s' % ('kernel', 'calls', 'total_ms', 'avg_us', '(vgpr, agpr, sgpr, lds, wg, grid) of last launch'))
            for kn, (n, tot, res) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                lines.append('%-72s %8d %14.3f %12.2f  %s' % (kn, n, tot / 1e6, tot / n / 1e3, res))
        lines.append('')
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:80]))


def append_counter_run_durations(stats_txt, pmc_txt):
    """The --stats figure of a kernel that overlaps the result copy of the previous pass is inflated (the fold-in: 366 us
    under --kernel-trace --stats, 197 us in every counter run and in bench.py's own HIP events): append the kernel_trace
    table of a counter run of the same command to the stats summary, kernel by kernel, with the ratio."""
    stats, trace = {}, {}
    for line in open(stats_txt):
        m = re.match(r'^(.{72}) +(\d+) +([\d.]+) +([\d.]+) +[\d.]+\s*$', line)
        if m:
            stats[m.group(1).strip()] = float(m.group(4))
    for line in open(pmc_txt):
        m = re.match(r'^(.{72}) +(\d+) +([\d.]+) +([\d.]+)  \(', line)
        if m:
            trace[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    rows = ['', '## durations of the same kernels in a counter run of the same command (%s): no overlap with the result copy' % os.path.basename(pmc_txt),
            '%-72s %8s %12s %12s %7s' % ('kernel', 'calls', 'avg_us', 'stats_avg_us', 'ratio')]
    for k, (n, avg) in sorted(trace.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:30]:
        if k in stats:
            rows.append('%-72s %8d %12.2f %12.2f %7.2f' % (k, n, avg, stats[k], stats[k] / avg if avg else 0.0))
    with open(stats_txt, 'a') as f:
        f.write('\n'.join(rows) + '\n')


if __name__ == '__main__':
    import re
    if sys.argv[1] == '--cross-check':
        append_counter_run_durations(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1], sys.argv[2])
