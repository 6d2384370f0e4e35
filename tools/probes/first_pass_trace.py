"""Where the FIRST scoring pass of bench.py's process goes (cold.first_pass_ms): bench.main() itself, with the first call of
scoring.recommend traced — every call into the operator set timed, with a device synchronisation after it.
    python tools/probes/first_pass_trace.py [bench.py flags]"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from polara_amd import scoring

real = scoring.recommend
state = {'done': False, 'depth': 0}


def traced(ops, *a, **kw):
    if state['done'] or state['depth']:
        return real(ops, *a, **kw)
    state['depth'] = 1
    rows, saved = [], {}
    for name in dir(ops):
        f = getattr(ops, name)
        if name.startswith('_') or not callable(f) or isinstance(f, type) or name in ('stream', 'monitor_stream'):
            continue
        saved[name] = f

        def make(name, f):
            def g(*aa, **kk):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = f(*aa, **kk)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                rows.append((name, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
                return r
            return g
        try:
            setattr(ops, name, make(name, f))
        except AttributeError:
            saved.pop(name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        return real(ops, *a, **kw)
    finally:
        torch.cuda.synchronize()
        total = 1e3 * (time.perf_counter() - t0)
        for name in saved:
            try:
                delattr(ops, name)
            except AttributeError:
                setattr(ops, name, saved[name])
        state['done'] = True
        state['depth'] = 0
        print('first pass, traced: %.2f ms in all; calls (host ms, device tail ms):' % total, file=sys.stderr)
        for name, h, d in rows:
            print('  %-28s %8.3f %8.3f' % (name, h, d), file=sys.stderr)
        print('  sum of calls %.2f ms' % sum(h + d for _, h, d in rows), file=sys.stderr)


scoring.recommend = traced
bench.main()
