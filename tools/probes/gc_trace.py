"""Does the cycle collector land in bench.py's cold figures?  bench.main() with every collection logged (generation, duration,
offset from the start of the process's main()); `nogc` as first argument runs it with the collector off instead.
    python tools/probes/gc_trace.py [nogc] [bench.py flags]"""
import gc, sys, time
sys.path.insert(0, '.')
nogc = len(sys.argv) > 1 and sys.argv[1] == 'nogc'
if nogc:
    del sys.argv[1]
import torch
import bench
T0 = time.perf_counter()
marks = {}


def cb(phase, info):
    if phase == 'start':
        marks['t'] = time.perf_counter()
    else:
        dt = time.perf_counter() - marks['t']
        if dt > 1e-3:
            print('gc gen %d: %.1f ms at %.3f s (collected %d)' % (info['generation'], 1e3 * dt, marks['t'] - T0, info['collected']), file=sys.stderr)


if nogc:
    gc.disable()
else:
    gc.callbacks.append(cb)
bench.main()
