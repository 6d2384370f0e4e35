"""polara_amd: MI355X-native PureSVD / CoFFee hot path behind Polara's RecommenderModel surface.

    from polara_amd import SVDModel, ScaledSVD, CoffeeModel        # the names polara/__init__.py exports for this path
    from polara_amd import ArrayData, ShardedArrayData              # NumPy / on-disk data providers

Resolved on first use, so that importing the package (or its build / binding modules) does not pull in torch."""
import os as _os

__version__ = '0.1.0'

# The block Lanczos build runs its convergence monitors on a side stream next to the sparse products, the scoring loop
# alternates passes between two streams, and RCCL brings streams of its own.  The HIP runtime maps streams round-robin
# onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the other: with a process
# group initialised the default costs the build a quarter (ML-20M-shaped solve 41 ms instead of 32; bench.py's two-stream
# loop 0.81 instead of 0.66 ms per step).  The runtime reads the variable when it initialises, i.e. at the first use of
# the device: `configure_runtime()` sets it (a value the user exported wins) and RECORDS whether that was in time;
# `runtime_info()` says what the process ended up with, HipOps() warns loudly when it is the slow configuration, and
# bench.py prints it in its line.  Importing the package calls configure_runtime() once — the documented side effect of
# the import (README) — so that `import polara_amd` before the first device use is all a program needs.
HW_QUEUES_WANTED = 8
_runtime = {'hw_queues': None, 'source': None, 'in_time': None}


def _hip_already_initialised():
    import sys as _sys
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:          # a torch without a HIP build: nothing was initialised
        return False


def configure_runtime(hw_queues=HW_QUEUES_WANTED):
    """Ask the HIP runtime for `hw_queues` hardware queues (GPU_MAX_HW_QUEUES) unless the user exported a value, and
    record whether the request can still take effect.  Returns runtime_info().  Call it (or import the package) BEFORE
    the first use of the device; afterwards the runtime keeps the value it started with."""
    started = _hip_already_initialised()
    user = _os.environ.get('GPU_MAX_HW_QUEUES')
    if _runtime['hw_queues'] is not None and _runtime['in_time']:
        return runtime_info()                       # settled by an earlier call that was in time
    if user is not None:
        try:
            q = int(user)
        except ValueError:
            q = 4
        _runtime.update(hw_queues=q, source='environment', in_time=True)
    elif started:
        _runtime.update(hw_queues=4, source='runtime default (the device was used before polara_amd was imported)', in_time=False)
    else:
        _os.environ['GPU_MAX_HW_QUEUES'] = str(int(hw_queues))
        _runtime.update(hw_queues=int(hw_queues), source='polara_amd.configure_runtime', in_time=True)
    return runtime_info()


def runtime_info():
    """dict(hw_queues, source, in_time, ok): the hardware queues this process' HIP runtime works with as far as the package
    can know (the runtime has no query for it), and whether that is the configuration the multi-stream paths were
    measured on (ok: at least HW_QUEUES_WANTED)."""
    info = dict(_runtime)
    info['ok'] = bool(info['hw_queues'] is not None and info['hw_queues'] >= HW_QUEUES_WANTED)
    return info


configure_runtime()


def freeze_imports():
    """For a process that is about to build or serve: import the package's host modules, then move everything allocated so
    far into the cycle collector's permanent generation (`gc.freeze()`).  CPython triggers a FULL collection whenever the
    young survivors exceed a quarter of what the last full collection saw; right after `import torch` that count is small and
    the heap holds ~half a million objects, so the first second of the process runs three full collections of 34 ms each —
    one of them used to land inside the first build or the first scoring pass of `bench.py` (tools/probes/gc_trace.py: the
    35-38 ms `cold.first_pass_ms` of rounds 5-6 were this; the pass itself takes 1.4-1.6 ms cold).  Frozen objects are still
    freed by reference counting; only cycles among them are never looked at again.  Process-global, therefore explicit: the
    library never calls it by itself.  Returns the number of frozen objects."""
    import gc
    import importlib
    for mod in ('ops', 'solver', 'scoring', 'models', 'data', 'operator', 'dist'):
        importlib.import_module('.' + mod, __name__)
    gc.freeze()
    return gc.get_freeze_count()

_EXPORTS = {
    'RecommenderModel': 'models', 'SVDModel': 'models', 'ScaledSVD': 'models', 'CoffeeModel': 'models',
    'ArrayData': 'data', 'ShardedArrayData': 'data',
    'SparseProduct': 'operator', 'find_optimal_svd_rank': 'pipelines', 'find_optimal_tucker_ranks': 'pipelines',
    'find_optimal_config': 'pipelines',
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        import importlib
        return getattr(importlib.import_module('.' + _EXPORTS[name], __name__), name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))


def __dir__():
    return sorted(list(globals()) + __all__)
