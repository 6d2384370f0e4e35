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
# loop 0.81 instead of 0.66 ms per step).  Read by the runtime when it initialises, i.e. at the first use of the device:
# setting it here works when the package is imported before that (a value the user exported wins).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

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
