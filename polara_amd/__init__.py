"""polara_amd: MI355X-native PureSVD / CoFFee hot path behind Polara's RecommenderModel surface.

    from polara_amd import SVDModel, ScaledSVD, CoffeeModel        # the names polara/__init__.py exports for this path
    from polara_amd import ArrayData, ShardedArrayData              # NumPy / on-disk data providers

Resolved on first use, so that importing the package (or its build / binding modules) does not pull in torch."""
__version__ = '0.1.0'

_EXPORTS = {
    'RecommenderModel': 'models', 'SVDModel': 'models', 'ScaledSVD': 'models', 'CoffeeModel': 'models',
    'ArrayData': 'data', 'ShardedArrayData': 'data',
    'SparseProduct': 'operator', 'find_optimal_svd_rank': 'pipelines',
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        import importlib
        return getattr(importlib.import_module('.' + _EXPORTS[name], __name__), name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))


def __dir__():
    return sorted(list(globals()) + __all__)
