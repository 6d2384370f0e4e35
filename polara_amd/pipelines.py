"""Rank sweep of an SVD-based model on one build (evaluation/pipelines.py:81-116 restated without pandas).

The reference's `find_optimal_svd_rank` builds once at the largest rank and then only TRUNCATES the cached
factors (the `rank` setter, models.py:812-832) before every evaluation; here the truncated item factors are
re-imaged on the device (scoring.FactorImage) and the whole loop — scoring, hit ranks — stays there.
"""
import numpy as np


def _metric_value(scores, target_metric):
    scores = scores if isinstance(scores, list) else [scores]
    table = {}
    for s in scores:
        table.update(s._asdict())
    if isinstance(target_metric, str):
        return table[target_metric]
    if callable(target_metric):
        return target_metric(table)
    raise NotImplementedError


def find_optimal_svd_rank(model, ranks, target_metric, return_scores=False, protect_factors=True, verbose=False,
                          metric_type='all', **evaluate_kwargs):
    """Returns the rank (of `ranks`) with the largest `target_metric` — a metric field name such as 'hr',
    'precision', 'map', 'ndcg', or a callable on the dict of all computed fields — and, with
    return_scores=True, also the dict rank -> value in the order of `ranks`.
    protect_factors: put the full-rank factors back when the sweep is over (the default, as in the reference)."""
    ranks = list(ranks)
    top = max(ranks + [model.rank])
    model.rank = top                                  # growing the rank invalidates the model, shrinking never does
    if not model._is_ready:
        quiet, model.verbose = model.verbose, verbose
        try:
            model.build()
        finally:
            model.verbose = quiet
    full = dict(model.factors) if protect_factors else None
    value = {}
    try:
        for r in sorted(set(ranks), reverse=True):    # every step is a truncation of what the previous one left
            model.rank = r
            model._recommendations = None             # lists of another rank are not this rank's lists
            value[r] = _metric_value(model.evaluate(metric_type, **evaluate_kwargs), target_metric)
    finally:
        model._recommendations = None
        if full is not None:
            model._rank, model.factors = top, full    # behind the setter's back: nothing to truncate, nothing to rebuild
    # ties go to the largest rank, as in the reference (idxmax over a table filled from the largest rank down)
    best = max(sorted(set(ranks)), key=lambda r: (value[r], r))
    return (best, {r: value[r] for r in ranks}) if return_scores else best


def set_config(model, config, convert_nan=True):
    """model.<name> = value for every item of `config`; NaN means None (evaluation/pipelines.py:56-60)."""
    for name, value in config.items():
        if convert_nan:
            value = value if value == value else None
        setattr(model, name, value)


def find_optimal_tucker_ranks(model, tucker_ranks, target_metric, return_scores=False, config=None, verbose=False,
                              same_space=False, metric_type='all', **evaluate_kwargs):
    """Multilinear-rank sweep of a CoFFee model on ONE HOOI build (evaluation/pipelines.py:119-160): built at the largest
    rank of every mode, each smaller (r1, r2, r3) comes from the `mlrank` setter — core rounding on the device
    (models.py:949-980 of the reference; `CoffeeModel.mlrank` here) — and the full factors go back after every step.
    Combinations a Tucker core cannot have (one rank above the product of the other two) are skipped, with
    same_space=True also those with r1 != r2.  Returns the best (r1, r2, r3) — ties go to the first in sorted order, as
    pandas' idxmax over the sorted index does — and with return_scores=True the dict mlrank -> value in sorted order."""
    if config:
        set_config(model, config)
    model.mlrank = tuple(max(mode_ranks) for mode_ranks in tucker_ranks)
    if not model._is_ready:
        quiet, model.verbose = model.verbose, verbose
        try:
            model.build()
        finally:
            model.verbose = quiet
    factors = dict(model.factors)
    full = model.mlrank
    value = {}
    for r1 in tucker_ranks[0]:
        for r2 in tucker_ranks[1]:
            if same_space and r2 != r1:
                continue
            for r3 in tucker_ranks[2]:
                if r1 * r2 < r3 or r1 * r3 < r2 or r2 * r3 < r1:
                    continue
                try:
                    model.mlrank = (r1, r2, r3)
                    value[(r1, r2, r3)] = _metric_value(model.evaluate(metric_type, **evaluate_kwargs), target_metric)
                    model._recommendations = None
                finally:
                    model._mlrank = full               # behind the setter's back, as the reference does
                    model.factors = dict(factors)
    order = sorted(value)
    best = max(order, key=lambda r: (value[r], [-x for x in r]))
    return (best, {r: value[r] for r in order}) if return_scores else best


def random_grid(params, n=60, grid_cache=None, skip_config=None, rng=None):
    """Up to n distinct random points of the grid `params` (name -> list of values): (set of value tuples, names)
    (evaluation/pipelines.py:23-53; `rng`: a numpy RandomState / Generator for reproducible draws — the reference uses
    the `random` module's global state)."""
    if not isinstance(n, int):
        raise TypeError('n must be an integer, not {}'.format(type(n)))
    if n < 0:
        raise ValueError('n should be >= 0')
    names, values = zip(*params.items())
    grid = set(grid_cache) if grid_cache is not None else set()
    max_n = 1
    for vals in values:
        max_n *= len(vals)
    n = min(n if n > 0 else max_n, max_n)
    skipped = set()
    skip_config = skip_config or (lambda config: False)
    if rng is None:
        import random
        pick = random.choice
    else:
        pick = lambda vals: vals[int(rng.randint(len(vals)) if hasattr(rng, 'randint') else rng.integers(len(vals)))]
    while len(grid) < n - len(skipped):
        point = tuple(pick(vals) for vals in values)
        if skip_config(point):
            skipped.add(point)
            continue
        grid.add(point)
    return grid, names


def _params_to_dict(names, params):
    try:
        return dict(zip(names, params))
    except TypeError:                                   # a single parameter
        return {names: params}


def find_optimal_config(model, param_grid, param_names, target_metric, return_scores=False, init_config=None,
                        reset_config=None, verbose=False, force_build=True, metric_type='all', **evaluate_kwargs):
    """Grid search (evaluation/pipelines.py:170-214): for every point of `param_grid` set the attributes, (re)build,
    evaluate.  Returns the best configuration as a dict (first maximum in the order of the grid) and, with
    return_scores=True, the dict point -> value in that order."""
    if init_config:
        for cfg in (init_config if isinstance(init_config, (list, tuple)) else [init_config]):
            set_config(model, cfg)
    quiet, model.verbose = model.verbose, verbose
    value = {}
    try:
        for params in param_grid:
            try:
                set_config(model, _params_to_dict(param_names, params))
                if not model._is_ready or force_build:
                    model.build()
                value[params] = _metric_value(model.evaluate(metric_type, **evaluate_kwargs), target_metric)
            finally:
                if reset_config is not None:
                    if isinstance(reset_config, dict):
                        set_config(model, reset_config)
                    elif callable(reset_config):
                        reset_config(model)
                    else:
                        raise NotImplementedError
    finally:
        model.verbose = quiet
    if not value:
        raise ValueError('find_optimal_config: empty param_grid')
    finite = [p for p in value if value[p] == value[p]]    # pandas' idxmax skips NaN; first maximum in grid order
    if not finite:
        raise ValueError('find_optimal_config: the target metric is NaN at every grid point')
    best = finite[0]
    for p in finite:
        if value[p] > value[best]:
            best = p
    best_config = _params_to_dict(param_names, best)
    return (best_config, value) if return_scores else best_config
