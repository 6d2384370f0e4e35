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
    return_scores=True, also the dict rank -> value in the order of `ranks`."""
    ranks = list(ranks)
    model_verbose = model.verbose
    model.rank = svd_rank = max(max(ranks), model.rank)
    if not model._is_ready:
        model.verbose = verbose
        model.build()
    if protect_factors:
        svd_factors = dict(**model.factors)      # the truncations below must not eat the full factors
    res = {}
    try:
        for rank in sorted(ranks, key=lambda x: -x):
            model.rank = rank
            res[rank] = _metric_value(model.evaluate(metric_type, **evaluate_kwargs), target_metric)
            model._recommendations = None        # no stale lists across ranks
    finally:
        if protect_factors:
            model._rank = svd_rank
            model.factors = svd_factors
            model._recommendations = None
            if hasattr(model, '_factor_image'):
                model._factor_image = None
        model.verbose = model_verbose
    best_rank = max(ranks, key=lambda r: (res[r], -ranks.index(r)))
    if return_scores:
        return best_rank, {r: res[r] for r in ranks}
    return best_rank
