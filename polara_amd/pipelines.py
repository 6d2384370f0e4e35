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
