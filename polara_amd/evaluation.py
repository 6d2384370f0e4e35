"""Metrics over a `[n_users x topk]` recommendation array and a holdout — the consumer side of the hot path
(`RecommenderModel.evaluate`, models.py:408-485; formulas of recommender/evaluation.py:90-253), restated on
plain NumPy arrays so that a model built here can be evaluated without Polara installed.  Host code by nature
(the reference's is pandas/SciPy); nothing here is on the timed path.

Inputs follow the reference's conventions: the holdout triplets are sorted by user, every test user has at
least one holdout item, and row r of `recommendations` belongs to the r-th distinct holdout user
(evaluation.py:47-62 builds its row pointers from `np.diff(keys)` under the same assumption).

Holdout entries with feedback exactly 0 (when feedback is used): the reference's hit matrices are sparse products
of the boolean image of the holdout matrix with the rank matrix (evaluation.py:75-84), so such an entry is never
a hit or a miss; with a positive/negative split they also drop out of the per-class counts (`eliminate_zeros`,
evaluation.py:60-73), without one they still count as holdout items (`eval_matrix.getnnz`, evaluation.py:188,131).
Mirrored here through `_Matched.nz`.

One deliberate difference: the reference divides through `np.divide(a, b, where=mask)` without `out=`
(evaluation.py:19-21), so the rows a mask excludes hold UNINITIALISED memory and leak into the means of
recall-type ratios (its own outputs show miss_rate + recall != 1 and NDCG > 1).  Here excluded rows contribute
0, which is what the formulas intend; the metrics that do not pass through that division (hit counts,
precision/recall when every masked numerator is 0, MAP, ARHR, MRR, HR, coverage) are pinned bit-for-bit to the
reference's outputs in tests/golden.
"""
from collections import namedtuple

import numpy as np

Hits = namedtuple('Hits', ['true_positive', 'false_positive', 'true_negative', 'false_negative'])
Relevance = namedtuple('Relevance', ['precision', 'recall', 'fallout', 'specifity', 'miss_rate'])
RelevanceHR = namedtuple('Relevance', ['hr'])
Ranking = namedtuple('Ranking', ['ndcg', 'ndcl', 'map', 'arhr'])
RankingRR = namedtuple('Ranking', ['arhr', 'mrr'])
Experience = namedtuple('Experience', ['coverage'])


def _ratio(a, b, mask):
    out = np.zeros(len(a), dtype=np.float64)
    np.divide(a, b, out=out, where=mask)
    return out


class _Matched:
    """Per holdout entry: the user's row, whether it counts as positive, its relevance, and the rank (1-based)
    at which the item was recommended (0 = not recommended)."""

    def __init__(self, recommendations, holdout_user, holdout_item, holdout_fdbk, is_positive, ranks=None,
                 n_valid_recs=None):
        """ranks / n_valid_recs: precomputed on the device (pk_eval_ranks) when the recommendation array stays
        there; `recommendations` may then be just its shape (n_users, topk)."""
        if ranks is not None:
            self._from_ranks(recommendations, holdout_user, holdout_item, holdout_fdbk, is_positive, ranks, n_valid_recs)
            return
        recs = np.atleast_2d(np.asarray(recommendations))
        users = np.asarray(holdout_user)
        if (np.diff(users) < 0).any():
            raise ValueError('holdout must be sorted by user')
        row = np.r_[0, np.cumsum(np.diff(users) != 0)] if len(users) else np.zeros(0, np.int64)
        if len(users) and row[-1] + 1 != recs.shape[0]:
            raise ValueError('recommendations have %d rows, the holdout %d users' % (recs.shape[0], row[-1] + 1))
        self.n_users, self.topk = recs.shape
        self.recs = recs
        self.row = row
        self.item = np.asarray(holdout_item)
        self.rel = np.ones(len(users)) if holdout_fdbk is None else np.asarray(holdout_fdbk, dtype=np.float64)
        self.positive = np.ones(len(users), bool) if is_positive is None else np.asarray(is_positive, bool)
        self.split = is_positive is not None
        self.nz = self.rel != 0
        # rank of every holdout item in its user's list
        match = recs[row] == self.item[:, None]                      # [n_holdout x topk]
        self.rank = np.where(match.any(axis=1), match.argmax(axis=1) + 1, 0)
        self.n_valid_recs = (recs >= 0).sum(axis=1)                  # models may pad with negative ids

    def _from_ranks(self, shape, holdout_user, holdout_item, holdout_fdbk, is_positive, ranks, n_valid_recs):
        users = np.asarray(holdout_user)
        if (np.diff(users) < 0).any():
            raise ValueError('holdout must be sorted by user')
        self.n_users, self.topk = int(shape[0]), int(shape[1])
        self.recs = None
        self.row = np.r_[0, np.cumsum(np.diff(users) != 0)] if len(users) else np.zeros(0, np.int64)
        if len(users) and self.row[-1] + 1 != self.n_users:
            raise ValueError('recommendations have %d rows, the holdout %d users' % (self.n_users, self.row[-1] + 1))
        self.item = np.asarray(holdout_item)
        self.rel = np.ones(len(users)) if holdout_fdbk is None else np.asarray(holdout_fdbk, dtype=np.float64)
        self.positive = np.ones(len(users), bool) if is_positive is None else np.asarray(is_positive, bool)
        self.split = is_positive is not None
        self.nz = self.rel != 0
        self.rank = np.asarray(ranks).astype(np.int64)
        self.n_valid_recs = (np.full(self.n_users, self.topk) if n_valid_recs is None else np.asarray(n_valid_recs))

    def per_user(self, values, mask):
        return np.bincount(self.row[mask], weights=values[mask] if values is not None else None,
                           minlength=self.n_users).astype(np.float64)


def _relevance_counts(m, not_rated_penalty, per_key):
    hit = m.positive & (m.rank > 0) & m.nz
    tp = m.per_user(None, hit)
    n_recs = m.n_valid_recs.astype(np.float64)
    n_hold = m.per_user(None, np.ones(len(m.row), bool))
    if not m.split:
        fp = not_rated_penalty * (n_recs - tp) if not_rated_penalty > 0 else np.zeros(m.n_users)
        fn = n_hold - tp
        tn = None
    else:
        miss = ~m.positive & (m.rank > 0) & m.nz
        fp = m.per_user(None, miss)
        tn = m.per_user(None, ~m.positive & m.nz) - fp
        fn = m.per_user(None, m.positive & m.nz) - tp
        if not_rated_penalty > 0:
            fp = fp + not_rated_penalty * (n_recs - tp - fp)
    if per_key:
        return tp, fp, tn, fn
    total = lambda x: None if x is None else (x.sum() if np.ndim(x) else x)
    return total(tp), total(fp), total(tn), total(fn)


def get_hits(m, not_rated_penalty):
    tp, fp, tn, fn = _relevance_counts(m, not_rated_penalty, per_key=False)
    # without feedback-based negatives and without a penalty the reference returns the scalar 0 (evaluation.py:190)
    return Hits(tp, fp, tn, fn)


def get_relevance_scores(m, not_rated_penalty):
    tp, fp, tn, fn = _relevance_counts(m, not_rated_penalty, per_key=True)
    precision = _ratio(tp, tp + fp, tp > 0).mean()
    recall = _ratio(tp, tp + fn, tp > 0).mean()
    miss_rate = _ratio(fn, fn + tp, fn > 0).mean()
    fallout = specifity = None
    if tn is not None:
        fallout = _ratio(fp, fp + tn, fp > 0).mean()
        specifity = _ratio(tn, fp + tn, tn > 0).mean()
    return Relevance(precision, recall, fallout, specifity, miss_rate)


def get_hr_score(m):
    return RelevanceHR(m.per_user(None, m.positive & (m.rank > 0) & m.nz).mean())


def _reciprocal_ranks(m):
    hit = m.positive & (m.rank > 0) & m.nz
    rr = np.zeros(len(m.rank))
    rr[hit] = 1.0 / m.rank[hit]
    return hit, rr


def get_arhr_score(m):
    hit, rr = _reciprocal_ranks(m)
    return m.per_user(rr, hit).mean()


def get_mrr_score(m):
    hit, rr = _reciprocal_ranks(m)
    best = np.zeros(m.n_users)
    np.maximum.at(best, m.row[hit], rr[hit])
    return best.mean()


def get_rr_scores(m):
    return RankingRR(get_arhr_score(m), get_mrr_score(m))


def get_map_score(m, topk):
    hit = m.positive & (m.rank > 0) & m.nz
    # precision at the rank of every hit = (hits of that user ranked at or above it) / rank
    order = np.lexsort((m.rank[hit], m.row[hit]))
    rows, ranks = m.row[hit][order], m.rank[hit][order]
    first = np.r_[True, rows[1:] != rows[:-1]] if len(rows) else np.zeros(0, bool)
    pos_in_user = np.arange(len(rows)) - np.maximum.accumulate(np.where(first, np.arange(len(rows)), 0)) + 1
    prec = np.bincount(rows, weights=pos_in_user / ranks, minlength=m.n_users) if len(rows) else np.zeros(m.n_users)
    n_rel = m.per_user(None, np.ones(len(m.row), bool))
    return (prec / np.where(n_rel < topk, n_rel, topk)).mean()


def _ideal_discounts(m):
    """evaluation.py:136-152: within a user, holdout items ordered by `np.argsort(relevance)[::-1]` get the
    discounts 1/log2(2), 1/log2(3), ..."""
    ideal = np.zeros(len(m.row))
    bounds = np.r_[0, np.flatnonzero(np.diff(m.row)) + 1, len(m.row)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        order = np.argsort(m.rel[a:b])[::-1]
        ideal[a + order] = 1.0 / np.log2(np.arange(2, b - a + 2, dtype=np.float64))
    return ideal


def _ndcr(m, mask, rel, discount_sign):
    disc = np.zeros(len(m.rank))
    rec = m.rank > 0
    disc[rec] = 1.0 / np.log2(1.0 + m.rank[rec])
    ideal = _ideal_discounts(m)
    dcr = m.per_user(rel * discount_sign * disc, mask)
    idcr = m.per_user(rel * discount_sign * ideal, mask)
    return _ratio(dcr, idcr, dcr > 0).mean()


def get_ranking_scores(m, topk, switch_positive=None, alternative=False):
    gain = (lambda r: np.exp2(r) - 1.0) if alternative else (lambda r: r)
    ndcg = _ndcr(m, m.positive, gain(m.rel), 1.0)
    ndcl = None
    if m.split:
        ndcl = _ndcr(m, ~m.positive, gain(m.rel - switch_positive), -1.0)
    return Ranking(ndcg, ndcl, get_map_score(m, topk), get_arhr_score(m))


def get_experience_scores(recommendations, n_items):
    return Experience(len(np.unique(recommendations)) / n_items)


# columns of the per-user table of pk_eval_user_metrics (csrc/evalmetrics.hip)
_EV = dict(tp=0, fp=1, tn=2, fn=3, precision=4, recall=5, fallout=6, specifity=7, miss_rate=8, arhr=9, mrr=10, map=11,
           ndcg=12, ndcl=13, n_recs=14, n_hold=15)


def _from_device_sums(device_sums, metric_type, n_items, split, single):
    """The namedtuples of `evaluate` from the sums that left the device: device_sums = (sums float64[16], n_users,
    n_unique_items) — see HipOps.eval_metrics.  Every mean is sum / n_users."""
    sums, n_users, n_unique = device_sums
    g = lambda name: float(sums[_EV[name]])
    mean = lambda name: g(name) / n_users
    scores = []
    if 'relevance' in metric_type:
        if single:
            scores.append(RelevanceHR(mean('tp')))
        else:
            scores.append(Relevance(mean('precision'), mean('recall'), mean('fallout') if split else None,
                                    mean('specifity') if split else None, mean('miss_rate')))
    if 'ranking' in metric_type:
        if single:
            scores.append(RankingRR(mean('arhr'), mean('mrr')))
        else:
            scores.append(Ranking(mean('ndcg'), mean('ndcl') if split else None, mean('map'), mean('arhr')))
    if 'experience' in metric_type:
        scores.append(Experience(n_unique / n_items))
    if 'hits' in metric_type:
        scores.append(Hits(g('tp'), g('fp'), g('tn') if split else None, g('fn')))
    if not scores:
        raise NotImplementedError
    return scores[0] if len(scores) == 1 else scores


def evaluate(recommendations, holdout_user, holdout_item, holdout_fdbk, n_items, metric_type='all', topk=None,
             not_rated_penalty=None, switch_positive=None, ignore_feedback=False, simple_rates=False,
             holdout_size=None, ndcg_alternative=True, device_ranks=None, device_sums=None):
    """models.py:408-485 on arrays.  Returns the same namedtuples in the same order (relevance, ranking,
    experience, hits — whatever the order of `metric_type`); a single family returns the tuple itself."""
    if metric_type == 'all':
        metric_type = ['hits', 'relevance', 'ranking', 'experience']
    if metric_type == 'main':
        metric_type = ['relevance', 'ranking']
    if not isinstance(metric_type, (list, tuple)):
        metric_type = [metric_type]
    if device_sums is not None:
        return _from_device_sums(device_sums, metric_type, n_items, switch_positive is not None and holdout_fdbk is not None,
                                 (holdout_size == 1) or simple_rates)
    if device_ranks is not None:
        # device_ranks = (ranks [n_holdout], n_valid_recs [n_users], (n_users, topk), n_unique_items): everything
        # the metrics need from a recommendation array that stayed on the device
        ranks, n_valid, shape, n_unique = device_ranks
        recs = None
    else:
        recs = np.atleast_2d(np.asarray(recommendations))[:, :topk]
    if (switch_positive is None) or (holdout_fdbk is None):
        not_rated_penalty = 1 if not_rated_penalty is None else not_rated_penalty
        is_positive = None
    else:
        not_rated_penalty = not_rated_penalty or 0
        is_positive = np.asarray(holdout_fdbk) >= switch_positive
    if recs is None:
        m = _Matched(shape, holdout_user, holdout_item, None if ignore_feedback else holdout_fdbk, is_positive,
                     ranks=ranks, n_valid_recs=n_valid)
    else:
        m = _Matched(recs, holdout_user, holdout_item, None if ignore_feedback else holdout_fdbk, is_positive)
    single = (holdout_size == 1) or simple_rates
    scores = []
    if 'relevance' in metric_type:
        scores.append(get_hr_score(m) if single else get_relevance_scores(m, not_rated_penalty))
    if 'ranking' in metric_type:
        scores.append(get_rr_scores(m) if single else
                      get_ranking_scores(m, m.topk, switch_positive, ndcg_alternative))
    if 'experience' in metric_type:
        scores.append(Experience(n_unique / n_items) if recs is None else get_experience_scores(recs, n_items))
    if 'hits' in metric_type:
        scores.append(get_hits(m, not_rated_penalty))
    if not scores:
        raise NotImplementedError
    return scores[0] if len(scores) == 1 else scores
