"""polara_amd.evaluation (the consumer side: models.py:408-485, evaluation.py:90-253 on arrays) against the
reference's own `model.evaluate()` outputs stored in tests/golden (made by tests/golden/make_golden.py).

Pinned bit-for-bit: everything that does not pass through the reference's `safe_divide` with excluded rows —
hit counts, precision, recall (their excluded rows are exactly 0 anyway), MAP, ARHR, MRR, HR, coverage.
NOT pinned: miss_rate, NDCG, NDCL, fallout, specifity — the reference leaves excluded rows uninitialised
(`np.divide(..., where=mask)` without `out=`, evaluation.py:19-21) and its own numbers show it (NDCG = 8.2 on
the `svd_nofilter` fixture, miss_rate + recall != 1); for those the intended formulas are checked instead."""
import numpy as np
import pytest

from conftest import load_golden, GoldenData
from polara_amd import evaluation as ev

FIXTURES = ['svd_warm', 'svd_known', 'svd_fewunseen', 'svd_nofilter', 'svd_scaled']


def _run(g, **kw):
    per_user = np.bincount(np.unique(g['holdout_user'], return_inverse=True)[1])
    n_items = int(g['train_shape'][1])
    return ev.evaluate(g['recs'], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'], n_items,
                       holdout_size=int(per_user.max()), **kw), n_items


@pytest.mark.parametrize('name', FIXTURES)
def test_metrics_match_the_reference_outputs(name):
    g = load_golden(name)
    scores, n_items = _run(g)
    got = {}
    for tup in scores:
        for k, v in tup._asdict().items():
            got['metric_%s_%s' % (type(tup).__name__, k)] = np.nan if v is None else float(v)
    ref = {k: float(g[k]) for k in g.files if k.startswith('metric_')}
    assert set(got) == set(ref)                                   # same families, same field names
    # every metric is pinned: the fixtures were generated with the reference's `safe_divide` given a zero-initialised
    # output (make_golden.py), i.e. they hold what its formulas define instead of uninitialised memory
    for k, v in ref.items():
        assert (np.isnan(v) and np.isnan(got[k])) or np.isclose(got[k], v, rtol=1e-13, atol=0), (k, got[k], v)
    if 'metric_Relevance_recall' in got:                          # the intended values of the unpinned ones
        assert np.isclose(got['metric_Relevance_miss_rate'], 1.0 - got['metric_Relevance_recall'], atol=1e-12)
        assert 0.0 <= got['metric_Ranking_ndcg'] <= 1.0 + 1e-12
    # order of the families is the reference's, whatever the order asked for
    names = [type(t).__name__ for t in ev.evaluate(g['recs'], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'],
                                                   n_items, metric_type=['hits', 'experience', 'ranking', 'relevance'],
                                                   holdout_size=3)]
    assert names == ['Relevance', 'Ranking', 'Experience', 'Hits']


def _as_dict(scores, prefix):
    got = {}
    for tup in (scores if isinstance(scores, (list, tuple)) and not hasattr(scores, '_fields') else [scores]):
        for k, v in tup._asdict().items():
            got[prefix + type(tup).__name__ + '_' + k] = np.nan if v is None else float(v)
    return got


def _assert_pinned(got, g, prefix, rtol=1e-13):
    ref = {k: float(g[k]) for k in g.files if k.startswith(prefix)}
    assert ref and set(got) == set(ref), (sorted(got), sorted(ref))
    for k, v in ref.items():
        assert (np.isnan(v) and np.isnan(got[k])) or np.isclose(got[k], v, rtol=rtol, atol=0), (k, got[k], v)


@pytest.mark.parametrize('name', FIXTURES)
def test_split_and_at_k_metrics_match_the_reference_outputs(name):
    """the reference's evaluate(switch_positive=4) and evaluate(topk=3) on the same lists (`metricsp4_*`, `metricat3_*`):
    the positive / negative split of evaluation.py:176-205 and the @k roll-back of models.py:441-447"""
    g = load_golden(name)
    scores, _ = _run(g, switch_positive=4)
    _assert_pinned(_as_dict(scores, 'metricsp4_'), g, 'metricsp4_')
    scores, _ = _run(g, topk=3)
    _assert_pinned(_as_dict(scores, 'metricat3_'), g, 'metricat3_')


def test_switch_positive_splits_hits_and_misses():
    g = load_golden('svd_warm')
    n_items = int(g['train_shape'][1])
    rel, rank, hits = ev.evaluate(g['recs'], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'], n_items,
                                  metric_type=['relevance', 'ranking', 'hits'], switch_positive=4, holdout_size=3)
    pos = g['holdout_fdbk'] >= 4
    row = np.unique(g['holdout_user'], return_inverse=True)[1]
    in_recs = (g['recs'][row] == g['holdout_item'][:, None]).any(axis=1)
    assert hits.true_positive == (pos & in_recs).sum() and hits.false_positive == (~pos & in_recs).sum()
    assert hits.false_negative == (pos & ~in_recs).sum() and hits.true_negative == (~pos & ~in_recs).sum()
    assert rel.fallout is not None and 0.0 <= rel.fallout <= 1.0 and 0.0 <= rel.specifity <= 1.0
    assert rank.ndcl is not None and 0.0 <= rank.ndcl <= 1.0 + 1e-12
    # single metric family -> the tuple itself; @k roll-back through topk
    hr5 = ev.evaluate(g['recs'], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'], n_items,
                      metric_type='relevance', topk=5, simple_rates=True)
    hr10 = ev.evaluate(g['recs'], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'], n_items,
                       metric_type='relevance', simple_rates=True)
    assert hr5.hr <= hr10.hr
    with pytest.raises(ValueError):
        ev.evaluate(g['recs'][:-1], g['holdout_user'], g['holdout_item'], g['holdout_fdbk'], n_items)


def test_model_evaluate_without_polara():
    """RecommenderModel.evaluate on ArrayData goes through polara_amd.evaluation: same numbers as the reference's
    evaluate() on the same recommendations (CPU double of the device ops)."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import SVDModel
    g = load_golden('svd_known')
    idx = g['train_idx']
    shp = tuple(int(x) for x in g['train_shape'])
    hold = (g['holdout_user'], g['holdout_item'], g['holdout_fdbk'])
    d = ArrayData((idx[:, 0], idx[:, 1], g['train_val']), n_users=shp[0], n_items=shp[1], holdout=hold,
                  warm_start=False, holdout_size=3)
    m = SVDModel(d, ops=NumpyOps())
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    m._recommendations = g['recs'].copy()          # the reference's own lists: metrics are then comparable 1:1
    m._is_ready = True
    hits = m.evaluate('hits')
    assert (hits.true_positive, hits.false_positive, hits.false_negative) == (
        g['metric_Hits_true_positive'], g['metric_Hits_false_positive'], g['metric_Hits_false_negative'])
    rel, rank = m.evaluate('main')
    assert np.isclose(rel.precision, g['metric_Relevance_precision'], rtol=1e-13)
    assert np.isclose(rank.map, g['metric_Ranking_map'], rtol=1e-13) and np.isclose(rank.arhr, g['metric_Ranking_arhr'], rtol=1e-13)


def _model_on_arrays(g, ops):
    """The fixture's exact training / test triplets (GoldenData) + its holdout: rows of the recommendation array are
    the fixture's test users, in the fixture's order."""
    from polara_amd.models import SVDModel
    d = GoldenData(g)
    d.set_test_data(holdout=(g['holdout_user'], g['holdout_item'], g['holdout_fdbk']), notify=False)
    d.holdout_size = 3
    d.warm_start = False
    m = SVDModel(d, ops=ops)
    m.verbose = False
    m.rank, m.topk = int(g['rank']), int(g['topk'])
    return m


def _check_device_ranks_path(ops):
    """evaluate() after a real build + get_recommendations: the ranks come from the device-resident list
    (pk_eval_ranks); every metric equals the host computation on the returned array, and the reference's."""
    g = load_golden('svd_known')
    m = _model_on_arrays(g, ops)
    m.build()
    recs = m.recommendations
    assert m._recs_dev is not None and m._recs_dev[0] is recs
    dev = m.evaluate('all')
    order = np.argsort(g['holdout_user'], kind='stable')
    host = ev.evaluate(recs, g['holdout_user'][order], g['holdout_item'][order], g['holdout_fdbk'][order],
                       int(g['train_shape'][1]), holdout_size=3)
    for a, b in zip(dev, host):
        assert type(a).__name__ == type(b).__name__
        for x, y in zip(a, b):
            assert (x is None and y is None) or np.isclose(x, y, rtol=1e-12, atol=0), (type(a).__name__, x, y)
    notie = g['boundary_gap'] > 0
    if notie.all():
        assert np.isclose(dev[3].true_positive, g['metric_Hits_true_positive'])
    at5_dev, at5_host = m.evaluate('relevance', topk=5), ev.evaluate(recs, g['holdout_user'][order], g['holdout_item'][order],
                                                                   g['holdout_fdbk'][order], int(g['train_shape'][1]),
                                                                   metric_type='relevance', topk=5, holdout_size=3)
    assert np.isclose(at5_dev.precision, at5_host.precision, rtol=1e-12)


def test_model_evaluate_device_ranks_cpu_double():
    from numpy_ops import NumpyOps
    _check_device_ranks_path(NumpyOps())


@pytest.mark.gpu
def test_model_evaluate_device_ranks(hip_ops):
    _check_device_ranks_path(hip_ops)


@pytest.mark.gpu
@pytest.mark.parametrize('name', FIXTURES)
def test_device_metrics_are_pinned_to_the_reference(hip_ops, name):
    """evalmetrics.hip against the REFERENCE's evaluate() outputs (evaluation.py:90-253), not against this
    package's host formulas.  (1) The reference's own lists (`recs` of the fixture) are put on the device in the
    model's internal id space and reduced there: every `metric_*`, `metricsp4_*` (switch_positive = 4) and `metricat3_*`
    (@3) value of the fixture must come back to 1e-12 — MAP normaliser, NDCG / NDCL tie order, zero-feedback holdout
    entries included.  (2) The model's own lists on the HIP backend: where no test user has a boundary tie they ARE
    the reference's lists and `evaluate('all')` is pinned end to end; otherwise the hit count may differ by at most the
    tied users' holdout items."""
    import torch
    from polara_amd.models import ScaledSVD, SVDModel
    g = load_golden(name)
    d = GoldenData(g)
    d.set_test_data(holdout=(g['holdout_user'], g['holdout_item'], g['holdout_fdbk']), notify=False)
    per_user = np.bincount(np.unique(g['holdout_user'], return_inverse=True)[1])
    d.holdout_size = int(per_user.max())
    d.warm_start = False
    m = (ScaledSVD if 'col_scaling' in g.files else SVDModel)(d, ops=hip_ops)
    if 'col_scaling' in g.files:
        m.col_scaling, m.row_scaling = float(g['col_scaling']), float(g['row_scaling'])
    m.verbose = False
    m.rank, m.topk, m.filter_seen = int(g['rank']), int(g['topk']), bool(g['filter_seen'])
    m.build()
    own = m.recommendations
    assert m._recs_dev is not None and m._recs_dev[0] is own
    own_scores = m.evaluate('all')
    notie = g['boundary_gap'] > 0
    assert np.array_equal(own[notie], g['recs'][notie])
    if notie.all():
        _assert_pinned(_as_dict(own_scores, 'metric_'), g, 'metric_', rtol=1e-12)
    else:
        tp = [t for t in own_scores if type(t).__name__ == 'Hits'][0].true_positive
        assert abs(tp - float(g['metric_Hits_true_positive'])) <= int((~notie).sum()) * d.holdout_size
    # (1) the reference's lists, reduced on the device
    ref_recs = np.ascontiguousarray(g['recs'])
    m._recommendations = ref_recs
    m._recs_dev = (ref_recs, torch.from_numpy(m._item_rank[ref_recs].astype(np.int64)).to(hip_ops.device))
    for prefix, kw in (('metric_', {}), ('metricsp4_', dict(switch_positive=4)), ('metricat3_', dict(topk=3))):
        scores = m.evaluate('all', **kw)
        assert m._recs_dev is not None and m._recs_dev[0] is ref_recs        # still the device path
        _assert_pinned(_as_dict(scores, prefix), g, prefix, rtol=1e-12)


def test_find_optimal_svd_rank_one_build():
    """pipelines.find_optimal_svd_rank: one build at the largest rank, truncation for the others — every rank's
    metric equals that of a model built at that rank from scratch (PureSVD factors are nested)."""
    from numpy_ops import NumpyOps
    from polara_amd.pipelines import find_optimal_svd_rank
    g = load_golden('svd_known')
    m = _model_on_arrays(g, NumpyOps())
    ranks = [3, 5, 8, int(g['rank'])]
    best, table = find_optimal_svd_rank(m, ranks, 'precision', return_scores=True, metric_type='relevance')
    assert len(m.training_time) == 1 and best in ranks and table[best] == max(table.values())
    assert m.rank == max(ranks) and m.factors['singular_values'].shape == (max(ranks),)   # factors protected
    for r in (3, 8):
        fresh = _model_on_arrays(g, NumpyOps())
        fresh.rank = r
        fresh.build()
        m.rank = r
        same = (m.recommendations == fresh.recommendations).all(axis=1)
        # this fixture zeroes the below-threshold test feedback: ~4 % of the users have an all-zero profile, i.e.
        # all-tie scores whose top-k is implementation-defined (and depends on the internal item order, which
        # follows the factor norms of whichever rank was BUILT); every other row must agree exactly
        assert same[g['boundary_gap'] > 0].all() and same.mean() > 0.9
        assert abs(fresh.evaluate('relevance').precision - table[r]) < 0.01
    m.rank = max(ranks)
    assert find_optimal_svd_rank(m, ranks, lambda t: -t['miss_rate'], metric_type='relevance') in ranks


def test_array_data_infers_holdout_size_from_the_holdout():
    """a 3-items-per-user holdout must not take the holdout_size == 1 (HR / MRR) branch of evaluate()."""
    from polara_amd.data import ArrayData
    u = np.repeat(np.arange(5), 4)
    tr = (u, np.tile(np.arange(4), 5), np.ones(20))
    hold3 = (np.repeat(np.arange(5), 3), np.tile(np.arange(4, 7), 5), np.ones(15))
    d = ArrayData(tr, n_users=5, n_items=10, holdout=hold3)
    assert d.holdout_size == 3
    d.set_test_data(holdout=(np.arange(5), np.full(5, 7), np.ones(5)))
    assert d.holdout_size == 1
    assert ArrayData(tr, n_users=5, n_items=10).holdout_size == 0
    assert ArrayData(tr, n_users=5, n_items=10, holdout=hold3, holdout_size=3).holdout_size == 3
    with pytest.raises(ValueError):
        ArrayData(tr, n_users=5, n_items=10, holdout=hold3, holdout_size=1)
    # a list (not an ndarray) of recommendations is accepted (NumPy 2: np.array(copy=False) would raise)
    r = ev.evaluate([[4, 5, 0], [6, 1, 2], [0, 1, 2], [5, 4, 6], [9, 8, 7]], hold3[0], hold3[1], hold3[2], 10,
                    metric_type='hits', holdout_size=3)
    assert r.true_positive == 2 + 1 + 0 + 3 + 0


@pytest.mark.gpu
@pytest.mark.parametrize('switch_positive', [None, 3.0])
@pytest.mark.parametrize('ignore_feedback', [False, True])
def test_device_metric_reductions_equal_the_host_formulas(hip_ops, switch_positive, ignore_feedback):
    """pk_eval_user_metrics + pk_eval_reduce + pk_unique_count_i64 against polara_amd.evaluation on the same arrays:
    ragged holdouts (1..9 items, some users with none recommended), explicit zero feedback, padded (-1) lists, @k."""
    import torch
    rng = np.random.RandomState(11)
    n_users, n_items, topk = 5000, 700, 20
    recs = np.stack([rng.permutation(n_items)[:topk] for _ in range(n_users)]).astype(np.int64)
    recs[rng.rand(n_users) < 0.05, -3:] = -1                       # models may pad short lists with negative ids
    per = rng.randint(1, 10, n_users)
    hu = np.repeat(np.arange(n_users), per)
    # up to three recommended items (never a padded slot) + items outside the list: no duplicates within a user
    hi_ = np.concatenate([np.r_[r[rng.permutation(topk - 3)[:min(p, 3)]],
                                rng.permutation(np.setdiff1d(np.arange(n_items), r))[:max(p - 3, 0)]] for r, p in zip(recs, per)])
    hf = rng.randint(0, 6, len(hu)).astype(np.float64)
    ptr = np.r_[0, np.cumsum(per)].astype(np.int64)
    rd = torch.from_numpy(recs).to(hip_ops.device)
    for k in (topk, 7, 1):
        split = switch_positive is not None
        penalty = 0 if split else 1
        sums = hip_ops.eval_metrics(rd, k, torch.from_numpy(ptr), torch.from_numpy(hi_),
                                    None if ignore_feedback else torch.from_numpy(hf),
                                    torch.from_numpy((hf >= switch_positive).astype(np.uint8)) if split else None,
                                    not_rated_penalty=penalty, switch_positive=switch_positive or 0.0, alternative=True)
        n_unique = hip_ops.unique_count(rd[:, :k].contiguous(), n_items)
        got = ev.evaluate(None, hu, hi_, hf, n_items, metric_type='all', topk=k, switch_positive=switch_positive,
                          ignore_feedback=ignore_feedback, holdout_size=9, device_sums=(sums, n_users, n_unique))
        want = ev.evaluate(recs, hu, hi_, hf, n_items, metric_type='all', topk=k, switch_positive=switch_positive,
                           ignore_feedback=ignore_feedback, holdout_size=9)
        for a, b in zip(got, want):
            assert type(a).__name__ == type(b).__name__
            for name, x, y in zip(a._fields, a, b):
                assert (x is None and y is None) or np.isclose(x, y, rtol=1e-12, atol=1e-15), (k, type(a).__name__, name, x, y)
        single = ev.evaluate(None, hu, hi_, hf, n_items, metric_type=['relevance', 'ranking'], topk=k, simple_rates=True,
                             switch_positive=switch_positive, device_sums=(sums, n_users, n_unique))
        single_w = ev.evaluate(recs, hu, hi_, hf, n_items, metric_type=['relevance', 'ranking'], topk=k, simple_rates=True,
                               switch_positive=switch_positive, ignore_feedback=ignore_feedback)
        assert np.isclose(single[0].hr, single_w[0].hr, rtol=1e-13) and np.isclose(single[1].mrr, single_w[1].mrr, rtol=1e-12)


@pytest.mark.gpu
def test_rank_sweep_on_the_device(hip_ops):
    """pipelines.find_optimal_svd_rank (evaluation/pipelines.py:81-116) on the HIP backend: one build at the largest
    rank, truncation + device re-imaging per rank, every metric reduced on the device — equal to evaluating the
    returned lists on the host, rank by rank, and the factors are put back afterwards."""
    from polara_amd.data import ArrayData
    from polara_amd.models import SVDModel
    from polara_amd.pipelines import find_optimal_svd_rank
    from polara_amd.synth import planted_csr, csr_to_coo_triplets
    u, i, v = csr_to_coo_triplets(planted_csr(4000, 900, 40, 12, seed=23, min_items=12, max_items=200))
    rng = np.random.RandomState(1)
    last = np.r_[np.flatnonzero(np.diff(u)), len(u) - 1]            # hold out 3 interactions of every user
    hold = np.sort(np.concatenate([last, last - 1, last - 2]))
    keep = np.setdiff1d(np.arange(len(u)), hold)
    d = ArrayData((u[keep], i[keep], v[keep]), n_users=4000, n_items=900, holdout=(u[hold], i[hold], v[hold]), warm_start=False)
    m = SVDModel(d, ops=hip_ops)
    m.verbose = False
    m.topk = 10
    ranks = [40, 5, 24, 12, 3]
    best, table = find_optimal_svd_rank(m, ranks, 'precision', return_scores=True, metric_type='relevance')
    assert m.rank == 40 and m.factors[d.fields.itemid].shape[1] == 40 and len(m.training_time) == 1     # one build, restored
    hu, hi_, hf = d.test.holdout
    for r in sorted(ranks, reverse=True):                              # descending: every step is a truncation, no rebuild
        m.rank = r
        want = ev.evaluate(m.recommendations, hu, hi_, hf, 900, metric_type='relevance', holdout_size=3).precision
        assert np.isclose(table[r], want, rtol=1e-12), (r, table[r], want)
    assert len(m.training_time) == 1
    assert best == max(ranks, key=lambda r: (table[r], r))
    assert table[24] > table[3]


def _coffee_sweep(ops):
    from conftest import GoldenData
    from polara_amd.models import CoffeeModel
    from polara_amd.pipelines import find_optimal_tucker_ranks
    g = load_golden('coffee_small')
    d = GoldenData(g)
    d.set_test_data(holdout=(g['hold_user'], g['hold_item'], np.ones(len(g['hold_user']))), notify=False)
    d.holdout_size = 1
    d.warm_start = False
    m = CoffeeModel(d, ops=ops)
    m.verbose = False
    m.topk, m.seed = int(g['topk']), 0
    grid = [[3, 6], [2, 5], [2, 3]]
    best, table = find_optimal_tucker_ranks(m, grid, 'true_positive', return_scores=True, metric_type='hits')
    return m, g, grid, best, table


def _check_coffee_sweep(ops):
    m, g, grid, best, table = _coffee_sweep(ops)
    assert list(table) == sorted(table) and len(table) == 7 and (6, 2, 2) not in table and best == max(sorted(table), key=lambda r: (table[r], [-x for x in r]))
    assert m.mlrank == (6, 5, 3) and len(m.training_time) == 1                      # one HOOI build, full factors back
    f = m.data.fields
    assert m.factors[f.userid].shape[1] == 6 and m.factors['core'].shape == (6, 5, 3)
    hu, hi_ = g['hold_user'], g['hold_item']
    full = dict(m.factors)
    for r in [(3, 2, 2), (6, 5, 2)]:                                                 # a sweep value = evaluating that rank's own lists
        m.mlrank = r
        m._recommendations = None
        want = ev.evaluate(m.recommendations, hu, hi_, np.ones(len(hu)), int(g['train_shape'][1]), metric_type='hits', holdout_size=1)
        assert table[r] == want.true_positive, (r, table[r], want)
        m._mlrank, m.factors, m._recommendations = (6, 5, 3), dict(full), None
    assert len(m.training_time) == 1
    return table


def test_tucker_rank_sweep_one_build():
    """pipelines.find_optimal_tucker_ranks (evaluation/pipelines.py:119-160) on the NumPy double of the ops."""
    from numpy_ops import NumpyOps
    _check_coffee_sweep(NumpyOps())


@pytest.mark.gpu
def test_tucker_rank_sweep_on_the_device(hip_ops):
    """The same sweep on the HIP backend (HOOI build, core rounding per rank, scoring and hit counts on the device):
    the same table as the NumPy double of the ops, up to a hit per near-tied list."""
    from numpy_ops import NumpyOps
    dev = _check_coffee_sweep(hip_ops)
    host = _coffee_sweep(NumpyOps())[4]
    assert list(dev) == list(host)
    assert max(abs(dev[r] - host[r]) for r in dev) <= 2, (dev, host)
