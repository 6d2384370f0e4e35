"""Generates the golden fixtures under tests/golden/*.npz from the REFERENCE ITSELF.

Runs only in the build container: imports evfro/polara from /root/reference (read-only) through the
test-only numba shim in tests/golden/_numba_shim (numba is not installed in this image), drives
the unmodified `RecommenderData` / `SVDModel` / `CoffeeModel` on small seeded synthetic inputs, and
stores inputs + expected outputs.  Before a fixture is written, the oracle (oracle/polara_oracle.py)
is run on the same inputs with the same NumPy seed and asserted BIT-EQUAL to the reference — this is
what pins the oracle.  The reference cannot travel to the GPU box; only these vectors do.

usage:  python tests/golden/make_golden.py
"""
import os
import sys
import io
import contextlib
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, '_numba_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

import numpy as np
import pandas as pd

import polara  # the reference
from polara import RecommenderData, SVDModel
from polara.recommender.models import CoffeeModel, RecommenderModel, ScaledSVD
from polara.recommender import utils as ref_utils
from polara.lib import tensor as ref_tensor
from polara.lib import sparse as ref_sparse

from oracle import polara_oracle as orc
from polara_amd.synth import planted_csr, csr_to_coo_triplets

# The reference's `safe_divide` (recommender/evaluation.py:19-21) calls `np.divide(a, b, where=mask)` WITHOUT `out=`:
# the rows the mask excludes come out of uninitialised memory, so its precision / recall / miss_rate / NDCG / NDCL /
# fallout / specifity differ from run to run (NDCG = 8.2, 6.7, 14.1 ... on the same lists) and a fixture holding them
# could not be reproduced.  For the generation only, the reference's metric code runs with that one function given a
# zero-initialised output — in memory, nothing of the reference is copied or changed on disk — so that every metric
# stored below is what its formulas define and this script reproduces its fixtures byte for byte.
import polara.recommender.evaluation as _ref_evaluation


def _safe_divide_zero_init(a, b, mask=None, dtype=None):
    pos = mask if mask is not None else a > 0
    out = np.zeros(np.broadcast(np.asarray(a), np.asarray(b)).shape, dtype=dtype or np.float64)
    return np.divide(a, b, out=out, where=pos)


_ref_evaluation.safe_divide = _safe_divide_zero_init


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def frame(n_users, n_items, mean_items, rank, levels, seed, **kw):
    csr = planted_csr(n_users, n_items, mean_items, rank, levels=levels, seed=seed, **kw)
    u, i, v = csr_to_coo_triplets(csr)
    return pd.DataFrame({'userid': u, 'itemid': i, 'rating': v})


def metrics_to_dict(scores, prefix='metric_'):
    out = {}
    for s in scores:
        for name, val in s._asdict().items():
            out[prefix + type(s).__name__ + '_' + name] = np.float64(val)
    return out


def svd_fixture(name, df, data_cfg, rank, topk, filter_seen=True, feedback_threshold=None, seed=0,
                extra_ranks=(), scaled=None):
    data = RecommenderData(df, 'userid', 'itemid', 'rating', seed=seed)
    data.verbose = False
    for k, v in data_cfg.items():
        setattr(data, k, v)
    quiet(data.prepare)
    model = (ScaledSVD if scaled else SVDModel)(data, feedback_threshold=feedback_threshold)
    model.verbose = False
    if scaled:
        model.col_scaling, model.row_scaling = scaled
    model.rank = rank
    model.topk = topk
    model.filter_seen = filter_seen
    np.random.seed(seed)
    quiet(model.build)
    recs = model.get_recommendations()
    userid, itemid = data.fields.userid, data.fields.itemid
    V = model.factors[itemid]
    sigma = model.factors['singular_values']

    # inputs as the hot path sees them
    idx, val, shp = data.to_coo(tensor_mode=False, feedback_threshold=model.feedback_threshold)
    (tu, ti, tf), tshape, test_users = model._get_test_data()

    # ---- pin the oracle: same calls, same seed -> bit-equal --------------------------------
    if scaled:
        A = orc.scaled_training_matrix(idx, val, shp, col_scaling=scaled[0], row_scaling=scaled[1])
    else:
        A = orc.get_training_matrix(idx, val, shp, dtype=np.float64)
    np.random.seed(seed)
    _, o_sigma, o_V = orc.svd_build(A, rank)
    assert np.array_equal(o_sigma, sigma), name
    assert np.array_equal(o_V, V), name
    o_recs, o_scores = orc.svd_recommendations(o_V, (tu, ti, tf), tshape, topk, filter_seen,
                                               return_scores=True)
    assert np.array_equal(o_recs, recs), name

    # dense scores of a few users for score-tolerance checks (reference `_user_scores` path is the
    # same slice_recommendations + downvote; we store the raw un-downvoted scores)
    probe_users = np.unique(np.linspace(0, tshape[0] - 1, 6).astype(np.int64))
    probe_scores = np.stack([model.slice_recommendations((tu, ti, tf), tshape, int(u), int(u) + 1)[0][0]
                             for u in probe_users])
    # boundary-tie flags (reference result implementation-defined on those rows)
    full_scores, sd = orc.svd_slice_recommendations(o_V, (tu, ti, tf), tshape, 0, tshape[0])
    if filter_seen:
        orc.downvote_seen_items(full_scores, sd)
    gap = orc.boundary_gap(full_scores, topk)

    out = dict(train_idx=idx.astype(np.int64), train_val=val, train_shape=np.array(shp, np.int64),
               test_user=tu.astype(np.int64), test_item=ti.astype(np.int64), test_fdbk=np.asarray(tf, np.float64),
               test_shape=np.array(tshape, np.int64), test_users=np.asarray(test_users, np.int64),
               rank=np.int64(rank), topk=np.int64(topk), filter_seen=np.bool_(filter_seen),
               sigma=sigma, V=np.ascontiguousarray(V), recs=recs, rec_scores=o_scores,
               probe_users=probe_users, probe_scores=probe_scores, boundary_gap=gap,
               seed=np.int64(seed))
    if scaled:
        out['col_scaling'], out['row_scaling'] = np.float64(scaled[0]), np.float64(scaled[1])
    if data.test.holdout is not None:
        h = data.test.holdout
        out['holdout_user'] = h[userid].values.astype(np.int64)
        out['holdout_item'] = h[itemid].values.astype(np.int64)
        out['holdout_fdbk'] = h['rating'].values.astype(np.float64)
        out.update(metrics_to_dict(quiet(model.evaluate)))
        # the same lists with the positive / negative split of the holdout (evaluation.py:176-205: fallout, specifity,
        # NDCL and the true-negative count only exist then) and rolled back to @3 (models.py:441-447)
        out.update(metrics_to_dict(quiet(lambda: model.evaluate(switch_positive=4)), prefix='metricsp4_'))
        out.update(metrics_to_dict(quiet(lambda: model.evaluate(topk=3)), prefix='metricat3_'))
    # rank truncation contract (models.py:812-832): smaller rank = column prefix, no rebuild
    for r in extra_ranks:
        model.rank = r
        assert model._is_ready
        out['recs_rank%d' % r] = model.get_recommendations()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'train nnz', len(val), 'test users', tshape[0], 'ties', int((gap == 0).sum()))


def coffee_fixture(name, df, data_cfg, mlrank, topk, seed=0, num_iters=25, growth_tol=0.0001):
    data = RecommenderData(df, 'userid', 'itemid', 'rating', seed=seed)
    data.verbose = False
    for k, v in data_cfg.items():
        setattr(data, k, v)
    quiet(data.prepare)
    model = CoffeeModel(data)
    model.verbose = False
    model.mlrank = mlrank
    model.topk = topk
    model.seed = seed
    model.num_iters = num_iters
    model.growth_tol = growth_tol
    model._vectorize_target = 'cpu'
    np.random.seed(seed)
    quiet(model.build)
    recs = model.get_recommendations()
    userid, itemid, feedback = data.fields
    u0, u1, u2 = (model.factors[f] for f in (userid, itemid, feedback))
    core = model.factors['core']

    idx, val, shp = data.to_coo(tensor_mode=True)
    (tu, ti, tf), tshape, test_users = model._get_test_data()

    # ---- pin the oracle ---------------------------------------------------------------------
    trace = []
    np.random.seed(seed)
    o0, o1, o2, og = orc.hooi(idx, val, shp, mlrank, growth_tol=growth_tol, num_iters=num_iters,
                              seed=seed, trace=trace)
    for a, b in ((o0, u0), (o1, u1), (o2, u2), (og, core)):
        assert np.allclose(a, b, rtol=0, atol=1e-11), name   # np.add.at vs scalar loop: same order
    # TTM restatement vs the reference's own dttm_seq on one call (bit-equal expected)
    ref_res = ref_tensor.ttm3d_seq(idx, val, shp, u2, u1, ((2, 0), (1, 0)))
    orc_res = orc.ttm3d_seq(idx, val, shp, u2, u1, ((2, 0), (1, 0)))
    assert np.array_equal(ref_res, orc_res), name
    o_recs = orc.coffee_recommendations(u1, u2, (tu, ti, tf), tshape, topk, True,
                                        flattener=model.flattener)
    assert np.array_equal(o_recs, recs), name

    full_scores, sd = orc.coffee_slice_recommendations(u1, u2, (tu, ti, tf), tshape, 0, tshape[0],
                                                       model.flattener)
    raw_probe = full_scores[:4].copy()
    orc.downvote_seen_items(full_scores, sd)
    gap = orc.boundary_gap(full_scores, topk)
    # ---- the CoffeeModel extras (models.py:1027-1092): unfolded test slice, holdout slice, feedback prediction ----
    extras = {}
    a, b = 3, min(40, tshape[0])
    for mode in (0, 1, 2):
        ref_unf, ref_sl = model.unfold_test_tensor_slice((tu, ti, tf), tshape, a, b, mode)
        o_unf, o_sl = orc.unfold_test_tensor_slice((tu, ti, tf), tshape, a, b, mode)
        assert ref_unf.shape == o_unf.shape and (ref_unf != o_unf).nnz == 0 and all(np.array_equal(x, y) for x, y in zip(ref_sl, o_sl)), name
        ref_unf.sum_duplicates()
        ref_unf.sort_indices()
        extras['unfold%d_indptr' % mode] = ref_unf.indptr.astype(np.int64)
        extras['unfold%d_indices' % mode] = ref_unf.indices.astype(np.int64)
        extras['unfold%d_data' % mode] = ref_unf.data.astype(np.int64)
        extras['unfold%d_shape' % mode] = np.array(ref_unf.shape, np.int64)
    extras['unfold_range'] = np.array([a, b], np.int64)
    hold = data.test.holdout
    hu, hi = hold[userid].values.astype(np.int64), hold[itemid].values.astype(np.int64)
    ref_hs = model.get_holdout_slice(a, b)
    o_hs = orc.get_holdout_slice(hu, hi, a, b)
    assert all(np.array_equal(x, y) for x, y in zip(ref_hs, o_hs)), name
    extras.update(hold_user=hu, hold_item=hi, hold_slice_user=ref_hs[0], hold_slice_item=ref_hs[1])
    if not data.warm_start:
        ref_pred = model.predict_feedback()
        levels = data.index.feedback.sort_values('new')['old'].values
        o_idx, o_scores = orc.coffee_predict_feedback(u0, u1, u2, core, hu, hi)
        assert np.array_equal(levels[o_idx], ref_pred), name
        srt = np.sort(o_scores, axis=1)
        extras.update(feedback_levels=levels.astype(np.float64), predicted_feedback=np.asarray(ref_pred, dtype=np.float64),
                      predicted_level=o_idx.astype(np.int64), predict_gap=srt[:, -1] - srt[:, -2])
    out = dict(train_idx=idx.astype(np.int64), train_val=val, train_shape=np.array(shp, np.int64),
               test_user=tu.astype(np.int64), test_item=ti.astype(np.int64), test_fdbk=tf.astype(np.int64),
               test_shape=np.array(tshape, np.int64), mlrank=np.array(mlrank, np.int64),
               topk=np.int64(topk), seed=np.int64(seed), num_iters=np.int64(num_iters),
               growth_tol=np.float64(growth_tol),
               u0=u0, u1=u1, u2=u2, core=np.ascontiguousarray(core), core_norm_trace=np.array(trace),
               ttm_mode0=ref_res, recs=recs, probe_scores=raw_probe, boundary_gap=gap, **extras)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'nnz', len(val), 'iters', len(trace), 'test users', tshape[0], 'ties', int((gap == 0).sum()))


def micro_fixtures():
    rng = np.random.RandomState(7)
    out = {}
    # downvote + topsort on dense blocks, incl. a user with fewer than k unseen items
    scores = rng.randn(9, 14)
    users = np.r_[np.repeat(0, 12), np.repeat(3, 4), np.repeat(8, 2)]
    items = np.r_[rng.permutation(14)[:12], rng.permutation(14)[:4], rng.permutation(14)[:2]]
    s_ref = scores.copy()
    RecommenderModel.downvote_seen_items(s_ref, (users, items, np.ones(len(users))))
    s_orc = scores.copy()
    orc.downvote_seen_items(s_orc, (users, items, np.ones(len(users))))
    assert np.array_equal(s_ref, s_orc)
    top_ref = np.apply_along_axis(RecommenderModel.topsort, 1, s_ref, 5)
    assert np.array_equal(top_ref, orc.get_topk_elements(s_orc, 5))
    out.update(dv_scores=scores, dv_users=users, dv_items=items, dv_lowered=s_ref, dv_top5=top_ref)
    # single-user path (models.py:513-515 ValueError fallback is for 1-d scores)
    one = rng.randn(11)
    seen1 = (np.zeros(3, np.int64), np.array([2, 5, 7]))
    one_ref = one.copy()
    RecommenderModel.downvote_seen_items(one_ref[None, :], seen1 + (np.ones(3),))
    out.update(dv1_scores=one, dv1_items=seen1[1], dv1_lowered=one_ref)
    # k == n_items and the k > n_items error
    a = rng.randn(6)
    out.update(ts_a=a, ts_full=RecommenderModel.topsort(a, 6))
    try:
        RecommenderModel.topsort(a, 7)
        raise AssertionError('expected ValueError')
    except ValueError:
        pass
    # chunk boundaries (utils.py:16-53) for the BASELINE shapes at the default hard limit
    for tag, shp, k, mult in (('ml1m', (1208, 3706), 10, 1), ('ml20m', (138493, 26744), 20, 1),
                              ('s1m', (1000000, 100000), 10, 1), ('coffee', (1208, 3706, 5), 10, 4)):
        split = ref_utils.array_split(shp, k, mult)
        assert np.array_equal(split, orc.array_split(shp, k, mult))
        out['split_' + tag] = split
    np.savez_compressed(os.path.join(HERE, 'micro.npz'), **out)
    print('micro ok')


if __name__ == '__main__':
    micro_fixtures()
    # state 4 (warm start, default config defaults.py:5-14)
    svd_fixture('svd_warm', frame(700, 500, 40, 8, 5, seed=11, min_items=12, max_items=200),
                {}, rank=8, topk=10, seed=11, extra_ranks=(5,))
    # state 2/3-like: known users, all scored, holdout from training; threshold zeroes test feedback
    svd_fixture('svd_known', frame(500, 320, 30, 6, 5, seed=12, min_items=10, max_items=120),
                dict(test_ratio=0, warm_start=False, holdout_size=3), rank=12, topk=7, seed=12,
                feedback_threshold=3)
    # few unseen items: seen items must re-enter the list after all unseen ones
    svd_fixture('svd_fewunseen', frame(300, 48, 20, 4, 5, seed=13, min_items=8, max_items=46),
                dict(test_ratio=0, warm_start=False, holdout_size=1), rank=6, topk=10, seed=13)
    svd_fixture('svd_nofilter', frame(400, 260, 25, 5, 5, seed=14, min_items=10, max_items=100),
                {}, rank=10, topk=10, filter_seen=False, seed=14)
    svd_fixture('svd_scaled', frame(450, 300, 28, 6, 5, seed=17, min_items=10, max_items=110),
                {}, rank=9, topk=10, seed=17, scaled=(0.4, 0.8))
    coffee_fixture('coffee_small', frame(260, 180, 22, 5, 5, seed=15, min_items=8, max_items=80),
                   dict(test_ratio=0, warm_start=False, holdout_size=1), mlrank=(6, 5, 3), topk=10,
                   seed=15)
    coffee_fixture('coffee_warm', frame(300, 150, 18, 4, 5, seed=16, min_items=8, max_items=60),
                   {}, mlrank=(5, 4, 2), topk=5, seed=16)
