"""Records the drop-in walk of tests/test_dropin_polara.py::test_data_events_reach_both_models_alike from the
REFERENCE ITSELF, so that it can be replayed on the GPU box (where Polara cannot travel): tests/golden/dropin_walk_{svd,
coffee}.npz.

Polara's own `RecommenderData` is walked through a sequence of configuration changes (data.py:166-330); the
reference's `SVDModel` / `CoffeeModel` is subscribed to it.  After every `data.update()` the script stores
  * what the data object now hands to a model through the protocol the hot path uses: `to_coo`
    (matrix and tensor mode), `test_to_coo` (both modes), `get_test_shape`, `warm_start`, `holdout_size`, the holdout;
  * which events it fired (`on_change`, `on_update`: probed by a subscriber);
  * the reference model's state — readiness, whether its cached lists were dropped, how many builds so far — and its
    next recommendations, with the rows whose top-(k+1) scores are pairwise distinct (the others are
    implementation-defined in the reference) and its `evaluate()` hit counts.
The replay (tests/test_gpu_dropin_replay.py) feeds the same protocol outputs and events to OUR model on the HIP
backend and requires the same states, build counts, lists and hit counts at every step.

usage:  python tests/golden/make_dropin_walk.py        (build container)
"""
import contextlib
import io
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, '_numba_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

import numpy as np
import pandas as pd

import polara
from polara.recommender.models import CoffeeModel as RefCoffee

from polara_amd.synth import planted_csr, csr_to_coo_triplets

WALK = [('holdout_size', 2), ('random_holdout', True), ('test_sample', 10), ('test_sample', None), ('test_fold', 3),
        ('holdout_size', 1), ('warm_start', False), ('test_ratio', 0.25), ('test_fold', 2), ('holdout_size', 3),
        ('warm_start', True)]
TOPK = 8


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


class EventProbe:
    def __init__(self, data):
        self.seen = []
        self.data = data
        data.subscribe(data.on_change_event, self.changed)
        data.subscribe(data.on_update_event, self.updated)

    def changed(self):
        self.seen.append('change')

    def updated(self):
        self.seen.append('update')

    def drain(self):
        out, self.seen = self.seen, []
        return out


def clear_rows(model, topk):
    test_data, shape, _ = model._get_test_data()
    scores, sd = model.slice_recommendations(test_data, shape, 0, shape[0])
    if model.filter_seen:
        model.downvote_seen_items(scores, sd)
    top = -np.sort(-scores, axis=1)[:, :topk + 1]
    return (np.diff(-top, axis=1) > 1e-9 * np.abs(top[:, :1])).all(axis=1)


def snapshot(data, model, probe, out, step):
    p = 's%02d_' % step
    events = probe.drain()
    out[p + 'events'] = np.array(','.join(events))
    out[p + 'ready'] = np.bool_(model._is_ready)
    out[p + 'lists_dropped'] = np.bool_(model._recommendations is None)
    for tensor_mode, tag in ((False, 'mat'), (True, 'ten')):
        idx, val, shp = data.to_coo(tensor_mode=tensor_mode)
        out[p + tag + '_idx'], out[p + tag + '_val'], out[p + tag + '_shape'] = idx.astype(np.int64), val, np.array(shp, np.int64)
        tu, ti, tf = data.test_to_coo(tensor_mode=tensor_mode)
        out[p + tag + '_test'] = np.stack([np.asarray(tu, np.float64), np.asarray(ti, np.float64), np.asarray(tf, np.float64)])
        out[p + tag + '_test_shape'] = np.array(data.get_test_shape(tensor_mode=tensor_mode), np.int64)
    f = data.fields
    h = data.test.holdout
    out[p + 'holdout'] = np.stack([h[f.userid].values.astype(np.float64), h[f.itemid].values.astype(np.float64),
                                   h[f.feedback].values.astype(np.float64)])
    out[p + 'warm_start'] = np.bool_(data.warm_start)
    out[p + 'holdout_size'] = np.int64(data.holdout_size)
    np.random.seed(0)
    recs = quiet(lambda: model.recommendations)
    out[p + 'recs'] = recs.astype(np.int32)
    out[p + 'clear'] = clear_rows(model, TOPK)
    out[p + 'builds'] = np.int64(len(model.training_time))
    hits = quiet(model.evaluate, 'hits')
    out[p + 'true_positive'] = np.int64(hits.true_positive)


def record(kind):
    u, i, v = csr_to_coo_triplets(planted_csr(400, 150, 18, 5, levels=5, seed=11, min_items=6, max_items=60))
    df = pd.DataFrame({'userid': 1000 + 3 * u, 'itemid': 50000 - 7 * i, 'rating': v})
    data = polara.RecommenderData(df, 'userid', 'itemid', 'rating', seed=0)
    data.verbose = False
    data.warm_start, data.holdout_size, data.test_ratio = True, 3, 0.2
    quiet(data.prepare)
    model = polara.SVDModel(data) if kind == 'svd' else RefCoffee(data)
    model.verbose = False
    model.topk = TOPK
    if kind == 'svd':
        model.rank = 6
    else:
        model.mlrank, model.seed, model.growth_tol = (5, 5, 3), 1, 1e-6
    probe = EventProbe(data)
    out = dict(kind=np.array(kind), topk=np.int64(TOPK), n_steps=np.int64(len(WALK) + 1),
               walk=np.array(['%s=%r' % w for w in WALK]))
    snapshot(data, model, probe, out, 0)
    for step, (attr, value) in enumerate(WALK, 1):
        setattr(data, attr, value)
        quiet(data.update)
        snapshot(data, model, probe, out, step)
    np.savez_compressed(os.path.join(HERE, 'dropin_walk_%s.npz' % kind), **out)
    print(kind, 'steps', len(WALK) + 1, 'builds', int(out['s%02d_builds' % len(WALK)]),
          'events', [str(out['s%02d_events' % s]) for s in range(len(WALK) + 1)])


if __name__ == '__main__':
    record('svd')
    record('coffee')
