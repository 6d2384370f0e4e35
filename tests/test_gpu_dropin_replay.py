"""-m gpu: the drop-in walk of tests/test_dropin_polara.py REPLAYED on the HIP backend.

Polara cannot travel to the GPU box, so `tests/golden/make_dropin_walk.py` recorded, from the reference driven in
the build container, what its `RecommenderData` handed to a model after each of 11 configuration changes (the
protocol outputs of SURVEY.md §8b and the events it fired) together with the reference model's state, build count,
lists and hit counts.  Here a stand-in data object replays those outputs and events to OUR models on `HipOps`:
readiness, cache state, number of rebuilds, lists and hits must be the reference's at every step."""
import numpy as np
import pytest

from conftest import load_golden
from polara_amd import data as pk_data
from polara_amd.data import ArrayData
from polara_amd.models import SVDModel, CoffeeModel


class ReplayData(ArrayData):
    """Hands out the recorded protocol outputs of one step of the walk and fires the recorded events."""

    def __init__(self, g):
        self.g = g
        self.step = 0
        idx = g['s00_mat_idx']
        shp = tuple(int(x) for x in g['s00_mat_shape'])
        super().__init__((idx[:, 0], idx[:, 1], g['s00_mat_val']), n_users=shp[0], n_items=shp[1])
        self._load()

    def _key(self, name):
        return self.g['s%02d_%s' % (self.step, name)]

    def _load(self):
        h = self._key('holdout')
        self._test = pk_data.TestData(None, pk_data.Triplets(h[0].astype(np.int64), h[1].astype(np.int64), h[2]))
        self.warm_start = bool(self._key('warm_start'))
        self.holdout_size = int(self._key('holdout_size'))

    def goto(self, step):
        self.step = step
        self._load()
        for ev in str(self._key('events')).split(','):
            if ev == 'change':
                self._notify(self.on_change_event)
            elif ev == 'update':
                self._notify(self.on_update_event)

    def to_coo(self, tensor_mode=False, feedback_threshold=None):
        tag = 'ten' if tensor_mode else 'mat'
        idx, val = self.threshold_data(self._key(tag + '_idx'), self._key(tag + '_val'), feedback_threshold)
        return idx.astype(np.intp), np.ascontiguousarray(val), tuple(int(x) for x in self._key(tag + '_shape'))

    def test_to_coo(self, tensor_mode=False, feedback_threshold=None):
        t = self._key(('ten' if tensor_mode else 'mat') + '_test')
        users, items = t[0].astype(np.int64), t[1].astype(np.int64)
        vals = t[2].astype(np.intp) if tensor_mode else t[2]
        (users, items), vals = self.threshold_data((users, items), vals, feedback_threshold, filter_values=False)
        return users, items, vals

    def get_test_shape(self, tensor_mode=False):
        return tuple(int(x) for x in self._key(('ten' if tensor_mode else 'mat') + '_test_shape'))


def _replay(kind, ops):
    g = load_golden('dropin_walk_' + kind)
    data = ReplayData(g)
    m = (SVDModel if kind == 'svd' else CoffeeModel)(data, ops=ops)
    m.verbose = False
    m.topk = int(g['topk'])
    if kind == 'svd':
        m.rank = 6
    else:
        m.mlrank, m.seed, m.growth_tol = (5, 5, 3), 1, 1e-6
    n_steps = int(g['n_steps'])
    changes = ['start'] + [str(w) for w in g['walk']]
    for step in range(n_steps):
        if step:
            data.goto(step)
        state = (m._is_ready, m._recommendations is None)
        want = (bool(g['s%02d_ready' % step]), bool(g['s%02d_lists_dropped' % step]))
        assert state == want, (step, changes[step], state, want)
        recs = m.recommendations
        ref = g['s%02d_recs' % step].astype(np.int64)
        clear = g['s%02d_clear' % step]
        assert recs.shape == ref.shape, (step, changes[step])
        assert clear.mean() > 0.6 and np.array_equal(recs[clear], ref[clear]), (step, changes[step],
                                                                                 int((recs[clear] != ref[clear]).any(axis=1).sum()))
        assert len(m.training_time) == int(g['s%02d_builds' % step]), (step, changes[step])   # rebuilt exactly as often
        tp = m.evaluate('hits').true_positive
        slack = int((~clear).sum()) * int(g['topk'])
        assert abs(tp - int(g['s%02d_true_positive' % step])) <= slack, (step, changes[step], tp)
    return n_steps


def test_replay_data_on_the_cpu_double():
    """the replay harness itself, on the NumPy double of the device ops (runs anywhere)"""
    from numpy_ops import NumpyOps
    assert _replay('svd', NumpyOps()) == 12


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['svd', 'coffee'])
def test_dropin_walk_replayed_on_hip(hip_ops, kind):
    assert _replay(kind, hip_ops) == 12
