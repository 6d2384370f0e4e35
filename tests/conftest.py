import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: a CPU test of a minute or more (regenerates a large fixture from the imported reference)')
    config.addinivalue_line('markers', 'probe: kernel variants of csrc/experiments/ — need a probe build selected with POLARA_HIP_LIB')


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the two-rank tests: a fixed number may be taken)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def hip_ops():
    """The product backend.  Fails (does not skip) when the library or the GPU is missing:
    a -m gpu run that silently fell back to anything else would prove nothing."""
    from polara_amd.ops import HipOps
    return HipOps()


class GoldenData:
    """Feeds a model the exact COO triplets the reference model saw when the fixture was made."""

    def __new__(cls, g):
        from polara_amd.data import ArrayData

        class _GD(ArrayData):
            def __init__(self, g):
                self.g = g
                idx = g['train_idx']
                shp = tuple(int(x) for x in g['train_shape'])
                super().__init__((idx[:, 0], idx[:, 1], g['train_val']), n_users=shp[0], n_items=shp[1])

            def to_coo(self, tensor_mode=False, feedback_threshold=None):
                g = self.g
                return g['train_idx'].astype(np.intp), g['train_val'], tuple(int(x) for x in g['train_shape'])

            def test_to_coo(self, tensor_mode=False, feedback_threshold=None):
                g = self.g
                return (g['test_user'], g['test_item'], g['test_fdbk'])

            def get_test_shape(self, tensor_mode=False):
                s = tuple(int(x) for x in self.g['test_shape'])
                return s if tensor_mode else s[:2]
        return _GD(g)


def pytest_collection_modifyitems(config, items):
    # the reference's own code raises pandas FutureWarnings by the dozen under pandas 2: not ours to fix, not worth a page of output
    for item in items:
        if 'test_dropin_polara' in item.nodeid:
            item.add_marker(pytest.mark.filterwarnings('ignore::FutureWarning'))
            item.add_marker(pytest.mark.filterwarnings('ignore::DeprecationWarning'))


def check_coffee_extras(m, g):
    """The extras of the reference's CoffeeModel (models.py:1027-1092) on a built model fed a golden fixture: unfolded
    test-tensor slices and the holdout slice equal to the reference's, `predict_feedback` equal to the reference's
    predictions wherever the best and the second-best level are apart (our factors differ from the reference's by the
    signs the reference leaves arbitrary; the reconstructed scores do not)."""
    import scipy.sparse as sps
    a, b = (int(x) for x in g['unfold_range'])
    td = (g['test_user'], g['test_item'], g['test_fdbk'])
    shape = tuple(int(x) for x in g['test_shape'])
    for mode in (0, 1, 2):
        unf, sl = m.unfold_test_tensor_slice(td, shape, a, b, mode)
        ref = sps.csr_matrix((g['unfold%d_data' % mode], g['unfold%d_indices' % mode], g['unfold%d_indptr' % mode]),
                             shape=tuple(int(x) for x in g['unfold%d_shape' % mode]))
        assert unf.dtype == np.uint8 and unf.shape == ref.shape and (unf.astype(np.int64) != ref).nnz == 0, mode
        assert all(np.array_equal(x, y) for x, y in zip(sl, m._slice_test_data(td, a, b)))
    hold = (g['hold_user'], g['hold_item'], np.ones(len(g['hold_user'])))
    m.data.set_test_data(holdout=hold, notify=False)
    hu, hi = m.get_holdout_slice(a, b)
    assert np.array_equal(hu, g['hold_slice_user']) and np.array_equal(hi, g['hold_slice_item'])
    if 'predicted_feedback' in g:
        m.data._feedback_levels = g['feedback_levels']          # the original feedback values of the levels
        pred = m.predict_feedback()
        clear = g['predict_gap'] > 1e-9
        assert clear.mean() > 0.99 and pred.shape == g['predicted_feedback'].shape
        assert np.array_equal(pred[clear], g['predicted_feedback'][clear])
    else:
        m.data.warm_start = True
        with pytest.raises(NotImplementedError):
            m.predict_feedback()


@pytest.fixture
def pk_options():
    """Process-wide options of the library for the duration of one test (pk_set_option: explicit calls, the library reads no
    environment variable): `pk_options('score_head_tiles', 3)`; every option touched returns to its default afterwards."""
    from polara_amd import _lib
    lib = _lib.load()
    touched = []

    def set_option(name, value):
        _lib.check(lib.pk_set_option(name.encode(), int(value), 0), 'pk_set_option')
        touched.append(name)
    yield set_option
    for name in touched:
        lib.pk_set_option(name.encode(), 0, 1)
