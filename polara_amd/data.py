"""ArrayData: the slice of the `RecommenderData` protocol that the factorization-and-scoring hot
path touches, backed by plain NumPy triplets.

It lets the device models run and be benchmarked where Polara itself (pandas state machine,
polara/recommender/data.py) is not installed — e.g. the GPU box.  When Polara is present, the same
model classes accept its `RecommenderData` unchanged, because they only use:
  fields, subscribe/on_change_event/on_update_event, to_coo, test_to_coo, get_test_shape,
  warm_start, test_sample, holdout_size, test.testset/test.holdout.
Semantics restated from data.py:777-884 (threshold_data, to_coo, test_to_coo, get_test_shape).
"""
from collections import namedtuple
from weakref import WeakKeyDictionary
import numpy as np

Fields = namedtuple('Fields', 'userid itemid feedback')
TestData = namedtuple('TestData', 'testset holdout')
Triplets = namedtuple('Triplets', 'userid itemid feedback')


class _Notifier:
    """Weak-reference observer used by RecommenderData (data.py:35-76): callbacks must be bound
    methods; they are invoked as func(subscriber)."""

    def __init__(self, events):
        self._subs = {e: WeakKeyDictionary() for e in events}

    def subscribe(self, event, callback):
        self._subs[event].setdefault(callback.__self__, set()).add(callback.__func__)

    def __call__(self, event):
        for ref in list(self._subs[event].keyrefs()):
            sub = ref()
            if sub is not None:
                for func in list(self._subs[event].get(sub, ())):
                    func(sub)


class ArrayData:
    """training: (users, items, feedback) arrays with contiguous 0-based ids.
    test: triplets of the test users' known interactions, sorted by user (ids are re-based by the
    model exactly like models.py:244-255), or None when test users are training users
    (warm_start=False: the testset is recovered from the training rows of the holdout users,
    data.py:820-832).  holdout: triplets hidden from the model (consumed by evaluation only)."""

    @staticmethod
    def _frozen(u, i, f):
        """The stored columns: arrays that already have the working types are ALIASED, not copied (480 MB at 2e7 entries).
        The contract: the caller does not write to them afterwards — cached level sets, test-user counts and
        device images would go stale with no change event.  The stored views are marked read-only, so a write through
        `data.training` / `data.test` raises; a write through the caller's own reference cannot be caught — call
        `set_training_data` / `set_test_data` with the new arrays instead."""
        out = []
        for a, dt in ((u, np.int64), (i, np.int64), (f, np.float64)):
            a = np.asarray(a)
            v = a.astype(dt, copy=False)
            if v is a or v.base is not None:
                v = v.view()
            v.setflags(write=False)
            out.append(v)
        return Triplets(*out)

    def __init__(self, training, n_users=None, n_items=None, test=None, holdout=None, warm_start=False,
                 fields=('userid', 'itemid', 'rating'), holdout_size=None):
        u, i, f = (np.asarray(a) for a in training)
        self._train = self._frozen(u, i, f)
        self.n_users = int(n_users if n_users is not None else u.max() + 1)
        self.n_items = int(n_items if n_items is not None else i.max() + 1)
        self.fields = Fields(*fields)
        self.warm_start = bool(warm_start)
        self.test_sample = None
        # the reference takes holdout_size from the data configuration (data.py: `holdout_size`), where it always
        # equals the number of items actually held out per user; here it is inferred from the holdout itself
        # (largest number of holdout items of a user) unless given — and a given value must agree with the holdout,
        # because evaluate() switches to the HR / reciprocal-rank family when it is 1 (models.py:453-462)
        self._holdout_size_given = holdout_size
        self.holdout_size = holdout_size if holdout_size is not None else 0
        self.on_change_event = 'on_change'
        self.on_update_event = 'on_update'
        self._notify = _Notifier([self.on_change_event, self.on_update_event])
        self._feedback_levels = None
        self.set_test_data(test, holdout, notify=False)

    # ---- observer protocol (data.py:160-164) ----------------------------------------------------
    def subscribe(self, event, model_callback):
        self._notify.subscribe(event, model_callback)

    def update(self):
        pass

    @property
    def training(self):
        return self._train

    @property
    def test(self):
        return self._test

    def set_training_data(self, training):
        u, i, f = (np.asarray(a) for a in training)
        self._train = self._frozen(u, i, f)
        self._feedback_levels = None
        self._notify(self.on_change_event)

    def set_test_data(self, testset=None, holdout=None, notify=True):
        def norm(t):
            if t is None:
                return None
            u, i, f = (np.asarray(a) for a in t)
            if len(u) < 2 or bool((u[1:] >= u[:-1]).all()):
                # already sorted by user (the usual case): a stable sort would be the identity — no order array, no gathers
                return self._frozen(u, i, f)
            order = np.argsort(u, kind='stable')  # data.py `_try_sort_test_data`
            return Triplets(u[order].astype(np.int64), i[order].astype(np.int64),
                            np.asarray(f, dtype=np.float64)[order])
        self._test = TestData(norm(testset), norm(holdout))
        self._set_holdout_size()
        self._holdout_size_given = None      # the constructor's value described the constructor's holdout only
        if notify:
            self._notify(self.on_update_event)

    def _set_holdout_size(self):
        hold = self._test.holdout
        per_user = 0
        if hold is not None and len(hold.userid):
            u = hold.userid
            first = np.flatnonzero(np.r_[True, u[1:] != u[:-1]])          # sorted by user (set_test_data)
            per_user = int(np.diff(np.r_[first, len(u)]).max())
        given = getattr(self, '_holdout_size_given', None)
        if given is not None and hold is not None and int(given) != per_user:
            raise ValueError('holdout_size=%d, but the holdout has up to %d items per user' % (given, per_user))
        self.holdout_size = int(given) if given is not None else per_user

    # ---- the hot-path protocol ----------------------------------------------------------------------
    @staticmethod
    def threshold_data(idx, val, threshold, filter_values=True):
        """data.py:777-791."""
        if threshold is None:
            return idx, val
        keep = val >= threshold
        if filter_values:
            val = val[keep]
            idx = tuple(x[keep] for x in idx) if isinstance(idx, tuple) else idx[keep, :]
        else:
            val = val.copy()
            val[~keep] = 0
        return idx, val

    def _levels(self):
        if self._feedback_levels is None:
            self._feedback_levels = np.unique(self._train.feedback)  # sorted, like reindex(sort=True)
        return self._feedback_levels

    def to_coo(self, tensor_mode=False, feedback_threshold=None):
        """data.py:794-817."""
        u, i, f = self._train
        if tensor_mode:
            new_f = np.searchsorted(self._levels(), f)
            idx = np.stack([u, i, new_f], axis=1)
            val = np.ones(len(f))
            shp = (self.n_users, self.n_items, len(self._levels()))
        else:
            idx = np.stack([u, i], axis=1)
            val = f
            shp = (self.n_users, self.n_items)
        idx, val = self.threshold_data(idx, val, feedback_threshold)
        return idx.astype(np.intp, copy=False), np.ascontiguousarray(val), tuple(int(s) for s in shp)

    def matrix_triplets(self, feedback_threshold=None):
        """(users, items, feedback, shape) of `to_coo(tensor_mode=False)` WITHOUT the [nnz x 2] index array: the three
        columns as they lie in memory.  A shortcut for device models (the stacked copy of a 2e7-entry index is 35 ms of
        host time next to a 55 ms solver); anything written against the reference's protocol calls `to_coo`."""
        u, i, f = self._train
        (u, i), f = self.threshold_data((u, i), f, feedback_threshold)
        return u, i, np.ascontiguousarray(f), (int(self.n_users), int(self.n_items))

    def tensor_triplets(self):
        """(users, items, feedback, levels, shape) of `to_coo(tensor_mode=True)` WITHOUT the stacked [nnz x 3] index: the
        columns as they lie and the sorted feedback levels, for a device model that looks the level of every entry up (and
        relabels the items) on the device — the stacked index and its host-side passes are a quarter of a 50 ms HOOI
        build.  Anything written against the reference's protocol calls `to_coo`."""
        u, i, f = self._train
        levels = self._levels()
        return u, i, f, levels, (int(self.n_users), int(self.n_items), len(levels))

    def _recover_testset(self):
        """data.py:820-832: training rows of the holdout users, sorted by user."""
        users = np.unique(self._test.holdout.userid)
        u, i, f = self._train
        if len(users) == self.n_users:
            sel = slice(None)
        else:
            sel = np.isin(u, users)
        u, i, f = u[sel], i[sel], f[sel]
        order = np.argsort(u, kind='stable')
        return Triplets(u[order], i[order], f[order])

    def test_to_coo(self, tensor_mode=False, feedback_threshold=None):
        """data.py:835-862."""
        testset = self._test.testset
        if testset is None:
            if self.warm_start or self._test.holdout is None:
                raise ValueError('Unable to read test data')
            testset = self._recover_testset()
        u, i, f = testset
        if tensor_mode:
            levels = self._levels()
            pos = np.searchsorted(levels, f)
            if (pos >= len(levels)).any() or (levels[np.minimum(pos, len(levels) - 1)] != f).any():
                raise NotImplementedError('Not all values of feedback are present in training data')
            coo, val = (u, i), pos.astype(np.intp)
        else:
            coo, val = (u, i), f
        coo, val = self.threshold_data(coo, val, feedback_threshold, filter_values=False)
        return coo + (val,)

    def get_test_shape(self, tensor_mode=False):
        """data.py:865-884."""
        src = self._test.holdout if self._test.holdout is not None else self._test.testset
        if getattr(self, '_n_test_users', None) is None or self._n_test_users[0] is not src:
            u = src.userid                                   # sorted by user (set_test_data): distinct = boundaries + 1
            self._n_test_users = (src, (int(np.count_nonzero(u[1:] != u[:-1])) + 1) if len(u) else 0)
        num_users = self._n_test_users[1]
        shape = (num_users, self.n_items)
        if tensor_mode:
            shape = shape + (len(self._levels()),)
        return shape


class ShardedArrayData(ArrayData):
    """One rank's row block of a user-sharded dataset (polara_amd/shards.py).

    User ids are LOCAL (0 .. hi-lo); `user_range = (lo, hi)` places the block among `n_users_total` users.
    A model built on it with the job's communicator factorizes the WHOLE matrix (item factors, singular values
    and — for CoFFee — feedback factors and core are global and identical on every rank) while everything
    per-user stays with the rank that owns the user: the user-factor rows, `get_recommendations()` and
    `evaluate()` cover the local users only (nothing of size n_users x topk is gathered; sum the hit counts
    over ranks for job-wide metrics).  score_all=True makes every local user a test user with its training
    row as the known preferences; with a `holdout` (local user ids) the test users are the holdout's users and
    their known preferences are recovered from the training rows (the reference's `test_ratio=0,
    warm_start=False` state, data.py:820-832)."""

    def __init__(self, block, n_users_total, feedback_levels=None, score_all=True, holdout=None,
                 fields=('userid', 'itemid', 'rating')):
        indptr = np.asarray(block.indptr, dtype=np.int64)
        u = np.repeat(np.arange(block.n_rows, dtype=np.int64), np.diff(indptr))
        i = np.asarray(block.indices, dtype=np.int64)
        f = np.ones(len(i)) if block.values is None else np.asarray(block.values, dtype=np.float64)
        self.user_range = (int(block.row_lo), int(block.row_hi))
        self.n_users_total = int(n_users_total)
        self.local_csr = block
        self._fixed_levels = None if feedback_levels is None else np.asarray(feedback_levels, dtype=np.float64)
        # the users to score are exactly the training rows: a model may score its device-resident training matrix
        self.scores_training_rows = bool(score_all and holdout is None)
        super().__init__((u, i, f), n_users=block.n_rows, n_items=block.n_cols,
                         test=(u, i, f) if (score_all and holdout is None) else None, holdout=holdout, fields=fields)

    @classmethod
    def from_shards(cls, path, rank=0, world=1, **kwargs):
        """The block of rank `rank` of `world` from a dataset directory written by shards.write_csr_shards /
        ShardWriter."""
        from .shards import load_rank_block
        block, manifest = load_rank_block(path, rank, world)
        kwargs.setdefault('feedback_levels', manifest.get('feedback_levels'))
        return cls(block, manifest['n_rows'], **kwargs)

    def set_test_data(self, testset=None, holdout=None, notify=True):
        if hasattr(self, 'scores_training_rows') and hasattr(self, '_test'):
            self.scores_training_rows = False          # an explicit test set from now on
        super().set_test_data(testset, holdout, notify)

    def set_training_data(self, training):
        raise NotImplementedError('a shard is immutable: write a new dataset (polara_amd.shards) instead')

    def _levels(self):
        # the SAME level set on every rank (a block need not contain every level)
        if self._fixed_levels is not None:
            return self._fixed_levels
        return super()._levels()
