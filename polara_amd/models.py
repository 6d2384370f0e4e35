"""Device-backed models behind Polara's `RecommenderModel` plugin surface.

`SVDModel` (PureSVD) and `CoffeeModel` (CoFFee / Tucker) keep the names, constructor, properties,
`factors` dict layout, cache-invalidation rules and return types of the reference classes
(polara/recommender/models.py:70-604, 800-861, 901-1054) so they drop into `RecommenderData`
/ `evaluate()` / `evaluation_engine` unchanged, while `build()` and `get_recommendations()` run on
the GPU through libpolarahip.so.  There is no CPU path here: constructing the device backend
without the library or a GPU raises.
"""
from timeit import default_timer as timer

import numpy as np
import torch

from . import defaults
from .operator import DeviceChain, HostOperator, SparseProduct
from .csr import coo_to_csr, nnz_balanced_row_partition, popularity_order
from .solver import svd_topk, NoComm, NoConvergence
from . import scoring
from . import tucker


def get_default(name):
    return defaults.get_config([name])[name]


def _format_elapsed(seconds_total):
    """tools/timing.py:10-17."""
    minutes, seconds = divmod(seconds_total, 60)
    hours, minutes = divmod(minutes, 60)
    if hours == 0:
        if minutes == 0:
            return f'{seconds:.3f}s'
        return f'{minutes:>02.0f}m:{seconds:>02.0f}s'
    return f'{hours:.0f}h:{minutes:>02.0f}m:{seconds:>02.0f}s'


def _setting(name, on_change, doc):
    """A configuration attribute stored as `_<name>` whose CHANGE (assigning an equal value is a no-op) calls the
    model method `on_change` — the reference's cache rules (models.py:119-148, 870-887, 930-940) in one place."""
    slot = '_' + name

    def get(self):
        return getattr(self, slot)

    def put(self, value):
        if getattr(self, slot) != value:
            setattr(self, slot, value)
            getattr(self, on_change)()
    return property(get, put, doc=doc)


class RecommenderModel:
    """Base class: configuration, caching and the recommend pipeline (models.py:70-604)."""
    _config = ('topk', 'filter_seen', 'switch_positive', 'feedback_threshold', 'verify_integrity')

    def __init_subclass__(cls, **kw):
        # the reference wraps every `build` with cache invalidation through a metaclass
        # (models.py:34-67); same effect here
        super().__init_subclass__(**kw)
        if 'build' in cls.__dict__:
            raw = cls.__dict__['build']

            def build(self, *args, **kwargs):
                self._is_ready = False
                self._recommendations = None
                res = raw(self, *args, **kwargs)
                self._is_ready = True
                return res
            build.__name__ = 'build'
            build.__doc__ = raw.__doc__
            cls.build = build

    def __init__(self, recommender_data, feedback_threshold=None, ops=None, comm=None):
        self.data = recommender_data
        self._recommendations = None
        self._recs_dev = None
        self.method = 'ABC'
        self._topk = get_default('topk')
        self._filter_seen = get_default('filter_seen')
        self._feedback_threshold = feedback_threshold or get_default('feedback_threshold')
        self.switch_positive = get_default('switch_positive')
        self.verify_integrity = get_default('verify_integrity')
        self.max_test_workers = get_default('max_test_workers')  # accepted, unused: no host chunk loop (passes issued by several
        # host threads on one ops object are enqueued one at a time: scoring.recommend holds ops.pass_lock)
        self._prediction_key = self.data.fields.userid
        self._prediction_target = self.data.fields.itemid
        self._is_ready = False
        self.verbose = True
        self.training_time = []
        self.recommend_stats = {}
        self.collect_recommend_stats = False   # sweep statistics cost host round trips: tuning / tests only
        self._ops = ops
        self.comm = comm or NoComm()
        self._factor_image = None
        self._factor_src = None       # the host array in `factors` the cached image was made from
        self._train_dev = None        # (device CSR of the training rows, its internal->external item map) when they double as the test rows
        self._test_dev = None         # (item order it was built for, device CSR of the test users, n_users, n_items): dropped by data events
        # internal item order of the device path (csr.popularity_order): external id -> internal
        # position and back; None = identity.  `factors` and every result stay in EXTERNAL ids.
        self._item_rank = None
        self._item_inv = None
        self.data.subscribe(self.data.on_change_event, self._renew_model)
        self.data.subscribe(self.data.on_update_event, self._refresh_model)

    # ---- device backend (fails loudly; no CPU fallback) ---------------------------------------------
    @property
    def ops(self):
        if self._ops is None:
            from .ops import HipOps
            self._ops = HipOps()
        return self._ops

    # ---- caching protocol (models.py:99-148) -----------------------------------------------------------
    @property
    def recommendations(self):
        if self._recommendations is None:
            if not self._is_ready:
                if self.verbose:
                    print('{} model is not ready. Rebuilding.'.format(self.method))
                self.build()
            self._recommendations = self.get_recommendations()
        return self._recommendations

    def _renew_model(self):
        self._recommendations = None
        self._is_ready = False
        self._factor_image = None
        self._train_dev = None
        self._resident = None
        self._test_dev = None

    def _refresh_model(self):
        self._recommendations = None
        self._test_dev = None

    @property
    def topk(self):
        return self._topk

    @topk.setter
    def topk(self, new_value):
        # cached lists are cut when k shrinks (their leading columns) and dropped only when it grows (models.py:123-128)
        cached = self._recommendations
        if cached is not None and new_value > cached.shape[1]:
            self._recommendations = None
        self._topk = new_value

    feedback_threshold = _setting('feedback_threshold', '_renew_model', 'training entries below it are dropped: a new model')
    filter_seen = _setting('filter_seen', '_refresh_model', 'whether seen items may be recommended: new lists, same model')

    def get_base_configuration(self):
        return {attr: getattr(self, attr) for attr in self._config}

    def build(self):
        raise NotImplementedError('This must be implemented in subclasses')

    def _track(self, start):
        """tools/timing.py:20-34 contract: one entry per build, same message."""
        elapsed = timer() - start
        if self.training_time is not None:
            self.training_time.append(elapsed)
        if self.verbose:
            print('{} training time: {}'.format(self.method, _format_elapsed(elapsed)))

    # ---- data access (models.py:160-257) -----------------------------------------------------------------
    def _training_csr(self, dtype=np.float64, ignore_feedback=False):
        threshold = self.feedback_threshold
        idx, val, shp = self.data.to_coo(tensor_mode=False, feedback_threshold=threshold)
        val = np.ones_like(val, dtype=dtype) if ignore_feedback else np.asarray(val, dtype=dtype)
        indptr, indices, values = coo_to_csr(idx[:, 0], idx[:, 1], val, shp)
        return indptr, indices, values, shp

    def get_training_matrix(self, feedback_threshold=None, ignore_feedback=False, sparse_format='csr', dtype=None):
        """models.py:160-177 (returns a SciPy matrix for API compatibility)."""
        from scipy.sparse import csr_matrix
        threshold = feedback_threshold or self.feedback_threshold
        idx, val, shp = self.data.to_coo(tensor_mode=False, feedback_threshold=threshold)
        dtype = dtype or val.dtype
        val = np.ones_like(val, dtype=dtype) if ignore_feedback else val.astype(dtype)
        indptr, indices, values = coo_to_csr(idx[:, 0], idx[:, 1], val, shp)
        m = csr_matrix((values, indices, indptr), shape=shp)
        return m if sparse_format == 'csr' else m.asformat(sparse_format)

    def _tensor_mode(self):
        try:
            return self.factors.get(self.data.fields.feedback, None) is not None
        except AttributeError:
            return False

    def _get_test_data(self, feedback_threshold=None):
        """The contract of models.py:227-257: ((user, item, feedback) sorted by user with users numbered 0..n-1
        without gaps, test shape, the data-level id of the user behind every row)."""
        tensor_mode = self._tensor_mode()
        shape = self.data.get_test_shape(tensor_mode=tensor_mode)
        threshold = feedback_threshold or self.feedback_threshold
        if self.data.warm_start and threshold:
            print('Specifying threshold has no effect in warm start.')
        elif (not self.data.warm_start) and self.data.test_sample and threshold is not None:
            print('Specifying both threshold value and test_sample may change test data.')
        if self.data.warm_start:
            threshold = None
        users, items, feedback = self.data.test_to_coo(tensor_mode=tensor_mode, feedback_threshold=threshold)
        users = np.asarray(users)
        step = users[1:] - users[:-1]
        if (step < 0).any():
            raise AssertionError('the test set must be sorted by users')
        if len(users) and users[0] == 0 and not (step > 1).any():
            return (users, items, feedback), shape, np.arange(shape[0])      # already numbered without gaps
        first = np.concatenate(([True], step > 0)) if len(users) else np.zeros(0, dtype=bool)
        dense_ids = (np.cumsum(first) - 1).astype(users.dtype)               # row number of every entry
        return (dense_ids, items, feedback), shape, users[first]

    @staticmethod
    def _slice_test_data(test_data, start, stop):
        """models.py:260-270: the entries of rows [start, stop), rows renumbered from 0."""
        users, items, feedback = test_data
        inside = np.flatnonzero((users >= start) & (users < stop))
        return users[inside] - start, items[inside], feedback[inside]

    def get_test_matrix(self, test_data=None, shape=None, user_slice=None, dtype=None, ignore_feedback=False):
        """models.py:180-211 (host-side, API compatibility): the SciPy CSR of the test users — entries with zero
        feedback left out of the matrix but kept in the returned triplet, where they still count as seen — and
        that triplet.  The device path builds its own test CSR (ops.csr_from_coo) and never chunks."""
        from scipy.sparse import csr_matrix
        if test_data is None:
            test_data, shape, _ = self._get_test_data()
        elif shape is None:
            raise ValueError('Shape of test data must be provided')
        n_rows, triplet = shape[0], test_data
        if user_slice:
            lo, hi = user_slice[0], min(user_slice[1], shape[0])
            n_rows, triplet = hi - lo, self._slice_test_data(test_data, lo, hi)
        users, items, feedback = triplet
        keep = np.flatnonzero(feedback)                                      # explicit zeros do not enter the matrix
        dtype = dtype or feedback.dtype
        vals = np.ones(len(keep), dtype=dtype) if ignore_feedback else feedback[keep]
        return csr_matrix((vals, (users[keep], items[keep])), shape=(n_rows, shape[1]), dtype=dtype), triplet

    def verify_data_integrity(self):
        """models.py:581-604 reduced to the checks that do not need pandas."""
        itemid, feedback = self.data.fields.itemid, self.data.fields.feedback
        n_items = self.data.get_test_shape(tensor_mode=False)[1]
        f = getattr(self, 'factors', {})
        if f.get(itemid, None) is not None:
            assert f[itemid].shape[0] == n_items
        if feedback is not None and f.get(feedback, None) is not None:
            assert f[feedback].shape[0] == self.data.get_test_shape(tensor_mode=True)[2]

    # ---- user-sharded datasets (data.ShardedArrayData) --------------------------------------------------
    def _presharded(self):
        """True when the data object is one rank's row block of a larger dataset: the model then never
        partitions or gathers anything per-user itself."""
        rng = getattr(self.data, 'user_range', None)
        if rng is None:
            return False
        if self.comm.world == 1 and tuple(rng) != (0, self.data.n_users_total):
            raise ValueError('users %d..%d of %d: a row block of a sharded dataset needs the communicator of its job'
                             % (rng[0], rng[1], self.data.n_users_total))
        return True

    def _item_counts(self, cols, n_items):
        """Interactions per item over the WHOLE dataset (summed over ranks for a sharded one): every rank must
        derive the same internal item order."""
        counts = np.bincount(np.asarray(cols, dtype=np.int64), minlength=n_items).astype(np.int64)
        if self._presharded() and self.comm.world > 1:
            counts = self.ops.to_host(self.comm.allreduce(self.ops.to_device(counts)))
        return counts

    # ---- recommend pipeline (models.py:359-405) ----------------------------------------------------------
    def _item_factors_device(self):
        """FactorImage of the item factors in INTERNAL item order.  The cached image belongs to ONE host array: it is
        rebuilt whenever `factors` holds another one — after a rank truncation, or when a consumer swaps the
        `factors` dict itself (the reference's rank-sweep pipelines restore it behind the model's back,
        evaluation/pipelines.py:106-108)."""
        src = self.factors[self.data.fields.itemid]
        if self._factor_image is None or self._factor_src is not src:
            v = np.ascontiguousarray(src)
            if self._item_inv is not None:
                v = np.ascontiguousarray(v[self._item_inv])
            self._factor_image = scoring.FactorImage(self.ops, self.ops.to_device(v))
            self._factor_src = src
        return self._factor_image

    def _test_weights(self, test_data):
        """Per-entry fold-in coefficients; None = the feedback values themselves."""
        return None

    def _resident_test_csr(self):
        """(test CSR in the current internal item order, mask of users with interactions) when the users to score
        are the training rows already on the device (see _training_device_csr), else None.  A renaming of the
        resident matrix's columns (one device gather) replaces frame -> triplets -> upload -> sort."""
        if self._train_dev is None or not getattr(self.data, 'scores_training_rows', False) or self.feedback_threshold:
            return None
        A, inv_at_build = self._train_dev
        cached = getattr(self, '_resident', None)
        if cached is not None and cached[0] is A and cached[1] is self._item_rank:
            return cached[2], cached[3]               # same matrix, same serving order: keep its seen-tile streams
        # build-time internal id j = external item inv_at_build[j]; its current internal id is item_rank[that]
        T = self.ops.csr_relabel_cols(A, self._item_rank[inv_at_build], sort=False)
        nonempty = self.ops.to_host(A.indptr[1:] > A.indptr[:-1])
        self._resident = (A, self._item_rank, T, nonempty)
        return T, nonempty

    def _device_test_csr(self):
        """The test users' known interactions as a device CSR in the current internal item order (zeros kept: still
        "seen").  Built from the protocol's triplets (`_get_test_data`, models.py:227-257) by the ingest kernels —
        external ids go up as they are, the renaming into the internal order runs on the device — and kept until a
        data event (`on_update` / `on_change`), a threshold / flattener change or a new item order drops it: the
        reference rebuilds its test matrix per chunk and per call (models.py:180-211)."""
        key = self._test_csr_depends_on()
        cached = self._test_dev
        if cached is not None and len(cached[0]) == len(key) and all(a is b for a, b in zip(cached[0], key)):
            return cached[1:]
        ops = self.ops
        fast = self._training_rows_test_csr()
        if fast is not None:
            self._test_dev = (key,) + fast
            return fast
        test_data, test_shape, _ = self._get_test_data()
        n_users, n_items = int(test_shape[0]), int(test_shape[1])
        w = self._test_weights(test_data)
        vals = np.asarray(test_data[2] if w is None else w, dtype=np.float64)
        T = ops.csr_from_coo(test_data[0], test_data[1], vals, (n_users, n_items))
        if self._item_rank is not None:
            T = ops.csr_relabel_cols(T, self._item_rank)
        self._test_dev = (key, T, n_users, n_items)
        return T, n_users, n_items

    def _training_rows_test_csr(self):
        """Shortcuts of `_device_test_csr` on our own data object that skip the host passes of the protocol
        (`_get_test_data`, models.py:227-257: a sortedness check, a gap check and a renumbering over every test entry;
        data.py:820-832: a stable sort of the training triplets by user plus three gathers — together 45-110 ms of NumPy
        for 2e7 entries in front of a 1 ms scoring pass).  The COO -> CSR kernels sort by (user, item) themselves, so the
        columns go up as they lie and the protocol's checks run on the device:
          * no explicit test set, the holdout names EVERY user: the test rows are the training rows;
          * an explicit test set (kept sorted by user by `set_test_data`): its rows, renumbered without gaps on the device
            where the protocol would.
        Taken only when it provably gives the protocol's matrix: matrix models (no per-entry weights), no threshold, no
        test sampling; in the first case also no warm start and every user with at least one interaction (else the
        protocol renumbers rows of a RECOVERED set — left to it).  None: not applicable."""
        from .data import ArrayData
        d = self.data
        if not (isinstance(d, ArrayData) and type(d).test_to_coo is ArrayData.test_to_coo
                and type(d)._recover_testset is ArrayData._recover_testset
                and type(self)._get_test_data is RecommenderModel._get_test_data):
            return None
        test = getattr(d, '_test', None)
        if test is None or self.feedback_threshold or self._tensor_mode() or getattr(d, 'test_sample', None):
            return None
        ops = self.ops
        n_users, n_items = (int(x) for x in d.get_test_shape(tensor_mode=False))
        if test.testset is not None:
            u, i, f = test.testset
            if len(u) == 0:
                return None
            rows = scoring.renumbered_test_rows(ops, u)       # raises like the protocol if the set is not sorted by user
            T = ops.csr_from_coo(rows, ops.to_device(np.ascontiguousarray(i, dtype=np.int64)),
                                 ops.to_device(np.asarray(f, dtype=np.float64)), (n_users, n_items))
        else:
            if test.holdout is None or d.warm_start or n_users != d.n_users:
                return None                           # nothing to recover from / the holdout names only some users
            u, i, f, shp = d.matrix_triplets()
            T = ops.csr_from_coo(u, i, np.asarray(f, dtype=np.float64), (n_users, n_items))
            if not bool((T.indptr[1:] > T.indptr[:-1]).all().item()):
                return None                           # users without interactions: rows are renumbered by the protocol
        if self._item_rank is not None:
            T = ops.csr_relabel_cols(T, self._item_rank)
        return T, n_users, n_items

    def _test_csr_depends_on(self):
        """What the cached device test CSR was built from besides the data (whose changes arrive as events): objects
        compared by identity."""
        return (self._item_rank,)

    def get_recommendations(self):
        if self.verify_integrity:
            self.verify_data_integrity()
        ops, comm = self.ops, self.comm
        resident = self._resident_test_csr()
        if resident is not None:
            T, nonempty = resident
            n_users, n_items = T.shape
        else:
            nonempty = None
            T, n_users, n_items = self._device_test_csr()
        lo, hi = 0, n_users
        gather = comm.world > 1 and not self._presharded()   # a pre-sharded dataset: T already is this rank's users
        if gather:  # user-sharded scoring; V is replicated, no collective in the data path
            bounds = nnz_balanced_row_partition(ops.to_host(T.indptr), comm.world)
            lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
            if hi > lo:
                T = ops.csr_rows(T, lo, hi)
        stats = {}
        recs_dev = None
        if hi > lo:
            recs_dev = scoring.recommend(ops, self._item_factors_device(), T, self.topk, self.filter_seen,
                                         stats=stats if self.collect_recommend_stats else None)
            if hasattr(ops, 'ids_to_host'):
                # internal positions -> external item ids on the device, one transfer into pinned memory
                recs = ops.ids_to_host(recs_dev, self._item_inv)
            else:
                recs = ops.to_host(recs_dev)
                if self._item_inv is not None:   # internal positions -> external item ids
                    recs = np.where(recs >= 0, self._item_inv[np.maximum(recs, 0)], -1).astype(np.int64)
        else:
            recs = np.empty((0, self.topk), dtype=np.int64)
        self.recommend_stats = stats
        if nonempty is not None and not nonempty.all():
            # the reference's test users are the users WITH interactions (rows of the rebased test triplet)
            recs = np.ascontiguousarray(recs[nonempty])
            recs_dev = None
        if gather:
            recs = comm.gather_rows(recs, n_users, self.topk)
            recs_dev = None
        # the device-resident list (internal item ids) stays available to evaluate(), keyed by the host array it
        # belongs to: any cache invalidation replaces that array
        self._recs_dev = (recs, recs_dev) if recs_dev is not None else None
        return recs

    def slice_recommendations(self, test_data, shape, start, stop, test_users=None):
        """Dense fp64 scores of test users [start, stop) + the slice triplet (models.py:857-861,
        1042-1054).  Kept for `_user_scores`/`show_recommendations`-style consumers; computed on
        device by pk_dense_scores_f64.  `get_recommendations` does NOT go through here."""
        stop = min(stop, shape[0])
        users, items, fdbk = test_data
        sel = (users >= start) & (users < stop)
        slice_data = (users[sel] - start, items[sel], fdbk[sel])
        w = self._test_weights(test_data)
        mapped = slice_data if self._item_rank is None else (
            slice_data[0], self._item_rank[np.asarray(slice_data[1], dtype=np.intp)], slice_data[2])
        indptr, indices, values = scoring.test_csr_from_triplet(
            mapped, (stop - start, shape[1]), None if w is None else w[sel])
        T = self.ops.csr(indptr, indices, values, (stop - start, shape[1]))
        scores = self.ops.to_host(scoring.dense_scores(self.ops, self._item_factors_device(), T, 0, stop - start))
        if self._item_rank is not None:
            scores = np.ascontiguousarray(scores[:, self._item_rank])   # back to external item order
        return scores, slice_data

    # ---- single-user conveniences on top of slice_recommendations (models.py:277-356, 488-563) -------------
    # These take and return caller-owned HOST arrays, like the reference's; the arithmetic runs on the model's
    # device through torch (selection / reductions on a dense score block are plumbing here, not a kernel of the
    # hot path: `get_recommendations` never goes through them — its masking and top-k are fused into the sweep).
    def topsort(self, a, topk):
        """models.py:488-491: indices of the `topk` largest entries of a vector, best first."""
        return self.get_topk_elements(np.asarray(a)[None, :], topk)[0]

    def downvote_seen_items(self, recs, idx_seen):
        """models.py:494-519, dense branch, IN PLACE on `recs` like the reference: seen entries drop below the
        minimum of the block, keeping their relative order (s <- min - (max_seen - s) - 1)."""
        if hasattr(recs, 'tocsr'):
            raise NotImplementedError('sparse score matrices are not produced by the factorization models')
        if recs.ndim == 1:                                    # single-user scores (models.py:513-515)
            flat = np.asarray(idx_seen[-1] if len(idx_seen) else [], dtype=np.int64)
        else:
            flat = np.ravel_multi_index(tuple(np.asarray(x, dtype=np.int64) for x in idx_seen[:2]), recs.shape)
        if flat.size == 0:
            return
        ops = self.ops
        t = ops.to_device(np.ascontiguousarray(recs, dtype=np.float64).reshape(-1))
        seen = t[ops.to_device(flat)]
        recs.flat[flat] = ops.to_host(t.min() - (seen.max() - seen) - 1)

    def get_topk_elements(self, scores, topk=None):
        """models.py:561-563, dense branch: [n_rows x topk] column ids by descending score."""
        if hasattr(scores, 'tocsr'):
            raise NotImplementedError('sparse score matrices are not produced by the factorization models')
        topk = self.topk if topk is None else topk
        scores = np.asarray(scores)
        if topk > scores.shape[-1]:
            raise ValueError('kth(=%d) out of bounds (%d)' % (scores.shape[-1] - topk, scores.shape[-1]))
        t = self.ops.to_device(np.ascontiguousarray(scores, dtype=np.float64))
        if hasattr(self.ops, 'topk_rows'):       # the device backend: pk_topk_rows_f64
            flat = t.reshape(-1, t.shape[-1])
            return self.ops.to_host(self.ops.topk_rows(flat, int(topk))).reshape(scores.shape[:-1] + (int(topk),))
        import torch
        return self.ops.to_host(torch.topk(t, int(topk), dim=-1, largest=True, sorted=True).indices)

    def _user_scores(self, i):
        """models.py:277-293: dense scores of test user `i` (seen items downvoted when filter_seen) + its triplet."""
        if not self._is_ready:
            if self.verbose:
                print('{} model is not ready. Rebuilding.'.format(self.method))
            self.build()
        test_data, test_shape, test_users = self._get_test_data()
        if not self.data.warm_start:
            i, = np.where(test_users == i)[0]
        scores, seen_idx = self.slice_recommendations(test_data, test_shape, i, i + 1)
        if self.filter_seen:
            self.downvote_seen_items(scores, seen_idx)
        return scores, seen_idx

    def show_recommendations(self, user_info, topk=None):
        """models.py:320-356: top items (external ids) for a test user given by its index, or for an ad-hoc user
        given as a list of items / an {item: feedback} dict; returns (recommended items, the user's seen items)."""
        if isinstance(user_info, (int, np.integer)):
            scores, seen_idx = self._user_scores(int(user_info))
        else:
            with self._ad_hoc_test_user(user_info):
                scores, seen_idx = self._user_scores(0)
        top = self.get_topk_elements(scores, self.topk if topk is None else topk).squeeze()
        seen = np.asarray(seen_idx[1])
        index = getattr(self.data, 'get_entity_index', None)
        if index is not None:                                 # Polara's data model: internal -> external item ids
            old = index(self.data.fields.itemid).set_index('new')['old']
            return old.loc[top].values, old.loc[seen].values
        return top, seen

    def _ad_hoc_test_user(self, user_info):
        """Context manager that swaps in a one-user test set (models.py:296-317, 325-336) and restores the data
        object's own on exit."""
        import contextlib
        data = self.data
        userid, itemid, feedback = data.fields
        if isinstance(user_info, dict):
            items, fdbk = (list(x) for x in zip(*user_info.items()))
        elif isinstance(user_info, (list, tuple, set, np.ndarray)):
            items, fdbk = list(user_info), None
        else:
            raise ValueError('Unrecognized input for `user_info`.')

        @contextlib.contextmanager
        def swapped():
            saved = data._test
            saved_flag = getattr(data, 'scores_training_rows', None)
            self._test_dev = None
            try:
                if hasattr(data, 'get_entity_index'):         # Polara: a one-user frame in internal item ids
                    import pandas as pd
                    try:
                        item_index = data.index.itemid.training
                    except AttributeError:
                        item_index = data.index.itemid
                    frame = {userid: [0] * len(items), itemid: item_index.set_index('old').loc[items, 'new'].values}
                    if feedback is not None:
                        frame[feedback] = fdbk if fdbk is not None else [data.training[feedback].max()] * len(items)
                    data._test = type(saved)(pd.DataFrame(frame), None)
                else:                                         # ArrayData: ids are internal already
                    f = fdbk if fdbk is not None else [float(np.max(data.training.feedback))] * len(items)
                    data.set_test_data(testset=(np.zeros(len(items), dtype=np.int64), np.asarray(items), np.asarray(f)),
                                       notify=False)
                yield
            finally:
                data._test = saved
                self._test_dev = None
                if saved_flag is not None:
                    data.scores_training_rows = saved_flag
        return swapped()

    def evaluate(self, metric_type='all', topk=None, not_rated_penalty=None, switch_positive=None,
                 ignore_feedback=False, simple_rates=False, on_feedback_level=None):
        """models.py:408-485.  With a Polara `RecommenderData` (pandas holdout) the reference's own metric code
        runs on our recommendations; with `ArrayData` — or without Polara — polara_amd.evaluation does, the
        same formulas on arrays (see its header for the one deliberate difference)."""
        holdout = self.data.test.holdout
        if hasattr(holdout, 'columns'):              # a pandas frame: Polara's data model is in use
            from polara.recommender.models import RecommenderModel as _Ref
            return _Ref.evaluate(self, metric_type=metric_type, topk=topk, not_rated_penalty=not_rated_penalty,
                                 switch_positive=switch_positive, ignore_feedback=ignore_feedback,
                                 simple_rates=simple_rates, on_feedback_level=on_feedback_level)
        from . import evaluation
        if holdout is None:
            raise ValueError('evaluate() needs a holdout')
        if int(topk or 0) > self.topk:
            self.topk = topk                         # also flushes the cached recommendations (models.py:423-424)
        full = self.recommendations
        recs = full[:, :topk]
        users, items, fdbk = holdout
        order = np.argsort(users, kind='stable')     # rows of `recommendations` follow the sorted test users
        users, items = np.asarray(users)[order], np.asarray(items)[order]
        n_items = self.data.get_test_shape(tensor_mode=False)[1]
        device_sums = None
        cached = getattr(self, '_recs_dev', None)
        fb = None if fdbk is None else np.asarray(fdbk, dtype=np.float64)[order]
        sp = switch_positive or self.switch_positive
        if cached is not None and cached[0] is full and self._item_rank is not None:
            # every metric is a reduction over the device-resident list (pk_eval_user_metrics + pk_eval_reduce): the
            # holdout goes up (a few items per user), 16 sums and one count come back
            import torch
            ops = self.ops
            rd = cached[1]
            k_eff = int(topk) if topk else int(rd.shape[1])
            row = np.r_[0, np.cumsum(np.diff(users) != 0)] if len(users) else np.zeros(0, np.int64)
            if len(users) and row[-1] + 1 != rd.shape[0]:
                raise ValueError('recommendations have %d rows, the holdout %d users' % (rd.shape[0], row[-1] + 1))
            hold_ptr = np.r_[0, np.cumsum(np.bincount(row, minlength=rd.shape[0]))].astype(np.int64)
            split = sp is not None and fb is not None
            penalty = (1 if not_rated_penalty is None else not_rated_penalty) if not split else (not_rated_penalty or 0)
            sums = ops.eval_metrics(rd, k_eff, torch.from_numpy(hold_ptr),
                                    torch.from_numpy(self._item_rank[items.astype(np.intp)].astype(np.int64)),
                                    None if (ignore_feedback or fb is None) else torch.from_numpy(fb),
                                    torch.from_numpy((fb >= sp).astype(np.uint8)) if split else None,
                                    not_rated_penalty=penalty, switch_positive=sp if split else 0.0,
                                    alternative=get_default('ndcg_alternative'))
            n_unique = ops.unique_count(rd[:, :k_eff] if k_eff < rd.shape[1] else rd, n_items)
            device_sums = (sums, int(rd.shape[0]), n_unique)
        return evaluation.evaluate(recs, users, items, fb, n_items,
                                   device_sums=device_sums,
                                   metric_type=metric_type, not_rated_penalty=not_rated_penalty,
                                   switch_positive=sp,
                                   ignore_feedback=ignore_feedback, simple_rates=simple_rates,
                                   holdout_size=self.data.holdout_size,
                                   ndcg_alternative=get_default('ndcg_alternative'))


class SVDModel(RecommenderModel):
    """PureSVD (models.py:800-861)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._rank = defaults.svd_rank
        self.method = 'PureSVD'
        self.factors = {}
        self.svd_tol = defaults.svd_tol
        self.svd_seed = defaults.svd_seed
        self.svd_block = defaults.svd_oversample
        self.svd_max_outer = defaults.svd_max_outer
        self.svd_shard_items = defaults.svd_shard_items   # multi-GPU: row-shard the item-side blocks of the solver (solver.ItemRows); None = the solver's choice (replicated where the library runs the step)
        self.svd_on_no_convergence = 'raise'   # or 'warn': keep the best available factors (stats['converged'] False)
        self.build_stats = {}

    @property
    def rank(self):
        return self._rank

    @rank.setter
    def rank(self, new_value):
        if new_value != self._rank:
            self._rank = new_value
            self._check_reduced_rank(new_value)
            self._recommendations = None
            self._factor_image = None

    def _check_reduced_rank(self, rank):
        """models.py:819-832: a smaller rank is the leading columns of the cached factors (and the leading singular
        values) — no rebuild; a larger one empties them and marks the model as not ready."""
        have = {k: f for k, f in self.factors.items() if f is not None}
        if any(f.shape[-1] < rank for f in have.values()):
            self._is_ready = False
            self.factors = {k: None for k in self.factors}
            return
        # a NEW dict with views: whoever kept the old dict (the rank-sweep pipelines do) keeps the full factors
        self.factors = {k: (f[..., :rank] if f is not None else None) for k, f in self.factors.items()}

    def _training_device_csr(self):
        """The training matrix as a device CSR (COO -> CSR on device, models.py:160-177)."""
        blk = getattr(self.data, 'local_csr', None)
        if blk is not None and self.feedback_threshold is None:
            # a block of an on-disk CSR dataset (polara_amd/shards.py): the arrays go to the device as they lie
            # in the file, the internal item order is a renaming on the device — no COO, no sort
            shp = (blk.n_rows, blk.n_cols)
            self._item_rank, self._item_inv = popularity_order(None, shp[1], counts=self._item_counts(blk.indices, shp[1]))
            values = np.ones(blk.nnz, dtype=np.float32) if blk.values is None else blk.values
            A = self.ops.csr_relabel_cols(self.ops.csr(blk.indptr, blk.indices, values, shp), self._item_rank)
            if getattr(self.data, 'scores_training_rows', False):
                # the users to score ARE these rows (ShardedArrayData(score_all=True)): the matrix stays on the
                # device, get_recommendations renames its columns instead of rebuilding it from triplets
                self._train_dev = (A, self._item_inv)
            return A
        # COO -> CSR, per-item counts and the renaming into the internal (popularity) order all run on the device
        # (csrc/ingest.hip): the index array of `to_coo` goes up as it is
        from .data import ArrayData
        if getattr(type(self.data), 'to_coo', None) is ArrayData.to_coo:   # our own data object (to_coo not overridden): the
            # columns as they lie, no stacked index
            rows, cols, val, shp = self.data.matrix_triplets(feedback_threshold=self.feedback_threshold)
        else:
            idx, val, shp = self.data.to_coo(tensor_mode=False, feedback_threshold=self.feedback_threshold)
            rows, cols = idx[:, 0], idx[:, 1]
        A = self.ops.csr_from_coo(rows, cols, val, shp)
        self._item_rank, self._item_inv, _, rank_dev = self.ops.item_order(A, self.comm if self._presharded() else None)
        return self.ops.csr_relabel_cols(A, rank_dev)

    def _local_training_shard(self):
        A = self._training_device_csr()
        n_users = A.shape[0]
        comm = self.comm
        if self._presharded():
            lo, hi = self.data.user_range
            return A, (int(lo), int(hi), int(self.data.n_users_total))
        if comm.world > 1:
            bounds = nnz_balanced_row_partition(self.ops.to_host(A.indptr), comm.world)
            lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
            return self.ops.csr_rows(A, lo, hi), (lo, hi, n_users)
        return A, (0, n_users, n_users)

    def build(self, operator=None, return_factors='vh'):
        """models.py:835-855.  `operator` is used INSTEAD of the training matrix (models.py:838-839; HybridSVD
        passes L_K^T A L_S): a SciPy sparse matrix or an `operator.SparseProduct` of sparse factors lives on
        the device and the whole build runs there; any other LinearOperator's products run on the host with
        the block solver around them on the device (polara_amd/operator.py).  Single process only."""
        ops = self.ops
        if operator is not None:
            if self.comm.world > 1:
                raise NotImplementedError('build(operator=...) with a sharded communicator')
            idx, _, shp = self.data.to_coo(tensor_mode=False, feedback_threshold=self.feedback_threshold)
            if tuple(operator.shape)[1] != shp[1]:
                raise ValueError('operator has %d columns, the data %d items' % (operator.shape[1], shp[1]))
            self._item_rank, self._item_inv = popularity_order(idx[:, 1], shp[1])
            if hasattr(operator, 'tocsr'):          # a sparse matrix: one device CSR
                operator = SparseProduct(operator)
            if isinstance(operator, SparseProduct):
                A = DeviceChain.from_scipy(ops, operator, col_perm=self._item_rank)
                if len(A.factors) == 1:
                    A = A.factors[0]
            else:
                A = HostOperator(ops, operator, col_perm=self._item_rank)
            lo, hi, n_users = 0, A.shape[0], A.shape[0]
        else:
            A, (lo, hi, n_users) = self._local_training_shard()
        want_u = return_factors in (True, 'u')
        start = timer()
        U, sigma, V, stats = svd_topk(ops, A, self.rank, block=self.svd_block, tol=self.svd_tol,
                                      max_outer=self.svd_max_outer, seed=self.svd_seed, comm=self.comm, want_u=want_u,
                                      verbose=False, shard_items=self.svd_shard_items)
        ops.synchronize()
        self._track(start)
        self.build_stats = stats
        if not stats['converged']:
            # the reference's ARPACK raises ArpackNoConvergence here (models.py:844); every rank takes the same
            # decision: `stats` derives from all-reduced quantities only
            msg = ('%s: the block eigensolver did not converge in %d outer iterations (%d Gramian steps): worst '
                   'relative residual %.2e of the leading %d pairs, tolerance %.1e' %
                   (self.method, stats['outer'], stats['gramian_steps'], stats['final_rel_residual'], self.rank,
                    self.svd_tol))
            if self.svd_on_no_convergence == 'raise':
                raise NoConvergence(msg, sigma=ops.to_host(sigma), V=ops.to_host(V), stats=stats)
            import warnings
            warnings.warn(msg, RuntimeWarning)
        user_factors = None
        if want_u:
            user_factors = ops.to_host(U)
            if self.comm.world > 1 and not self._presharded():   # pre-sharded: the rows of the local users
                user_factors = self.comm.gather_rows(user_factors, n_users, self.rank, dtype=np.float64)
        item_factors = None
        if return_factors in (True, 'vh'):
            Vh = ops.to_host(V)
            item_factors = np.asfortranarray(Vh[self._item_rank])   # external item order
            # serving index: the internal item order becomes DESCENDING FACTOR NORM, the order in which the
            # pruning bound of the candidate sweep (a suffix maximum of these norms) falls fastest; a pure
            # relabelling, test matrices are built against it later (get_recommendations)
            by_norm = np.argsort(-np.linalg.norm(Vh, axis=1), kind='stable')
            self._item_inv = np.ascontiguousarray(self._item_inv[by_norm])
            self._item_rank = np.empty_like(self._item_inv)
            self._item_rank[self._item_inv] = np.arange(len(self._item_inv), dtype=self._item_inv.dtype)
            V = ops.to_device(np.ascontiguousarray(Vh[by_norm]))
        self.factors[self.data.fields.userid] = user_factors
        self.factors[self.data.fields.itemid] = item_factors
        self.factors['singular_values'] = ops.to_host(sigma)
        self._factor_image = scoring.FactorImage(ops, V) if item_factors is not None else None
        self._factor_src = item_factors



class ScaledMatrixMixin:
    """models.py:864-895: diagonal row/column rescaling of the training matrix before the build,
    A' = D_r A D_c with D = (sqrt(nnz per row/col))^(scaling-1)  (preprocessing/matrices.py:71-93,
    binary norm).  A two-vector epilogue on the CSR values; scoring is unchanged."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._col_scaling = 0.4
        self._row_scaling = 1
        self.method = f'{self.method}-s'

    col_scaling = _setting('col_scaling', '_refresh_model', 'exponent of the column (item) scaling, models.py:870-887')
    row_scaling = _setting('row_scaling', '_refresh_model', 'exponent of the row (user) scaling')

    def _scale_values(self, indptr, indices, values, shp):
        """A' = D_r A D_c, D = (sqrt(nnz per row / column))^(scaling - 1); the column counts are those of the WHOLE
        matrix (summed over ranks for a sharded dataset)."""
        row_nnz = np.diff(indptr).astype(np.float64)
        col_nnz = self._item_counts(indices, shp[1]).astype(np.float64)
        rs = np.ones_like(row_nnz)
        cs = np.ones_like(col_nnz)
        np.power(np.sqrt(row_nnz), self.row_scaling - 1, where=row_nnz != 0, out=rs)
        np.power(np.sqrt(col_nnz), self.col_scaling - 1, where=col_nnz != 0, out=cs)
        return (rs[np.repeat(np.arange(shp[0]), np.diff(indptr))] * values) * cs[indices]

    def _training_csr(self, dtype=np.float64, ignore_feedback=False):
        indptr, indices, values, shp = super()._training_csr(dtype=dtype, ignore_feedback=ignore_feedback)
        return indptr, indices, self._scale_values(indptr, indices, values, shp), shp

    def get_training_matrix(self, *args, **kwargs):
        """models.py:889-893: the SCALED matrix, like the reference's mixin."""
        m = super().get_training_matrix(*args, **kwargs)
        csr = m.tocsr()
        csr.data = self._scale_values(csr.indptr, csr.indices, csr.data.astype(np.float64), csr.shape)
        return csr if m.format == 'csr' else csr.asformat(m.format)

    def _training_device_csr(self):
        ops = self.ops
        if not hasattr(ops, 'csr_scale'):
            indptr, indices, values, shp = self._training_csr(dtype=np.float64)
            self._item_rank, self._item_inv = popularity_order(None, shp[1], counts=self._item_counts(indices, shp[1]))
            return ops.csr_relabel_cols(ops.csr(indptr, indices, values, shp), self._item_rank)
        # device path: the unscaled matrix through the ingest kernels (COO -> CSR, counts, popularity order), then the two
        # diagonals — vectors of n_users / n_items entries, computed on the host exactly as `_scale_values` does — applied
        # by pk_csr_scale_f64 (the host version sorts and scales 2e7 entries in NumPy: seconds against milliseconds)
        A = super()._training_device_csr()
        row_nnz = np.diff(ops.to_host(A.indptr)).astype(np.float64)
        col_nnz = ops.item_counts(A)
        if self._presharded() and self.comm.world > 1:
            col_nnz = ops.to_host(self.comm.allreduce(ops.to_device(col_nnz)))
        col_nnz = col_nnz.astype(np.float64)
        rs = np.ones_like(row_nnz)
        cs = np.ones_like(col_nnz)
        np.power(np.sqrt(row_nnz), self.row_scaling - 1, where=row_nnz != 0, out=rs)
        np.power(np.sqrt(col_nnz), self.col_scaling - 1, where=col_nnz != 0, out=cs)
        return ops.csr_scale(A, rs, cs)


class ScaledSVD(ScaledMatrixMixin, SVDModel):
    """models.py:898."""


def flatten_scores(tensor_scores, flattener=None):
    """How the feedback mode is folded away (models.py:983-1006): a level index, a list/slice of levels (summed),
    the name of a NumPy reduction, a (levels, reduction-name) pair, or any callable over the last axis."""
    flattener = flattener or slice(None)       # None — and, as in the reference, a bare level 0 — mean "sum all levels"
    if callable(flattener):
        return flattener(tensor_scores)
    if isinstance(flattener, (int, np.integer)):
        return tensor_scores[..., flattener]
    levels, reduce_name = slice(None), 'sum'
    if isinstance(flattener, str):
        reduce_name = flattener
    elif isinstance(flattener, tuple):
        levels, reduce_name = flattener[0] or slice(None), flattener[1]
    elif isinstance(flattener, (list, slice)):
        levels = flattener
    else:
        raise ValueError('Unrecognized value for flattener attribute')
    return getattr(np, reduce_name)(tensor_scores[..., levels], axis=-1)


class CoffeeModel(RecommenderModel):
    """CoFFee: Tucker/HOOI on the (user, item, feedback) tensor (models.py:901-1054)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._mlrank = defaults.mlrank
        self.factors = {}
        self.chunk = defaults.test_chunk_size
        self.method = 'CoFFee'
        self._flattener = defaults.flattener
        self.growth_tol = defaults.growth_tol
        self.num_iters = defaults.num_iters
        self.show_output = defaults.show_output
        self.seed = None
        self.parallel_ttm = defaults.parallel_ttm  # accepted; the device TTM is always parallel
        self.core_norm_trace = []

    @property
    def mlrank(self):
        return self._mlrank

    @mlrank.setter
    def mlrank(self, new_value):
        if new_value != self._mlrank:
            self._mlrank = new_value
            self._check_reduced_rank(new_value)
            self._recommendations = None
            self._factor_image = None

    flattener = _setting('flattener', '_refresh_model', 'how the feedback mode is folded away (see flatten_scores)')

    @staticmethod
    def round_core(core, mode, rank):
        """Truncated SVD of the mode-`mode` unfolding of the core (models.py:968-980): returns the rotation of that
        mode's factor [r_mode x rank] and the core with that mode cut to `rank`."""
        unfolded = np.moveaxis(core, mode, 0)
        rest = unfolded.shape[1:]
        u, s, vt = np.linalg.svd(unfolded.reshape((core.shape[mode], -1), order='F'), full_matrices=False)
        cut = np.ascontiguousarray(s[:rank, None] * vt[:rank]).reshape((rank,) + rest, order='F')
        return u[:, :rank], np.moveaxis(cut, 0, mode)

    def _check_reduced_rank(self, mlrank):
        """models.py:949-965: a smaller multilinear rank is served from the cached factors (each reduced mode is
        rotated onto the leading singular directions of the core's unfolding, the core shrinks); a larger one
        invalidates the model."""
        cached = [(mode, entity, self.factors.get(entity, None)) for mode, entity in enumerate(self.data.fields)]
        cached = [(mode, entity, f) for mode, entity, f in cached if f is not None]
        if any(f.shape[1] < mlrank[mode] for mode, _, f in cached):
            self._is_ready = False
            self.factors = {}
            return
        for mode, entity, f in cached:
            if f.shape[1] > mlrank[mode]:
                self.factors = dict(**self.factors)          # never write into a dict somebody may have kept
                rotation, self.factors['core'] = self.round_core(self.factors['core'], mode, mlrank[mode])
                self.factors[entity] = f.dot(rotation)

    def build(self):
        """models.py:1009-1024."""
        ops, comm = self.ops, self.comm
        presharded = self._presharded()
        from .data import ArrayData
        if (getattr(type(self.data), 'to_coo', None) is ArrayData.to_coo and hasattr(self.data, 'tensor_triplets')
                and (comm.world == 1 or presharded)):
            # our own data object: the three columns go up as they lie; feedback levels, item counts and the renaming
            # into the internal item order are device passes (no stacked index, no host pass over the entries)
            u, i, f, levels, shp = self.data.tensor_triplets()
            i0, i1, i2, bad = tucker.device_coordinates(ops, u, i, f, levels)
            counts = ops.bincount(i1, shp[1])
            counts_bad = torch.cat([counts.to(torch.int64), bad.reshape(1).to(torch.int64)])
            if presharded and comm.world > 1:
                # the count of entries outside the level set rides in the same sum: EVERY rank raises below, not only
                # the one that holds the offending entries (the others would wait in the next collective for ever)
                counts_bad = comm.allreduce(counts_bad)
            counts_bad = ops.to_host(counts_bad)                                              # one host read for both
            if int(counts_bad[-1]):
                raise ValueError('Not all values of feedback are present in the feedback levels of the training data '
                                 '(%d entries)' % int(counts_bad[-1]))
            self._item_rank, self._item_inv = popularity_order(None, shp[1], counts=counts_bad[:-1])
            idx, val = (i0, ops.to_device(np.asarray(self._item_rank, dtype=np.int64))[i1], i2), None
        else:
            idx, val, shp = self.data.to_coo(tensor_mode=True)
            self._item_rank, self._item_inv = popularity_order(None, shp[1], counts=self._item_counts(idx[:, 1], shp[1]))
            idx = idx.copy()
            idx[:, 1] = self._item_rank[idx[:, 1]]
        user_range = None
        if presharded:
            # the data object already is this rank's users (ids re-based to 0): only the shape is global
            user_range = tuple(int(x) for x in self.data.user_range)
            shp = (int(self.data.n_users_total),) + tuple(shp[1:])
        elif comm.world > 1:
            # users are sharded in nnz-balanced contiguous blocks; items / feedback factors replicated
            order = np.argsort(idx[:, 0], kind='stable')
            idx, val = idx[order], val[order]
            counts = np.bincount(idx[:, 0], minlength=shp[0])
            bounds = nnz_balanced_row_partition(np.r_[0, np.cumsum(counts)], comm.world)
            lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
            sel = (idx[:, 0] >= lo) & (idx[:, 0] < hi)
            idx = idx[sel].copy()
            val = val[sel]
            idx[:, 0] -= lo
            user_range = (lo, hi)
        start = timer()
        u0, u1, u2, core, trace = tucker.hooi(ops, idx, val, shp, self.mlrank, num_iters=self.num_iters,
                                              growth_tol=self.growth_tol, seed=self.seed,
                                              verbose=self.show_output, comm=comm, user_range=user_range,
                                              item_inv=self._item_inv)
        ops.synchronize()
        self._track(start)
        self.core_norm_trace = trace
        userid, itemid, feedback = self.data.fields
        u0_host = ops.to_host(u0)
        if comm.world > 1 and not presharded:
            u0_host = comm.gather_rows(u0_host, shp[0], u0_host.shape[1], dtype=np.float64)
        self.factors[userid] = u0_host
        self.factors[itemid] = np.ascontiguousarray(ops.to_host(u1)[self._item_rank])   # external item order
        self.factors[feedback] = ops.to_host(u2)
        self.factors['core'] = ops.to_host(core)
        self._factor_image = None


    # ---- the extras of the reference's class (models.py:1027-1092) ------------------------------------------------
    def _holdout_pairs(self):
        """(users, items) of the holdout as int64 arrays: Polara's frame (models.py:1075-1077) or ArrayData's triplets."""
        hold = self.data.test.holdout
        if hold is None:
            raise ValueError('the data object has no holdout')
        f = self.data.fields
        if hasattr(hold, 'userid') and not hasattr(hold, 'loc'):          # ArrayData: Triplets
            return np.asarray(hold.userid, dtype=np.int64), np.asarray(hold.itemid, dtype=np.int64)
        return hold[f.userid].values.astype(np.int64), hold[f.itemid].values.astype(np.int64)

    def get_holdout_slice(self, start, stop):
        """models.py:1056-1065: the holdout entries of test users [start, stop), user ids re-based to the slice."""
        users, items = self._holdout_pairs()
        sel = (users >= start) & (users < stop)
        return users[sel] - start, items[sel]

    def unfold_test_tensor_slice(self, test_data, shape, start, stop, mode):
        """models.py:1027-1039: the binary test tensor of users [start, stop) unfolded along `mode` — `mode` is the column
        index, the two other modes (in order) are flattened C-style into the row index (lib/sparse.py:178-187) — as a
        uint8 SciPy CSR matrix, with the slice's coordinates.  Index bookkeeping on the host, like the reference's."""
        import scipy.sparse as sps
        slice_idx = self._slice_test_data(test_data, start, stop)
        slice_shp = (stop - start, shape[1], shape[2])
        modes = [m for m in (0, 1, 2) if m != mode] + [mode]
        rows = np.asarray(slice_idx[modes[0]], dtype=np.int64) * slice_shp[modes[1]] + np.asarray(slice_idx[modes[1]], dtype=np.int64)
        cols = np.asarray(slice_idx[modes[2]], dtype=np.int64)
        val = np.ones(len(rows), dtype=np.uint8)
        unfolded = sps.csr_matrix((val, (rows, cols)), shape=(slice_shp[modes[0]] * slice_shp[modes[1]], slice_shp[modes[2]]),
                                  dtype=np.uint8)
        return unfolded, slice_idx

    def predict_feedback(self):
        """models.py:1068-1091: the feedback value the Tucker model reconstructs best for every holdout (user, item) pair
        — argmax over the feedback levels of  sum_abc core[a,b,c] u[user,a] v[item,b] w[f,c], on the device
        (pk_tucker_predict_f64: the factor rows are gathered and contracted there; the reference builds an
        [r0 x n_holdout x r2] intermediate with tensordot).  Returns the ORIGINAL feedback values, like the reference."""
        if self.data.warm_start:
            raise NotImplementedError
        if not self._is_ready:
            if self.verbose:
                print('{} model is not ready. Rebuilding.'.format(self.method))
            self.build()
        f = self.data.fields
        users, items = self._holdout_pairs()
        pred, _ = self.ops.tucker_predict(users, items, self.factors[f.userid], self.factors[f.itemid], self.factors[f.feedback],
                                          self.factors['core'])
        pred = self.ops.to_host(pred)
        index = getattr(self.data, 'index', None)
        fb = getattr(index, 'feedback', None) if index is not None else None
        if fb is not None and hasattr(fb, 'set_index'):                   # Polara: data.index.feedback maps new -> old
            return fb.set_index('new').loc[pred, 'old'].values
        return np.asarray(self.data._levels())[pred]

    def _test_csr_depends_on(self):
        # the per-entry weights come from the feedback factor and the flattener: a rank reduction or a restored
        # `factors` dict (the rank-sweep pipelines do that) makes a cached test CSR stale
        return (self._item_rank, self.factors.get(self.data.fields.feedback, None), self._flattener)

    def _test_weights(self, test_data):
        """models.py:1042-1054 folded algebraically: the per-nnz outer products,
        the reduceat over users and the tensordot with the flattened feedback factor collapse to a
        per-entry coefficient c_f = W[f, :] . flatten(W^T)."""
        w = self.factors[self.data.fields.feedback]
        wt_flat = flatten_scores(w.T, self.flattener)
        coef = w.dot(wt_flat)
        return coef[np.asarray(test_data[2], dtype=np.intp)]
