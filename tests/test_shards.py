"""The on-disk CSR shard format (polara_amd/shards.py) and the pre-sharded data object built on it."""
import json
import os

import numpy as np
import pytest

from polara_amd import shards
from polara_amd.data import ShardedArrayData
from polara_amd.synth import planted_csr, csr_to_numpy


def _matrix(seed=3, n_users=300, n_items=90):
    c = csr_to_numpy(planted_csr(n_users, n_items, mean_items=12, rank=4, seed=seed, min_items=0, max_items=30))
    return c['indptr'], c['indices'], c['values'], c['shape']


@pytest.mark.parametrize('n_shards', [1, 3, 7])
@pytest.mark.parametrize('vdt', [np.float32, np.float64, None])
def test_round_trip(tmp_path, n_shards, vdt):
    indptr, indices, values, shape = _matrix()
    path = str(tmp_path / 'ds')
    m = shards.write_csr_shards(path, indptr, indices, None if vdt is None else values, shape[1], n_shards, value_dtype=vdt)
    assert m['n_rows'] == shape[0] and m['nnz'] == len(indices) and len(m['files']) == n_shards
    assert m['feedback_levels'] == (None if vdt is None else sorted(set(values.tolist())))
    # nnz-balanced: no shard above its fair share by more than the longest row
    assert max(m['shard_nnz']) <= len(indices) / n_shards + np.diff(indptr).max()
    got_ptr, got_idx, got_val = [0], [], []
    for k in range(n_shards):
        for mm in (True, False):
            s = shards.open_shard(path, k, mmap=mm)
            assert (s.row_lo, s.row_hi) == (m['row_bounds'][k], m['row_bounds'][k + 1])
            assert s.indptr.dtype == np.int64 and s.indices.dtype == np.int32
            assert os.path.getsize(os.path.join(path, m['files'][k])) % shards.ALIGN == 0
        got_ptr += (np.asarray(s.indptr[1:]) + got_ptr[-1]).tolist()
        got_idx.append(np.asarray(s.indices))
        got_val.append(None if vdt is None else np.asarray(s.values))
    assert np.array_equal(got_ptr, indptr) and np.array_equal(np.concatenate(got_idx), indices)
    if vdt is not None:
        assert np.array_equal(np.concatenate(got_val), values.astype(vdt)) and got_val[0].dtype == vdt


@pytest.mark.parametrize('world', [1, 2, 3, 5])
def test_rank_blocks_tile_the_matrix(tmp_path, world):
    indptr, indices, values, shape = _matrix(seed=5)
    path = str(tmp_path / 'ds')
    m = shards.write_csr_shards(path, indptr, indices, values, shape[1], 5)
    owned = [shards.shards_for_rank(m, r, world) for r in range(world)]
    assert sum(owned, []) == list(range(5)) and all(owned)      # contiguous runs in rank order, none empty
    rows = 0
    for r in range(world):
        blk, _ = shards.load_rank_block(path, r, world)
        assert blk.row_lo == rows
        p0, p1 = indptr[blk.row_lo], indptr[blk.row_hi]
        assert np.array_equal(np.asarray(blk.indptr), indptr[blk.row_lo:blk.row_hi + 1] - p0)
        assert np.array_equal(np.asarray(blk.indices), indices[p0:p1])
        assert np.array_equal(np.asarray(blk.values), values[p0:p1])
        rows = blk.row_hi
    assert rows == shape[0]
    with pytest.raises(ValueError):
        shards.shards_for_rank(m, 0, 6)


def test_streaming_writer_empty_blocks_and_errors(tmp_path):
    path = str(tmp_path / 'ds')
    with shards.ShardWriter(path, n_cols=10, value_dtype=None) as w:
        w.add_shard([0, 2, 2, 3], [1, 9, 0])
        w.add_shard([0], [])                       # a block without rows
        w.add_shard([0, 0, 0], [])                 # rows without entries
        with pytest.raises(ValueError):
            w.add_shard([0, 1], [10])              # column out of bounds
        with pytest.raises(ValueError):
            w.add_shard([1, 2], [0])               # not a local row pointer
        with pytest.raises(ValueError):
            w.add_shard([0, 2], [0])               # length mismatch
    m = shards.read_manifest(path)
    assert m['row_bounds'] == [0, 3, 3, 5] and m['shard_nnz'] == [3, 0, 0] and m['value_dtype'] == 'none'
    s = shards.open_shard(path, 2)
    assert s.n_rows == 2 and s.nnz == 0 and s.values is None and len(s.indices) == 0
    blk, _ = shards.load_rank_block(path, 0, 1)
    assert np.array_equal(blk.indptr, [0, 2, 2, 3, 3, 3]) and np.array_equal(blk.indices, [1, 9, 0])
    with pytest.raises(FileExistsError):
        shards.ShardWriter(path, n_cols=10)
    with pytest.raises(ValueError):
        shards.ShardWriter(str(tmp_path / 'x'), n_cols=10, value_dtype=np.int32)


def test_corruption_is_detected(tmp_path):
    indptr, indices, values, shape = _matrix(seed=7)
    path = str(tmp_path / 'ds')
    m = shards.write_csr_shards(path, indptr, indices, values, shape[1], 2)
    fn = os.path.join(path, m['files'][1])
    with pytest.raises(FileNotFoundError):
        shards.read_manifest(str(tmp_path / 'missing'))
    blob = open(fn, 'rb').read()
    open(fn, 'wb').write(blob[:len(blob) // 2])                   # truncated
    with pytest.raises(ValueError, match='truncated'):
        shards.open_shard(path, 1)
    open(fn, 'wb').write(b'NOTPKCSR' + blob[8:])                   # wrong magic
    with pytest.raises(ValueError, match='not a pkcsr'):
        shards.open_shard(path, 1)
    open(fn, 'wb').write(blob)
    bad = dict(m, shard_nnz=[m['shard_nnz'][0], m['shard_nnz'][1] + 1])
    with pytest.raises(ValueError, match='disagrees'):
        shards.open_shard(path, 1, manifest=bad)
    json.dump(dict(m, version=99), open(os.path.join(path, 'manifest.json'), 'w'))
    with pytest.raises(ValueError):
        shards.read_manifest(path)


def test_presharded_single_block_equals_array_data(tmp_path):
    """A one-block dataset through ShardedArrayData gives the model the same matrix as the triplets do."""
    from numpy_ops import NumpyOps
    from polara_amd.data import ArrayData
    from polara_amd.models import SVDModel, ScaledSVD, CoffeeModel
    indptr, indices, values, shape = _matrix(seed=11, n_users=200, n_items=60)
    path = str(tmp_path / 'ds')
    shards.write_csr_shards(path, indptr, indices, values, shape[1], 1, value_dtype=np.float64)
    sd = ShardedArrayData.from_shards(path)
    assert sd.user_range == (0, shape[0]) and sd.n_users_total == shape[0]
    u = np.repeat(np.arange(shape[0]), np.diff(indptr))
    ad = ArrayData((u, indices, values), n_users=shape[0], n_items=shape[1], test=(u, indices, values))
    for cls, cfg in ((SVDModel, dict(rank=6)), (ScaledSVD, dict(rank=6)), (CoffeeModel, dict(mlrank=(5, 4, 3), seed=0))):
        out = []
        for data in (sd, ad):
            m = cls(data, ops=NumpyOps())
            m.verbose = False
            for k, v in cfg.items():
                setattr(m, k, v)
            m.topk = 5
            m.build()
            out.append((m.factors[data.fields.itemid], m.get_recommendations()))
            if data is sd:
                # plain SVD scores the device-resident training matrix itself (no triplets, no sort); the scaled and
                # the tensor model need other values per entry and take the general route
                assert (m._train_dev is not None) == (cls is SVDModel) and sd.scores_training_rows
                if cls is SVDModel:
                    assert out[-1][1].shape[0] == int((np.diff(indptr) > 0).sum()) < shape[0]   # users without a row: no list
                    top, seen = m.show_recommendations([1, 2, 3], topk=4)      # an ad-hoc user swaps the test set ...
                    assert set(seen) == {1, 2, 3} and sd.scores_training_rows  # ... and everything is put back
                    assert np.array_equal(m.get_recommendations(), out[-1][1])
        assert np.allclose(out[0][0] @ out[0][0].T, out[1][0] @ out[1][0].T, atol=1e-9)
        assert np.array_equal(out[0][1], out[1][1])
    # a block of a larger dataset cannot be modelled without its job's communicator
    part = shards.CSRShard(0, 100, shape[1], indptr[:101], indices[:indptr[100]], values[:indptr[100]])
    with pytest.raises(ValueError, match='communicator'):
        SVDModel(ShardedArrayData(part, shape[0]), ops=NumpyOps()).build()
    with pytest.raises(NotImplementedError):
        sd.set_training_data((u, indices, values))
