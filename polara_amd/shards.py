"""On-disk CSR shards of a user-item matrix: the ingest format of the multi-GPU path.

The reference reads a pandas frame and goes frame -> COO -> `tocsr()` on one host thread for every model
(polara/recommender/data.py:794-817, models.py:160-177).  With one process per GPU that preparation would be
repeated on every rank and dwarf the build (0.4 s for 1e8 nnz).  A dataset is therefore written ONCE as
contiguous user blocks in CSR form, nnz-balanced like the in-memory partition (csr.nnz_balanced_row_partition),
and every rank maps only its own block(s): no parsing, no sort, no COO — the three arrays go to the device as
they lie in the file.

Layout of a dataset directory
    manifest.json    {"format": "pkcsr", "version": 1, "n_rows", "n_cols", "nnz", "value_dtype": "f32"|"f64"|"none",
                      "row_bounds": [0, ..., n_rows], "shard_nnz": [...], "files": [...], "feedback_levels": [...]|null}
    shard_00000.pkcsr ...
Layout of a shard file (little endian; every section starts on a 4096-byte boundary so that it can be mapped,
read with O_DIRECT or registered with the HIP runtime as it is)
    header   64 B : magic "PKCSR\\0\\0\\1", u32 version, u32 value kind (0 none / 1 f32 / 2 f64),
                    i64 row_lo, i64 row_hi, i64 n_cols, i64 nnz, 16 B reserved
    indptr   int64[row_hi - row_lo + 1]   local (indptr[0] = 0)
    indices  int32[nnz]                   column ids, ascending within a row
    values   f32|f64[nnz]                 absent for kind 0 (implicit ones)
"""
import json
import os
import struct

import numpy as np

from .csr import nnz_balanced_row_partition

MAGIC = b'PKCSR\x00\x00\x01'
VERSION = 1
ALIGN = 4096
HEADER = struct.Struct('<8sIIqqqq16x')
_KINDS = {'none': 0, 'f32': 1, 'f64': 2}
_KIND_DTYPE = {0: None, 1: np.dtype('<f4'), 2: np.dtype('<f8')}


def _pad(n):
    return -(-n // ALIGN) * ALIGN


def _kind_of(value_dtype):
    if value_dtype is None:
        return 0
    dt = np.dtype(value_dtype)
    if dt == np.float32:
        return 1
    if dt == np.float64:
        return 2
    raise ValueError('values are stored as float32, float64 or not at all; got %s' % dt)


class CSRShard:
    """One row block: `indptr` is local, `row_lo`/`row_hi` place it in the full matrix."""

    def __init__(self, row_lo, row_hi, n_cols, indptr, indices, values):
        self.row_lo, self.row_hi, self.n_cols = int(row_lo), int(row_hi), int(n_cols)
        self.indptr, self.indices, self.values = indptr, indices, values

    @property
    def n_rows(self):
        return self.row_hi - self.row_lo

    @property
    def nnz(self):
        return int(self.indptr[-1])


def _validate_block(indptr, indices, values, n_cols):
    if indptr.ndim != 1 or len(indptr) < 1 or indptr[0] != 0:
        raise ValueError('indptr must be a local row pointer array starting at 0')
    if (np.diff(indptr) < 0).any():
        raise ValueError('indptr must be non-decreasing')
    nnz = int(indptr[-1])
    if len(indices) != nnz or (values is not None and len(values) != nnz):
        raise ValueError('indices/values length does not match indptr[-1] = %d' % nnz)
    if nnz and (int(indices.min()) < 0 or int(indices.max()) >= n_cols):
        raise ValueError('column index out of bounds')


class ShardWriter:
    """Streams a dataset to `path`, one row block per `add_shard` call, blocks in row order (the way a
    generator produces a matrix too large for one host: SURVEY.md §8d, S-50M)."""

    def __init__(self, path, n_cols, value_dtype=np.float32, feedback_levels=None):
        self.path = path
        self.n_cols = int(n_cols)
        self.kind = _kind_of(value_dtype)
        self.row_bounds = [0]
        self.shard_nnz = []
        self.files = []
        self.feedback_levels = None if feedback_levels is None else [float(x) for x in feedback_levels]
        os.makedirs(path, exist_ok=True)
        if os.path.exists(os.path.join(path, 'manifest.json')):
            raise FileExistsError('%s already holds a dataset' % path)

    def add_shard(self, indptr, indices, values=None):
        indptr = np.ascontiguousarray(indptr, dtype='<i8')
        indices = np.ascontiguousarray(indices, dtype='<i4')
        if self.kind == 0:
            values = None
        elif values is None:
            raise ValueError('this dataset stores values')
        else:
            values = np.ascontiguousarray(values, dtype=_KIND_DTYPE[self.kind])
        _validate_block(indptr, indices, values, self.n_cols)
        lo = self.row_bounds[-1]
        hi = lo + len(indptr) - 1
        name = 'shard_%05d.pkcsr' % len(self.files)
        tmp = os.path.join(self.path, name + '.tmp')
        with open(tmp, 'wb') as f:
            f.write(HEADER.pack(MAGIC, VERSION, self.kind, lo, hi, self.n_cols, int(indptr[-1])))
            for arr in (indptr, indices, values):
                if arr is None:
                    continue
                f.seek(_pad(f.tell()))
                f.write(memoryview(arr).cast('B'))
            f.truncate(_pad(f.tell()))
        os.replace(tmp, os.path.join(self.path, name))
        self.row_bounds.append(hi)
        self.shard_nnz.append(int(indptr[-1]))
        self.files.append(name)

    def close(self):
        manifest = dict(format='pkcsr', version=VERSION, n_rows=self.row_bounds[-1], n_cols=self.n_cols,
                        nnz=int(sum(self.shard_nnz)), value_dtype={v: k for k, v in _KINDS.items()}[self.kind],
                        row_bounds=self.row_bounds, shard_nnz=self.shard_nnz, files=self.files,
                        feedback_levels=self.feedback_levels)
        tmp = os.path.join(self.path, 'manifest.json.tmp')
        with open(tmp, 'w') as f:
            json.dump(manifest, f)
        os.replace(tmp, os.path.join(self.path, 'manifest.json'))   # the manifest appears last: a dataset is complete or absent
        return manifest

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()


def write_csr_shards(path, indptr, indices, values, n_cols, n_shards, value_dtype=np.float32, feedback_levels=None):
    """Writes an in-memory CSR as `n_shards` nnz-balanced contiguous row blocks.  `feedback_levels` defaults to
    the distinct values when there are at most 64 of them (CoFFee's third mode needs the SAME level set on
    every rank, data.py:802-806)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    if values is None:
        value_dtype = None
    elif feedback_levels is None:
        levels = np.unique(values)
        feedback_levels = levels if len(levels) <= 64 else None
    bounds = nnz_balanced_row_partition(indptr, int(n_shards))
    with ShardWriter(path, n_cols, value_dtype, feedback_levels) as w:
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            p0, p1 = int(indptr[lo]), int(indptr[hi])
            w.add_shard(indptr[lo:hi + 1] - p0, indices[p0:p1], None if values is None else values[p0:p1])
    return read_manifest(path)


def read_manifest(path):
    fn = os.path.join(path, 'manifest.json')
    if not os.path.exists(fn):
        raise FileNotFoundError('%s: no manifest.json (incomplete or missing dataset)' % path)
    with open(fn) as f:
        m = json.load(f)
    if m.get('format') != 'pkcsr' or m.get('version') != VERSION:
        raise ValueError('%s: not a pkcsr v%d dataset' % (path, VERSION))
    if len(m['row_bounds']) != len(m['files']) + 1 or m['row_bounds'][-1] != m['n_rows']:
        raise ValueError('%s: inconsistent manifest' % path)
    return m


def open_shard(path, k, manifest=None, mmap=True):
    """Shard `k` as a CSRShard whose arrays are private (copy-on-write: the file is never written) maps of the
    file (mmap=True) or host copies."""
    m = manifest or read_manifest(path)
    fn = os.path.join(path, m['files'][k])
    with open(fn, 'rb') as f:
        head = f.read(HEADER.size)
    if len(head) != HEADER.size:
        raise ValueError('%s: truncated header' % fn)
    magic, version, kind, lo, hi, n_cols, nnz = HEADER.unpack(head)
    if magic != MAGIC or version != VERSION or kind not in _KIND_DTYPE:
        raise ValueError('%s: not a pkcsr v%d shard' % (fn, VERSION))
    if (lo, hi) != (m['row_bounds'][k], m['row_bounds'][k + 1]) or n_cols != m['n_cols'] or nnz != m['shard_nnz'][k]:
        raise ValueError('%s: header disagrees with the manifest' % fn)
    n_rows = hi - lo
    off_ptr = _pad(HEADER.size)
    off_idx = _pad(off_ptr + 8 * (n_rows + 1))
    off_val = _pad(off_idx + 4 * nnz)
    vdt = _KIND_DTYPE[kind]
    need = off_val + (0 if vdt is None else vdt.itemsize * nnz)
    if os.path.getsize(fn) < need:
        raise ValueError('%s: truncated (%d bytes, %d needed)' % (fn, os.path.getsize(fn), need))

    def section(dtype, offset, count):
        if mmap and count:
            return np.memmap(fn, dtype=dtype, mode='c', offset=offset, shape=(count,))
        return np.fromfile(fn, dtype=dtype, count=count, offset=offset)
    indptr = section('<i8', off_ptr, n_rows + 1)
    indices = section('<i4', off_idx, nnz)
    values = None if vdt is None else section(vdt, off_val, nnz)
    if indptr[0] != 0 or int(indptr[-1]) != nnz:
        raise ValueError('%s: corrupt row pointers' % fn)
    return CSRShard(lo, hi, n_cols, indptr, indices, values)


def shards_for_rank(manifest, rank, world):
    """The contiguous run of shards rank `rank` of `world` owns: runs balanced by nnz, in shard order, so that
    rank order = row order (what TorchComm.gather_rows relies on).  Needs at least one shard per rank."""
    n = len(manifest['files'])
    if n < world:
        raise ValueError('%d shards cannot feed %d ranks; rewrite the dataset with more shards' % (n, world))
    cum = np.r_[0, np.cumsum(manifest['shard_nnz'], dtype=np.int64)]
    bounds = nnz_balanced_row_partition(cum, world)
    # every rank gets at least one shard: push empty runs forward, then backward
    for r in range(1, world + 1):
        bounds[r] = max(bounds[r], bounds[r - 1] + 1)
    bounds[world] = n
    for r in range(world - 1, 0, -1):
        bounds[r] = min(bounds[r], bounds[r + 1] - 1)
    return list(range(int(bounds[rank]), int(bounds[rank + 1])))


def load_rank_block(path, rank=0, world=1, mmap=True):
    """The row block of rank `rank`: one CSRShard (several files are concatenated on the host, a single file
    stays a map).  Returns (block, manifest)."""
    m = read_manifest(path)
    parts = [open_shard(path, k, m, mmap) for k in shards_for_rank(m, rank, world)]
    if len(parts) == 1:
        return parts[0], m
    offs = np.r_[0, np.cumsum([p.nnz for p in parts])].astype(np.int64)
    indptr = np.concatenate([np.asarray(parts[0].indptr)] + [np.asarray(p.indptr[1:]) + o for p, o in zip(parts[1:], offs[1:])])
    indices = np.concatenate([np.asarray(p.indices) for p in parts])
    values = None if parts[0].values is None else np.concatenate([np.asarray(p.values) for p in parts])
    return CSRShard(parts[0].row_lo, parts[-1].row_hi, m['n_cols'], indptr, indices, values), m
