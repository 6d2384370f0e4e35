"""Local MovieLens ratings -> the CSR the hot path consumes (no download: this package never touches the network).

The reference's loader (polara/datasets/movielens.py:11-80) reads the ratings member of an `ml-1m` / `ml-20m` archive —
`::`-separated without a header in the old format, a comma-separated file with a header in the new one — into a frame with
the columns userid / movieid / rating; `RecommenderData` then renumbers users and items contiguously in sorted order of
their ids (data.py `reindex(sort=True)`).  `load_movielens` does both steps for a file that is already on disk — the zip,
or an extracted `ratings.csv` / `ratings.dat` — and returns what `synth.planted_csr` returns, so that `bench.py` can run
the headline configuration on the real matrix whenever one is supplied:

    data/ml-20m.zip | data/ml-20m/ratings.csv          (138 493 x 26 744, 20 000 263 ratings, 10 levels)
    data/ml-1m.zip  | data/ml-1m/ratings.dat           (6 040 x 3 706, 1 000 209 ratings, 5 levels)
"""
import io
import os
import zipfile

import numpy as np

CANDIDATES = {
    'ml20m': ('ml-20m.zip', os.path.join('ml-20m', 'ratings.csv'), 'ml-20m-ratings.csv'),
    'ml1m': ('ml-1m.zip', os.path.join('ml-1m', 'ratings.dat'), 'ml-1m-ratings.dat'),
}


def find_movielens(workload, root):
    """the first existing candidate file of `workload` under root/data, or None"""
    for name in CANDIDATES.get(workload, ()):
        path = os.path.join(root, 'data', name)
        if os.path.exists(path):
            return path
    return None


def _ratings_bytes(path):
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            member = [n for n in z.namelist() if 'ratings' in n][0]       # movielens.py:33
            return z.read(member), ('latest' in member) or ('20m' in member) or member.endswith('.csv')
    with open(path, 'rb') as f:
        return f.read(), path.endswith('.csv')


def load_movielens(path):
    """dict(indptr int64, indices int32, values float32, shape, users, items): canonical CSR (rows and columns sorted,
    duplicate (user, item) pairs summed like `coo_matrix(...).tocsr()`, models.py:172-175) with users / items renumbered
    in sorted order of their ids; `users` / `items` hold the original ids of the rows / columns."""
    import pandas as pd
    raw, new_format = _ratings_bytes(path)
    raw = raw.replace(b'::', b',')                                           # movielens.py:40
    df = pd.read_csv(io.BytesIO(raw), sep=',', header=0 if new_format else None, engine='c',
                     names=['userid', 'movieid', 'rating', 'timestamp'], usecols=['userid', 'movieid', 'rating'])
    users, u = np.unique(df['userid'].values, return_inverse=True)
    items, i = np.unique(df['movieid'].values, return_inverse=True)
    from .csr import coo_to_csr
    indptr, indices, values = coo_to_csr(u, i, df['rating'].values.astype(np.float64), (len(users), len(items)))
    return dict(indptr=indptr, indices=indices.astype(np.int32), values=values.astype(np.float32),
                shape=(len(users), len(items)), users=users, items=items)
