"""Model/compute defaults of the hot path, mirroring the values of the reference's
polara/recommender/defaults.py (:16-51) for the settings this package honours.
Kept as one table so `get_config` (defaults.py:57-60 in the reference) has the same contract."""

_TABLE = dict(
    # data-side defaults the minimal provider understands (defaults.py:5-14)
    test_ratio=0.2, test_fold=5, warm_start=True, holdout_size=3, test_sample=None,
    # models (defaults.py:17-30)
    feedback_threshold=None, switch_positive=None, verify_integrity=True,
    svd_rank=10,
    mlrank=(13, 10, 2), growth_tol=0.0001, num_iters=25, show_output=False,
    flattener=slice(0, None), parallel_ttm=False, test_vectorize_target='parallel',
    # recommendations (defaults.py:41-42)
    topk=10, filter_seen=True,
    # evaluation (defaults.py:46): exponential relevance contribution in NDCG
    ndcg_alternative=True,
    # computation (defaults.py:48-51): accepted for API compatibility; the fused device path
    # has no dense score chunks, hence no use for the host-memory cap
    test_chunk_size=1000, max_test_workers=None, memory_hard_limit=1,
    # device solver knobs (new)
    svd_tol=1e-12, svd_oversample=None, svd_seed=0, svd_max_outer=200, svd_shard_items=None,
)

globals().update(_TABLE)


def get_config(params):
    return {p: globals()[p] for p in params}
