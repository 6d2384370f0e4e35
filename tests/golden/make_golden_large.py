"""Generates tests/golden/coffee_ml1m.npz: BASELINE.json configs[3] — CoFFee / HOOI on the ML-1M-shaped tensor
(6040 x 3706 x 5, ~1e6 nnz) at mlrank (30, 30, 4) — from the PINNED ORACLE (oracle/polara_oracle.py, asserted
bit-equal to the imported reference on every small fixture by make_golden.py).

Why the oracle and not the reference itself at this size: the reference's TTM is a numba-jitted scalar loop
(lib/sparse.py:203-216); numba is not installed here, and the un-jitted Python loop nest needs ~1e6 x 120 x 3 x 11
interpreted iterations.  The oracle's `dttm_seq` applies the same updates in the same order through `np.add.at`
(bit-equal to the reference's loop on the small fixtures, `make_golden.py: coffee_fixture`).  The rank reduction
(a15) stored next to it IS the reference's code: `CoffeeModel.round_core` (models.py:968-980) imported from
/root/reference and applied to the oracle's factors exactly as `_check_reduced_rank` (models.py:949-965) does.

The inputs are not stored (1e6 triplets): they are `polara_amd.synth.make_workload('ml1m')` on the CPU generator,
re-created by the test; a digest of the triplets is stored so that a generator drift is reported as such.

usage:  python tests/golden/make_golden_large.py        (build container, a few minutes on one BLAS thread)
"""
import hashlib
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, '_numba_shim'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

import numpy as np
from threadpoolctl import threadpool_limits

# ONE BLAS thread: with a threaded BLAS the reductions inside `svds` / `dot` change their summation order from run to run
# and the float arrays of the fixture move in their last digits (5e-16 ... 3e-15 relative between two
# generations; every integer array was equal).  Single-threaded, two generations ON THE SAME MACHINE are byte-identical;
# on another machine (another BLAS kernel selection) the float arrays still differ at 3e-16 ... 6e-15 while
# every integer array — the lists, the tie / clear flags — is equal.  So the claim is: integer arrays reproduce byte for
# byte anywhere, float arrays to 1e-13, and `--check` verifies exactly that against the committed file instead of
# overwriting it.  (The tests use the fixture at 1e-8 / 1e-9; it pins the ORACLE, whose un-jitted reference loop cannot
# run 1e6 entries.)
_one_thread = threadpool_limits(limits=1)

from polara.recommender.models import CoffeeModel as RefCoffee   # the reference (round_core only)

from oracle import polara_oracle as orc
from polara_amd.data import ArrayData
from polara_amd.synth import make_workload, csr_to_coo_triplets

MLRANK = (30, 30, 4)
REDUCED = (20, 15, 3)
TOPK = 10
SEED = 0
NUM_ITERS, GROWTH_TOL = 25, 1e-4          # defaults.py:25-26 of the reference
PROBE = 96


def triplet_digest(idx):
    return hashlib.sha1(np.ascontiguousarray(idx, dtype=np.int64).tobytes()).hexdigest()


def coffee_inputs():
    csr, _ = make_workload('ml1m')                       # CPU generator, seed 1
    u, i, v = csr_to_coo_triplets(csr)
    n_users, n_items = csr['shape']
    hold = (np.arange(n_users), np.zeros(n_users, np.int64), np.ones(n_users))   # every user is a test user
    data = ArrayData((u, i, v), n_users=n_users, n_items=n_items, holdout=hold, warm_start=False)
    idx, val, shp = data.to_coo(tensor_mode=True)
    tu, ti, tf = data.test_to_coo(tensor_mode=True)
    return data, idx, val, shp, (tu, ti, tf), data.get_test_shape(tensor_mode=True)


def reduce_like_reference(u0, u1, u2, core, mlrank):
    """models.py:949-965 with the reference's own round_core."""
    factors = [u0, u1, u2]
    for mode in range(3):
        if factors[mode].shape[1] > mlrank[mode]:
            rot, core = RefCoffee.round_core(core, mode, mlrank[mode])
            factors[mode] = factors[mode].dot(rot)
    return factors[0], factors[1], factors[2], core


def lists(u1, u2, test, tshape):
    recs = orc.coffee_recommendations(u1, u2, test, tshape, TOPK, True)
    gaps = []
    for a in range(0, tshape[0], 512):
        b = min(tshape[0], a + 512)
        sc, sd = orc.coffee_slice_recommendations(u1, u2, test, tshape, a, b)
        orc.downvote_seen_items(sc, sd)
        gaps.append(orc.boundary_gap(sc, TOPK))
    return recs, np.concatenate(gaps)


def main():
    data, idx, val, shp, test, tshape = coffee_inputs()
    t0 = time.time()
    trace = []
    u0, u1, u2, core = orc.hooi(idx, val, shp, MLRANK, num_iters=NUM_ITERS, growth_tol=GROWTH_TOL, seed=SEED, trace=trace)
    print('oracle hooi %s: %d iterations, %.0f s' % (MLRANK, len(trace), time.time() - t0))
    recs, gap = lists(u1, u2, test, tshape)
    r0, r1, r2, rcore = reduce_like_reference(u0, u1, u2, np.ascontiguousarray(core), REDUCED)
    rrecs, rgap = lists(r1, r2, test, tshape)
    rng = np.random.RandomState(3)
    pu, pi = np.sort(rng.choice(shp[0], PROBE, replace=False)), np.sort(rng.choice(shp[1], PROBE, replace=False))
    out = dict(digest=np.array(triplet_digest(idx)), nnz=np.int64(len(val)), shape=np.array(shp, np.int64),
               mlrank=np.array(MLRANK, np.int64), reduced=np.array(REDUCED, np.int64), topk=np.int64(TOPK),
               seed=np.int64(SEED), num_iters=np.int64(NUM_ITERS), growth_tol=np.float64(GROWTH_TOL),
               core_norm_trace=np.array(trace), core_norm=np.float64(np.linalg.norm(core)),
               probe_users=pu, probe_items=pi,
               proj0=u0[pu] @ u0[pu].T, proj1=u1[pi] @ u1[pi].T, proj2=u2 @ u2.T,
               core_sv0=np.linalg.svd(core.reshape(MLRANK[0], -1), compute_uv=False),
               recs=recs.astype(np.int32), tie=(gap <= 0),
               clear=(gap > 1e-9 * np.maximum(1.0, np.abs(gap).max())),
               r_core_norm=np.float64(np.linalg.norm(rcore)),
               r_proj0=r0[pu] @ r0[pu].T, r_proj1=r1[pi] @ r1[pi].T, r_proj2=r2 @ r2.T,
               r_recs=rrecs.astype(np.int32), r_clear=(rgap > 1e-9 * np.maximum(1.0, np.abs(rgap).max())))
    if '--check' in sys.argv:
        ref = np.load(os.path.join(HERE, 'coffee_ml1m.npz'))
        bad = []
        for key in sorted(out):
            a, b = np.asarray(out[key]), ref[key]
            if a.dtype.kind in 'iubUS':          # integers, flags and the digest string: equal or not
                if not np.array_equal(a, b):
                    bad.append(key)
            elif not np.allclose(a, b, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(b).max()))):
                bad.append(key)
        print('coffee_ml1m --check: %d arrays; integer arrays byte-equal, float arrays to 1e-13: %s' % (
            len(out), 'ok' if not bad else 'DIFFER: ' + ', '.join(bad)))
        if bad:
            raise SystemExit(1)
        return
    np.savez_compressed(os.path.join(HERE, 'coffee_ml1m.npz'), **out)
    print('coffee_ml1m: nnz %d, clear rows %d / %d (reduced %d)' % (len(val), int(out['clear'].sum()), len(gap),
                                                                   int(out['r_clear'].sum())))


if __name__ == '__main__':
    main()
