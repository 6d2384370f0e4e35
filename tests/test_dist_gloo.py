"""N>1 path on CPU: two processes, gloo backend, users row-sharded, Gramian all-reduce."""
import os
import subprocess
import sys

from conftest import ROOT, free_port


def test_two_rank_sharded_build_and_scoring():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'dist_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_RESULT' in r.stdout


import pytest


@pytest.mark.parametrize('world,method', [(2, 'subspace'), (3, 'subspace'), (2, 'lanczos'), (3, 'lanczos')])
def test_row_sharded_item_side_of_the_solver(world, method):
    """solver.ItemRows: all-gather X / reduce-scatter Z / all-reduced Gram matrices give the factors of the whole
    matrix, at item counts that are not multiples of the world size, and through the rank-deficient refill path —
    for the filtered subspace iteration and for the block Lanczos method (whose Krylov basis is row-sharded the same
    way; the rank-deficient and the 61-item cases exercise its hand-over to the subspace iteration)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='2', PK_TEST_SVD_METHOD=method)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker_solver.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'SOLVER_DIST_RESULT' in r.stdout


def test_eight_rank_job_item_count_not_a_multiple_of_eight_and_ranks_without_test_users():
    """The shape of the driver's 8-GPU job on CPU: 203 items over 8 ranks (padding rows on the last
    rank), 5 test users (most ranks score nobody and still join the result gather), factors and lists equal to the
    single-process model on every rank, one reduce-scatter per Gramian step, no reduction in scoring."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker_world8.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'WORLD8_RESULT' in r.stdout
