"""The coarse C entry points (pk_ctx_* / pk_mat_* / pk_svd_build / pk_score_topk) driven through ctypes
alone in a fresh interpreter — no torch, no polara_amd — as a host in another language would."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_coarse_entry_points_without_torch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'coarse_abi_worker.py')], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and 'COARSE_ABI_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_sharded_coarse_build_two_ranks_through_the_comm_callback():
    """pk_svd_build_sharded (the multi-device form of the coarse API): two ranks, each with its own context and
    its own users, the Gramian-step all-reduce through the host-supplied `pk_comm` callback."""
    from conftest import free_port
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'coarse_abi_sharded_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and 'COARSE_SHARDED_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
