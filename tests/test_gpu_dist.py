"""-m gpu: two ranks on the one visible GPU, real HIP kernels, all-reduce staged through gloo."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu


def test_two_rank_sharded_path_on_hip_backend():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'dist_worker_gpu.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_GPU_RESULT' in r.stdout
