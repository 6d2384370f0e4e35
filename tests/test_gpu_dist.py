"""-m gpu: the N > 1 path on the product backend.  Two ranks on the one visible GPU (all-reduce staged through gloo), the
bench under torchrun the same way, and — where the box has at least two GPUs — the real thing: one rank per GPU over
RCCL ("nccl").  The GPU test box has ONE device: the RCCL test then SKIPS, loudly, rather than pretending."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu


def test_sharded_path_over_rccl_when_two_gpus_are_visible(capsys):
    """FIRST in this file: wherever two GPUs are visible the real thing runs before anything else — one rank
    per GPU over RCCL, the sharded build + scoring against the single-GPU lists (dist_worker_nccl.py), then bench.py as
    the driver launches it.  The device count and RCCL's version banner (NCCL_DEBUG=VERSION) are printed either way."""
    import torch
    n = torch.cuda.device_count()
    with capsys.disabled():
        print('\n[dist] visible GPUs: %d (%s)' % (n, ', '.join(torch.cuda.get_device_name(i) for i in range(n))))
    if n < 2:
        pytest.skip('RCCL path NOT exercised: %d GPU visible, the nccl backend needs one process per GPU (>= 2)' % n)
    r = _torchrun(os.path.join(ROOT, 'tests', 'dist_worker_nccl.py'), 2, extra_env={'NCCL_DEBUG': 'VERSION'})
    with capsys.disabled():
        print('[dist] ' + '\n[dist] '.join(l for l in (r.stdout + r.stderr).splitlines() if 'NCCL' in l or 'RCCL' in l or 'DIST_NCCL' in l))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_NCCL_RESULT' in r.stdout
    r = _torchrun(os.path.join(ROOT, 'bench.py'), 2, args=['--gpus', '2', '--steps', '3', '--warmup', '1', '--scale', '0.1'])
    assert r.returncode == 0 and '"n_gpus":2' in r.stdout.replace(' ', ''), r.stdout[-2000:] + r.stderr[-3000:]


def test_two_rank_sharded_path_on_hip_backend():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'dist_worker_gpu.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_GPU_RESULT' in r.stdout


def _torchrun(script, n, extra_env=None, args=(), timeout=900):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script] + list(args)
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_under_torchrun_two_ranks_one_gpu():
    """bench.py as the driver launches it for N > 1 (torchrun, one JSON line from rank 0, MAX over ranks), with the
    two ranks sharing the one GPU and the collectives staged through gloo (PK_BENCH_DEBUG_BACKEND): the path check
    that can run on a one-GPU box."""
    import json
    r = _torchrun(os.path.join(ROOT, 'bench.py'), 2, extra_env={'PK_BENCH_DEBUG_BACKEND': 'gloo'},
                  args=['--gpus', '2', '--steps', '3', '--warmup', '1', '--scale', '0.1', '--no-cpu-baseline'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['scaling'] == 'strong' and d['build']['converged']


def test_rccl_collectives_on_one_rank():
    """The RCCL calls of polara_amd/dist.py issued for real on the one GPU this box has: backend "nccl", a group of one
    rank, TorchComm(exercise_collectives=True) — all-reduce (fp64 blocks, int64 counts), all_gather_into_tensor,
    reduce_scatter_tensor, the typed result gather, barrier — and the item-sharded solver through them, bit-equal to the
    communicator-free run (tests/dist_worker_rccl_one_rank.py).  What it cannot show is more than one rank: that stays
    with the two-GPU test above."""
    r = _torchrun(os.path.join(ROOT, 'tests', 'dist_worker_rccl_one_rank.py'), 1, extra_env={'NCCL_DEBUG': 'VERSION'})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'RCCL_ONE_RANK_RESULT' in r.stdout


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher.  Under the debug backend (two ranks sharing the one
    GPU) it re-executes itself under torch.distributed.run and the line says n_gpus = 2 with the per-rank shards and the
    build's collective counts; without the debug backend on a box with fewer than two GPUs it exits non-zero with a
    message — it never again measures one GPU under an N-GPU flag."""
    import json
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PK_BENCH_DEBUG_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--scale', '0.1',
                        '--no-cpu-baseline'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['dist']['world'] == 2 and d['dist']['backend'] == 'gloo'
    assert len(d['dist']['users_per_rank']) == 2 and sum(d['dist']['users_per_rank']) == d['config']['n_users']
    bc = d['dist']['build_collectives']
    assert bc['reduce_scatter'] == d['build']['gramian_steps'] and bc['all_gather'] >= bc['reduce_scatter'] and bc['MB'] > 0
    assert d['dist']['scoring_collectives'] == 0
    if torch.cuda.device_count() < 2:
        env.pop('PK_BENCH_DEBUG_BACKEND')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--scale', '0.1'],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and 'needs 2 visible GPUs' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]
