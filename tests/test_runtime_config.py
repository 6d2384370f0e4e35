"""The package's one process-wide side effect — asking the HIP runtime for 8 hardware queues before it starts — is
explicit, recorded and checked: `configure_runtime()` / `runtime_info()` (polara_amd/__init__.py).  CPU part: the
bookkeeping in fresh interpreters.  GPU part (-m gpu): a process that used the device BEFORE importing the package gets a
loud warning from HipOps() instead of silently running the multi-stream paths on four shared queues."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(code, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    e.update(env or {})
    r = subprocess.run([sys.executable, '-W', 'always', '-c', code], capture_output=True, text=True, cwd=ROOT, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def test_import_sets_the_queue_count_once_and_records_it():
    r = run('import os, json, polara_amd; print(json.dumps([polara_amd.runtime_info(), os.environ.get("GPU_MAX_HW_QUEUES")]))')
    info, env = json.loads(r.stdout.strip().splitlines()[-1])
    assert info == {'hw_queues': 8, 'source': 'polara_amd.configure_runtime', 'in_time': True, 'ok': True} and env == '8'


def test_a_value_the_user_exported_wins_and_a_small_one_is_reported_as_not_ok():
    r = run('import os, json, polara_amd; print(json.dumps([polara_amd.runtime_info(), os.environ.get("GPU_MAX_HW_QUEUES")]))',
            env={'GPU_MAX_HW_QUEUES': '2'})
    info, env = json.loads(r.stdout.strip().splitlines()[-1])
    assert info['hw_queues'] == 2 and info['source'] == 'environment' and info['ok'] is False and env == '2'
    r = run('import json, polara_amd; print(json.dumps(polara_amd.configure_runtime(16)))', env={'GPU_MAX_HW_QUEUES': '12'})
    assert json.loads(r.stdout.strip().splitlines()[-1])['hw_queues'] == 12


def test_freeze_imports_is_explicit_and_moves_the_heap_out_of_the_collectors_sight():
    """The package never freezes the collector's generations by itself (a process-global setting); `freeze_imports()` does,
    after importing the host modules, and says how much it froze (bench.py reports it in `cold.gc_frozen_objects`)."""
    r = run('import gc, json, polara_amd\n'
            'before = gc.get_freeze_count()\n'
            'n = polara_amd.freeze_imports()\n'
            'import sys\n'
            'print(json.dumps([before, n, gc.get_freeze_count(), "polara_amd.solver" in sys.modules, "polara_amd.scoring" in sys.modules]))')
    before, n, after, solver_in, scoring_in = json.loads(r.stdout.strip().splitlines()[-1])
    assert before == 0 and n == after and n > 50000 and solver_in and scoring_in


@pytest.mark.gpu
def test_device_used_before_the_import_is_detected_and_hipops_warns():
    code = ('import torch, json, warnings\n'
            'torch.zeros(1, device="cuda:0")\n'
            'import polara_amd\n'
            'from polara_amd.ops import HipOps\n'
            'with warnings.catch_warnings(record=True) as w:\n'
            '    warnings.simplefilter("always")\n'
            '    ops = HipOps(warm=False)\n'
            'print(json.dumps([polara_amd.runtime_info(), [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]]))\n')
    info, msgs = json.loads(run(code).stdout.strip().splitlines()[-1])
    assert info['in_time'] is False and info['ok'] is False and info['hw_queues'] == 4
    assert any('hardware queue' in m and 'configure_runtime' in m for m in msgs), msgs


@pytest.mark.gpu
def test_import_before_the_first_device_use_gets_the_fast_configuration_without_a_warning():
    code = ('import polara_amd, json, warnings\n'
            'import torch\n'
            'from polara_amd.ops import HipOps\n'
            'with warnings.catch_warnings(record=True) as w:\n'
            '    warnings.simplefilter("always")\n'
            '    ops = HipOps(warm=False)\n'
            'print(json.dumps([polara_amd.runtime_info(), [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]]))\n')
    info, msgs = json.loads(run(code).stdout.strip().splitlines()[-1])
    assert info['ok'] and info['in_time'] and not any('hardware queue' in m for m in msgs)
