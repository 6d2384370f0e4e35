"""-m gpu: bench.py end to end as the driver runs it (1 GPU, default blocks incl. the CPU baseline and the adversarial
catalogues), on a scaled-down matrix (the sub-blocks are full-size only): the LAST stdout line must be the compact JSON
record — parseable, under 3 KB, with `roofline` and `cpu_baseline`, the lists of the sample identical to the CPU path's."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_default_bench_prints_one_compact_parseable_line(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--scale', '0.05'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last.encode()) <= 3000
    d = json.loads(last)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['value'] > 0 and d['higher_is_better'] is True
    assert d['config']['workload'] and d['config']['scale'] == 0.05 and 'model' not in d['config']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(d['roofline']) and d['roofline']['bound'] in ('hbm', 'mfma')
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-3
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(d['cpu_baseline']) and d['cpu_baseline']['kind'] in ('port', 'reference')
    assert d['cpu_baseline']['identical_rows'] == 1.0                      # the GPU lists against the CPU path's on the sample
    assert 'adversarial_users_per_s' not in d['config']     # those three catalogues only ride along at full size (scale 1.0)
    # a pass over 7 K users is shorter on the device than on the host: the loop must have calibrated the replayed form
    # (scoring.RecordedPass) next to the launched ones, and whichever ran, the lists above came out of it
    detail = json.load(open(os.path.join(ROOT, 'bench_detail.json')))
    cal = detail['warmup_calibration_ms_per_step']
    assert cal['python_launch'] > 0 and cal.get('recorded_replay') is not None and cal['recorded_replay'] > 0, cal
    assert d['config']['launch'] in ('python', 'hipGraph', 'python, passes on 2 streams') or d['config']['launch'].startswith('recorded calls'), d['config']['launch']
