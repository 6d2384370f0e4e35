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
