"""C-ABI surface: the shared library loads and exports every symbol include/polara_hip.h declares,
with prototypes registered in polara_amd/_lib.py (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'polara_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from polara_amd import _lib
    from polara_amd.build_native import build
    build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'missing export: ' + n


def test_python_prototypes_cover_the_header():
    from polara_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_functions()
    lib = _lib.load()
    assert lib.pk_version() >= 100
    # pure host-side helpers are callable without a device
    assert lib.pk_pack_kq(50) == 8 and lib.pk_pack_kq(100) == 14 and lib.pk_pack_kq(10) == 2 and lib.pk_pack_kq(200) == 26
    assert lib.pk_candidate_capacity(10) == 16 and lib.pk_candidate_capacity(20) == 32
    assert lib.pk_candidate_capacity(50) == 64 and lib.pk_candidate_capacity(100) == 0
    assert lib.pk_pack_elems(33, 50) == 2 * 8 * 64 * 4
    assert lib.pk_gram_work_bytes(1000, 64, 64) > 0
    # launches of a candidate sweep (round 5): a pruned sweep is ONE launch whatever the catalogue; a full sweep is cut into
    # L2-sized item chunks; an explicit chunk length is honoured (doubling from launch to launch when pruned)
    assert lib.pk_score_chunk_launches(26744, 50, 1, 0, 1) == 1 and lib.pk_score_chunk_launches(500000, 200, 1, 0, 1) == 1
    assert lib.pk_score_chunk_launches(100000, 50, 1, 0, 0) == -(-3125 // 320)
    assert lib.pk_score_chunk_launches(26744, 50, 1, 100, 0) == 9 and lib.pk_score_chunk_launches(26744, 50, 1, 100, 1) == 4
    assert lib.pk_score_splits(138493, 16) == 1 and lib.pk_score_splits(2000, 16) == 4


def test_no_cpu_fallback_in_package():
    """The product package never imports the oracle nor SciPy's solvers (models.get_training_matrix
    builds a SciPy container for API compatibility only)."""
    pkg = os.path.join(ROOT, 'polara_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn
            assert not re.search(r'^\s*(from|import)\s+scipy\.sparse\.linalg', src, flags=re.M), fn
            assert 'import svds' not in src, fn
