"""Builds polara_amd/libpolarahip.so from polara_amd/csrc with hipcc for gfx950 (in-tree).

hipcc cross-compiles without a GPU, so this runs in the build container and the resulting .so
travels to the GPU box with the repo snapshot.  Object files are rebuilt only when their source
(or a header) is newer.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, 'libpolarahip.so')
OBJDIR = os.path.join(HERE, 'csrc', '_obj')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-pass-failed', '-I', os.path.join(ROOT, 'include'),
         # MFMA accumulators in arch VGPRs (gfx950 has a unified register file): the scoring
         # epilogue reads them with v_max3 directly instead of 16 v_accvgpr_read per tile
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']
# (kernel-tuning builds — the rejected sweep / fold-in variants of csrc/experiments/ with their -D switches — are made by
# tools/build_probe_lib.py into a library of their own, selected with POLARA_HIP_LIB: never into libpolarahip.so)

def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _digest(paths, extra=''):
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(ROOT, 'include', 'polara_hip.h'))
    return hs


def _compile(src, force=False, extra=()):
    """Rebuilds an object only when the CONTENT of its source, the headers or the flags changed
    (content hashes, not mtimes: the snapshot that travels to the GPU box does not keep mtimes)."""
    obj = os.path.join(OBJDIR, src + '.o')
    spath = os.path.join(CSRC, src)
    # (the repo's own location is not part of the stamp: the snapshot on the GPU box lives under another path)
    want = _digest([spath] + _headers(), (' '.join(FLAGS) + ' '.join(extra)).replace(ROOT, '<repo>'))
    stamp = obj + '.sha1'
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ['-x', 'hip', '-c', spath, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, 'w') as f:
        f.write(want)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('libpolarahip.so: %s (%d bytes)' % ('rebuilt' if rebuilt else 'up to date', os.path.getsize(LIB)))
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
