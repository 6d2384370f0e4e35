"""Builds a KERNEL-TUNING library next to the product one: libpolarahip.so's objects with score.hip / spmm.hip replaced by
their experiment trees (polara_amd/csrc/experiments/: the LDS-staged sweeps, the two-groups-per-wave kernel, the
two-buffer / depth-2 / two-chain tile loops, the cycle-counter and ablation switches, the LDS-head fold-in — built,
verified and measured in rounds 3-4, records in profiles/ and DESIGN.md).  The product library never contains them.

    python tools/build_probe_lib.py out.so [--trees score,spmm] [-DPK_SCORE_PROFILE -DPK_FAST_BUILD ...]
    POLARA_HIP_LIB=out.so python tools/probes/sweep_profile.py ...

The experiment trees export the product ABI of the round they were frozen in (round 4) plus their run-time switches
(PK_SCORE_SHARED, PK_SCORE_PAIR, PK_FOLD_HEAD, PK_SCORE_ABLATE); entry points added later (the packed fold-in) come from the
product objects.  --trees picks the experiment trees to swap in (default: score only — the product's spmm.hip has grown
entry points since, the flagged and the list-driven products, that driver.hip needs at link time; `--trees score,spmm`
links only against a driver that does not call them, i.e. a checkout of round 4's serving path)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polara_amd import build_native as bn      # noqa: E402


def main():
    out = os.path.abspath(sys.argv[1])
    defs = [a for a in sys.argv[2:] if a.startswith('-D')]
    trees = 'score'
    if '--trees' in sys.argv:
        trees = sys.argv[sys.argv.index('--trees') + 1]
    trees = set(trees.split(','))
    bn.build(verbose=False)
    exp = os.path.join(bn.CSRC, 'experiments')
    objs = []
    for src in bn.sources():
        alt = os.path.join(exp, src.replace('.hip', '_variants.hip'))
        if os.path.exists(alt) and src.replace('.hip', '') in trees:
            obj = os.path.join(bn.OBJDIR, 'probe_' + src + '.o')
            cmd = [bn.HIPCC] + bn.FLAGS + defs + ['-I', bn.CSRC, '-x', 'hip', '-c', alt, '-o', obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit('hipcc failed for %s:\n%s' % (alt, r.stderr))
            objs.append(obj)
        else:
            objs.append(os.path.join(bn.OBJDIR, src + '.o'))
    r = subprocess.run([bn.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit('link failed:\n' + r.stderr)
    print('%s (%d bytes; experiment trees swapped in: %s)' % (out, os.path.getsize(out), ', '.join(sorted(trees))))


if __name__ == '__main__':
    main()
