"""What does the candidate sweep cost WITHOUT its selection work?  Times pk_score_candidates_f32 alone (headline workload)
as it is, and with PK_SCORE_ABLATE=2 (no pushes: the thresholds stay at the bootstrap's lower bounds, so groups leave later
— the swept fraction is printed next to the time), =1 (no seen masks), =3 (neither).  The lists of the ablated runs are
wrong; only the time per tile-wave is of interest.  usage: python tools/probes/sweep_floor.py [ml20m|s1m]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as B


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    sys.argv = sys.argv[:1]
    bench = B.Bench(B.parse())
    ops = bench.ops
    c = bench.generate(wl)
    st, _ = bench.build(c, 50, True)
    from polara_amd import scoring
    F, A = st['F'], st['A']
    T, perm = A.by_activity()
    n_users, n_items = T.shape
    E = ops.spmm(T, F.V)
    Ep, ub = ops.pack_frag_bound(E)
    KC = 16
    args = dict(user_bound=ub, tile_bound=F.tile_bound, seen_tiles=T.seen_tiles(), seen_dense=T.seen_dense())
    os.environ['PK_SCORE_HEAD_TILES'] = '0'
    combos = [(b, a) for b in ('16', '0') for a in ('0', '2', '1', '3')]
    if os.environ.get('PK_FLOOR_DIAG'):      # a PK_SCORE_DIAG build: the FULL sweep (4) without pushes (2), then without re-loads (16) / products (32) / both
        combos = [('0', a) for a in ('6', '22', '38', '54', '7', '23')]
    shared = [None] * len(combos)
    if os.environ.get('PK_FLOOR_DIAG') == '2':   # the LDS-staged instance: as it is, and with its per-tile wait + barrier removed (scores wrong)
        combos = [('0', '6'), ('0', '6'), ('0', '70'), ('0', '22')]
        shared = ['0', '1', '1', '0']
    if os.environ.get('PK_FLOOR_DIAG') == '3':   # the same with the pushes ON (full sweep, bootstrap on): the regime of the unpruned bench lines
        combos = [('16', '4'), ('16', '4'), ('16', '68')]
        shared = ['0', '1', '1']
    if os.environ.get('PK_FLOOR_DIAG') == '4':   # default build: register-fed against LDS-staged, full sweep with and without pushes
        combos = [('16', '4'), ('16', '4'), ('0', '6'), ('0', '6')]
        shared = ['0', '1', '0', '1']
    for (boot, abl), sh in zip(combos, shared):
        if sh is not None:
            os.environ['PK_SCORE_SHARED'] = sh
        if True:
            os.environ['PK_SCORE_BOOT_TILES'] = boot
            os.environ['PK_SCORE_ABLATE'] = abl
            for _ in range(3):
                ops.score_candidates(F.Vp, Ep, n_users, n_items, 50, T.indptr, T.indices, KC, **args)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record()
                ops.score_candidates(F.Vp, Ep, n_users, n_items, 50, T.indptr, T.indices, KC, **args)
                b.record()
            torch.cuda.synchronize()
            ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            ex = ops.score_exit_tiles(n_users, 1)
            swept = float(ex.clamp_min(0).sum().item()) / (ex.shape[1] * (-(-n_items // 32)))
            tw = float(ex.clamp_min(0).sum().item())
            print(json.dumps(dict(shared=sh, boot=int(boot), ablate=int(abl), sweep_ms=round(ms, 4), swept=round(swept, 4), tile_waves=int(tw),
                                  ns_per_tile_wave_x1024simd=round(ms * 1e6 * 1024 / tw, 1))), flush=True)


if __name__ == '__main__':
    main()
