"""The packed ("Q20", csrc/foldq.hip) fold-in against the fp32-image fold-in on a bench workload:
   (1) image accuracy: ||V_j - decode_j|| <= 2^-24 D_j for every row, and how D_j compares with the fp32 image's norm column;
   (2) the product: E' = fold_q20 equals the fp64 SpMM of the decoded rows, ||E' - E|| <= 2^-24 w_u for every user;
   (3) kernel times of both fold-ins (HIP events, 20 launches each);
   (4) the pass with either image: identical lists, refolded users, kernel times, passes per second.
   usage: python tools/probes/foldq_probe.py [ml20m|s1m|ml1m] [rank] [topk]"""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch


def ev_ms(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    topk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    argv, sys.argv = sys.argv, ['bench.py', '--workload', wl]
    import bench
    args = bench.parse()
    sys.argv = argv
    from polara_amd import scoring
    B = bench.Bench(args)
    c = B.generate(wl)
    rank = rank or (50 if wl == 'ml20m' else c['cfg']['rank'])
    topk = topk or (10 if wl == 'ml20m' else c['cfg']['topk'])
    st, _ = B.build(c, rank)
    ops, F, A = B.ops, st['F'], st['A']
    K = F.K
    rec = dict(workload=wl, rank=K, topk=topk, users=A.shape[0], items=A.shape[1], nnz=A.nnz)
    assert F.Q20 is not None, 'no packed image was built'
    # (1) the image
    dec = ops.q20_decode(F.Q20, K)
    err = torch.linalg.vector_norm(dec[:, :K] - F.V, dim=1)
    D = dec[:, K]
    vn = torch.linalg.vector_norm(F.V, dim=1)
    rec['image'] = dict(bytes_per_row=int(F.Q20[0].shape[1]), rows_violating_bound=int((err > D * 2.0 ** -24).sum()),
                        err_over_bound_max=float((err / (D * 2.0 ** -24).clamp_min(1e-300)).max()),
                        D_over_norm_median=float((D / vn.clamp_min(1e-300)).median()),
                        D_over_norm_first_rows=float((D[:1000] / vn[:1000]).mean()),
                        fp32_norm_column_over_norm=float((F.V32x[:, K].double() / vn.clamp_min(1e-300)).median()))
    # (2) the product
    T = A.by_activity()[0] if A.shape[0] >= scoring.ORDER_USERS_MIN else A
    Kx = F.Kx
    Eq = ops.empty(T.shape[0], Kx)
    ops.fold_q20(T, F.Q20, K, Eq)
    Eref = ops.spmm(T, dec.contiguous())
    Eex = ops.spmm(T, F.V)
    scale = Eref[:, :K].abs().max().item()
    rec['product'] = dict(vs_decoded_rows_max_abs=float((Eq[:, :K + 1] - Eref).abs().max()), scale=scale,
                          w_rel_diff_max=float(((Eq[:, K] - Eref[:, K]).abs() / Eref[:, K].clamp_min(1e-300)).max()),
                          users_violating_bound=int((torch.linalg.vector_norm(Eq[:, :K] - Eex, dim=1) > Eq[:, K] * 2.0 ** -24).sum()),
                          err_over_bound_median=float((torch.linalg.vector_norm(Eq[:, :K] - Eex, dim=1) / (Eq[:, K] * 2.0 ** -24).clamp_min(1e-300)).median()),
                          padding_zero=bool((Eq[:, K + 1:] == 0).all()))
    E32 = ops.empty(T.shape[0], Kx)
    ops.spmm(T, F.V32x, out=E32)
    rec['w_ratio_q20_over_fp32_median'] = float((Eq[:, K] / E32[:, K].clamp_min(1e-300)).median())
    # (3) the kernels
    rec['fold_ms'] = dict(q20=ev_ms(lambda: ops.fold_q20(T, F.Q20, K, Eq)), fp32=ev_ms(lambda: ops.spmm(T, F.V32x, out=E32)),
                          fp64=ev_ms(lambda: ops.spmm(T, F.V)))
    # (4) the pass
    out = {}
    for packed in (True, False):
        scoring.PACKED_FOLD_IN = packed
        stats = {}
        recs = scoring.recommend(ops, F, A, topk, True, stats=stats)
        ops.timers = {}
        for _ in range(5):
            scoring.recommend(ops, F, A, topk, True, batches=1)
        torch.cuda.synchronize()
        ms = {k: float(np.sum(bench.events_ms(v))) / 5 for k, v in ops.timers.items()}
        ops.timers = None
        per = ev_ms(lambda: scoring.recommend(ops, F, A, topk, True), n=50, warm=20)
        out[packed] = (recs, dict(refolded_users=stats.get('refolded_users'), flagged_users=stats.get('flagged_users'),
                                  kernel_ms=ms, kernel_sum_ms=sum(ms.values()), serial_ms_per_pass=per))
    scoring.PACKED_FOLD_IN = True
    rec['pass'] = dict(packed=out[True][1], fp32_image=out[False][1], lists_identical=bool((out[True][0] == out[False][0]).all()))
    print(json.dumps(rec))
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/foldq_probe_%s_r%d.json' % (wl, K), 'w') as f:
        json.dump(rec, f, indent=1)


if __name__ == '__main__':
    main()
