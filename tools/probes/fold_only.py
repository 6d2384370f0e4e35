"""Only the fold-in kernels of a bench workload, for counter passes and variant timing: packed (Q20) fold-in, fp32-image
   fold-in, and the fp64 narrow product at 16 columns (128-byte rows through the plain SpMM kernel) as the memory-side yardstick.
   usage: python tools/probes/fold_only.py [ml20m|s1m] [rank] [reps]      (PK_FOLDQ_U=2|4|8 selects the L = 8 instance)"""
import json
import os
import sys

sys.path.insert(0, '.')
import torch


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'ml20m'
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    argv, sys.argv = sys.argv, ['bench.py', '--workload', wl]
    import bench
    args = bench.parse()
    sys.argv = argv
    from polara_amd import scoring
    from tools.probes.foldq_probe import ev_ms
    B = bench.Bench(args)
    c = B.generate(wl)
    ops = B.ops
    # the serving-order test matrix and a factor image of the right shape without a build: random rows with a norm decay
    A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
    rank_of, inv, counts, rank_dev = ops.item_order(A)
    A = ops.csr_relabel_cols(A, rank_dev)
    T = A.by_activity()[0] if A.shape[0] >= scoring.ORDER_USERS_MIN else A
    n_items = c['shape'][1]
    g = torch.Generator(device=ops.device)
    g.manual_seed(1)
    V = torch.randn(n_items, rank, generator=g, dtype=torch.float64, device=ops.device) * \
        (torch.arange(1, n_items + 1, device=ops.device, dtype=torch.float64) ** -0.4)[:, None]
    F = scoring.FactorImage(ops, V)
    Kx = F.Kx
    Eq = ops.empty(T.shape[0], Kx)
    E32 = ops.empty(T.shape[0], Kx)
    X16 = torch.randn(n_items, 16, generator=g, dtype=torch.float64, device=ops.device)
    rec = dict(workload=wl, rank=rank, U=os.environ.get('PK_FOLDQ_U', '4'), nnz=T.nnz, rows=T.shape[0])
    rec['q20_ms'] = ev_ms(lambda: ops.fold_q20(T, F.Q20, rank, Eq), n=reps)
    rec['fp32_ms'] = ev_ms(lambda: ops.spmm(T, F.V32x, out=E32), n=reps)
    rec['fp64_nc16_ms'] = ev_ms(lambda: ops.spmm(T, X16), n=reps)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
