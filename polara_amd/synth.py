"""Synthetic user-item interaction generators with *planted* low-rank structure.

Why planted: a uniform-random CSR has a flat Marchenko-Pastur bulk, which makes the top-k singular
subspace ill-conditioned for ANY solver (the reference's ARPACK included).  The latent model is

    logit[u, i] = p_u . q_i + b_i ,   p_u, q_i in R^{r*},  r* = 2*rank,  factor scales ~ j^(-1/2),
    b_i = 0.8 * log(1 / popularity_rank_i)                  (Zipf-like item popularity),
    n_u ~ logNormal clipped to [min_items, max_items]        (user activity),
    items(u) = top-n_u of (logit[u, :] + Gumbel noise)       (= sampling w/o replacement ~ softmax),
    rating   = quantile bin of the logit into `levels` levels (1..levels).

Written in torch so the same code generates the small CPU test matrices and, on the GPU box, the
1e8-nnz benchmark matrix in seconds (data generation is plumbing, not the product path).
Output is canonical CSR (row-sorted, column-sorted, no duplicates).
"""
import math
import numpy as np
import torch


def planted_csr(n_users, n_items, mean_items, rank, levels=5, seed=0, device='cpu',
                min_items=20, max_items=None, sigma_activity=1.0, chunk_rows=4096,
                noise=1.0, return_factors=False):
    """Returns dict(indptr int64[n_users+1], indices int32[nnz], values float32[nnz], shape)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    r_star = 2 * rank
    if max_items is None:
        max_items = max(min_items + 1, int(0.05 * n_items))
    max_items = min(max_items, n_items)
    min_items = min(min_items, max_items)

    scale = torch.arange(1, r_star + 1, device=dev, dtype=torch.float32).pow(-0.5)
    P = torch.randn(n_users, r_star, generator=g, device=dev) * scale * 1.5
    Q = torch.randn(n_items, r_star, generator=g, device=dev) * scale * 1.5
    pop_rank = torch.randperm(n_items, generator=g, device=dev).to(torch.float32) + 1.0
    bias = 0.8 * torch.log(1.0 / pop_rank)

    mu = math.log(max(mean_items, 1.0)) - 0.5 * sigma_activity ** 2
    n_u = torch.exp(mu + sigma_activity * torch.randn(n_users, generator=g, device=dev))
    n_u = n_u.clamp(min_items, max_items).round().to(torch.int64)

    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(n_u, 0)
    nnz = int(indptr[-1].item())
    indices = torch.empty(nnz, dtype=torch.int32, device=dev)
    values = torch.empty(nnz, dtype=torch.float32, device=dev)

    # process users in order of activity so that every chunk has a tight top-k width
    order = torch.argsort(n_u)
    edges = None
    for c0 in range(0, n_users, chunk_rows):
        users = order[c0:c0 + chunk_rows]
        cnt = n_u[users]
        kmax = int(cnt.max().item())
        logits = P[users] @ Q.T + bias
        u01 = torch.rand(logits.shape, generator=g, device=dev).clamp_(1e-12, 1.0 - 1e-7)
        gumbel = -torch.log(-torch.log(u01)) * noise
        _, top_items = torch.topk(logits + gumbel, kmax, dim=1)
        valid = torch.arange(kmax, device=dev)[None, :] < cnt[:, None]
        # sort the selected items of each row by item id (canonical CSR)
        keyed = torch.where(valid, top_items, torch.full_like(top_items, n_items))
        keyed, _ = torch.sort(keyed, dim=1)
        sel_logits = torch.gather(logits, 1, keyed.clamp(max=n_items - 1))
        if edges is None:
            # rating thresholds: quantiles of the selected logits of the first chunk(s) seen
            sample = sel_logits[valid]
            if sample.numel() > 200000:
                sample = sample[torch.randperm(sample.numel(), generator=g, device=dev)[:200000]]
            qs = torch.linspace(0, 1, levels + 1, device=dev)[1:-1]
            edges = torch.quantile(sample.float(), qs)
        rating = (torch.bucketize(sel_logits, edges) + 1).to(torch.float32)
        dest = indptr[users][:, None] + torch.arange(kmax, device=dev)[None, :]
        indices[dest[valid]] = keyed[valid].to(torch.int32)
        values[dest[valid]] = rating[valid]
        del logits, u01, gumbel, top_items, keyed, sel_logits, rating, dest

    out = dict(indptr=indptr, indices=indices, values=values, shape=(n_users, n_items))
    if return_factors:
        out['P'] = P
        out['Q'] = Q
    return out


def csr_to_numpy(csr):
    return dict(indptr=csr['indptr'].cpu().numpy(), indices=csr['indices'].cpu().numpy(),
                values=csr['values'].cpu().numpy(), shape=tuple(csr['shape']))


def csr_to_coo_triplets(csr):
    """(user_idx int64, item_idx int64, val float64) sorted by user then item."""
    c = csr_to_numpy(csr)
    counts = np.diff(c['indptr'])
    users = np.repeat(np.arange(c['shape'][0], dtype=np.int64), counts)
    return users, c['indices'].astype(np.int64), c['values'].astype(np.float64)


# Named workload shapes (BASELINE.json configs / SURVEY.md §8d).  `mean_items` ~ nnz / n_users.
WORKLOADS = {
    'ml1m':  dict(n_users=6040, n_items=3706, mean_items=165, rank=10, levels=5, seed=1,
                  min_items=20, max_items=2200, topk=10),
    's1m':   dict(n_users=1_000_000, n_items=100_000, mean_items=100, rank=50, levels=5, seed=2,
                  min_items=20, max_items=5000, topk=10),
    'ml20m': dict(n_users=138_493, n_items=26_744, mean_items=144, rank=100, levels=10, seed=3,
                  min_items=20, max_items=9000, topk=20),
}


def make_workload(name, device='cpu', scale=1.0, **override):
    """Generate a named workload; `scale` < 1 shrinks users and items proportionally (tests)."""
    cfg = dict(WORKLOADS[name])
    cfg.update(override)
    topk = cfg.pop('topk')
    if scale != 1.0:
        cfg['n_users'] = max(64, int(cfg['n_users'] * scale))
        cfg['n_items'] = max(64, int(cfg['n_items'] * scale))
        cfg['max_items'] = max(cfg['min_items'] + 1, min(cfg['max_items'], cfg['n_items'] // 2))
        cfg['mean_items'] = min(cfg['mean_items'], cfg['n_items'] // 4)
    csr = planted_csr(device=device, **cfg)
    return csr, dict(rank=cfg['rank'], topk=topk, levels=cfg['levels'])
