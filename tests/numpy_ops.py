"""TEST-ONLY NumPy/SciPy double of polara_amd.ops.HipOps (same method names and semantics on CPU
torch tensors).  It exists so the solver / scoring / distributed ORCHESTRATION code — which is
backend-agnostic — can be exercised in the GPU-less container (incl. world_size-2 gloo runs).
It is never imported by the package; on a GPU the product path is HipOps and nothing else.
"""
import numpy as np
import scipy.sparse as sps
import torch

from polara_amd.csr import csr_transpose


class NpCSR:
    def __init__(self, indptr, indices, values, shape):
        self.shape = (int(shape[0]), int(shape[1]))
        self.m = sps.csr_matrix((np.asarray(values, dtype=np.float64), indices, indptr), shape=self.shape)
        self.nnz = self.m.nnz
        self.indptr = torch.from_numpy(np.asarray(indptr, dtype=np.int64))
        self.indices = torch.from_numpy(np.asarray(indices, dtype=np.int32))
        self._T = None

    def seen_tiles(self):
        return None

    def nonneg(self):
        return bool((self.m.data >= 0).all())

    @property
    def T(self):
        if self._T is None:
            t = self.m.T.tocsr()
            t.sort_indices()
            self._T = NpCSR(t.indptr, t.indices, t.data, t.shape)
            self._T._T = self
        return self._T


class NumpyOps:
    name = 'numpy-test-double'
    device = torch.device('cpu')

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a)
        import warnings
        with warnings.catch_warnings():
            # ArrayData stores read-only views of the caller's columns; torch warns that a tensor over one must not be
            # written to — it is only the source of the copy below
            warnings.simplefilter('ignore', UserWarning)
            t = torch.as_tensor(a)
        return t.to(dtype) if dtype is not None else t

    def to_host(self, t):
        return t.detach().numpy()

    def csr(self, indptr, indices, values, shape, split=None):
        return NpCSR(indptr, indices, values, shape)

    def csr_from_coo(self, rows, cols, vals, shape, split=None):
        from polara_amd.csr import coo_to_csr
        rows, cols = (np.asarray(x) for x in (rows, cols))          # host arrays or CPU tensors
        ip, ix, vl = coo_to_csr(rows, cols, np.asarray(vals, dtype=np.float64), shape)
        return NpCSR(ip, ix, vl, shape)

    def csr_relabel_cols(self, A, col_map, sort=True):
        m = A.m.tocoo()
        from polara_amd.csr import coo_to_csr
        ip, ix, vl = coo_to_csr(m.row, np.asarray(col_map)[m.col], m.data, A.shape, sum_duplicates=False)
        return NpCSR(ip, ix, vl, A.shape)

    def bincount(self, keys, n_bins):
        return torch.bincount(keys, minlength=int(n_bins))[:int(n_bins)]

    def item_counts(self, A):
        return np.bincount(A.m.indices, minlength=A.shape[1]).astype(np.int64)

    def item_order(self, A, comm=None):
        from polara_amd.csr import popularity_order
        counts = self.item_counts(A)
        if comm is not None and getattr(comm, 'world', 1) > 1:
            counts = self.to_host(comm.allreduce(self.to_device(counts)))
        rank, inv = popularity_order(None, A.shape[1], counts=counts)
        return rank, inv, counts, torch.from_numpy(rank)

    def csr_rows(self, A, lo, hi):
        sub = A.m[lo:hi]
        return NpCSR(sub.indptr, sub.indices, sub.data, sub.shape)

    def randn(self, n, m, seed):
        g = torch.Generator(device='cpu')
        g.manual_seed(int(seed))
        return torch.randn(n, m, generator=g, dtype=torch.float64)

    def spmm(self, A, X, out=None, rows=None):
        if hasattr(A, 'apply'):
            return A.apply(X, out)
        r = torch.from_numpy(np.ascontiguousarray(A.m @ X.numpy().astype(np.float64)))
        if out is not None:
            nc = r.shape[1]
            if rows is not None:
                out[rows[0]:rows[1], :nc].copy_(r[rows[0]:rows[1]])
            else:
                out[:, :nc].copy_(r)
            return out
        return r

    def gram(self, A, B=None):
        B = A if B is None else B
        return A.t() @ B

    def tsmm(self, X, Cm, out=None):
        if out is not None:
            out.copy_(X @ Cm)
            return out
        return (X @ Cm).contiguous()

    def eigh_psd(self, S, max_sweeps=0, tol=0.0):
        w, v = np.linalg.eigh(S.numpy())
        w = np.maximum(w[::-1], 0.0).copy()
        v = v[:, ::-1].copy()
        return torch.from_numpy(w), torch.from_numpy(v)

    def tsmm_sub(self, Z, X, Cm, out=None):
        r = Z - X @ Cm
        if out is not None:
            out.copy_(r)
            return out
        return r

    def sym_eig_topk(self, T, k, X0=None, tol=1e-13, max_outer=200, seed=0, stats=None, lam0=None, r0_rel=-1.0, width=None):
        # (lam0 / r0_rel: what the product's segmented solve starts a warm look from; the double runs the general iteration)
        # the product path runs this loop in C++ (pk_sym_eig_topk_f64); the double runs the Python statement of it
        from polara_amd.solver import _subspace_iteration, _Dense, ItemRows, NoComm
        n = T.shape[0]
        if X0 is None:
            l = min(n, max(int(k), 8, int(width or 0)))
            X = torch.zeros(n, l, dtype=torch.float64)
            X[:l] = torch.eye(l, dtype=torch.float64)
        else:
            X = torch.zeros(n, X0.shape[1], dtype=torch.float64)
            X[:X0.shape[0]] = X0
        st = stats if stats is not None else {}
        st.setdefault('steps', 0)
        return _subspace_iteration(_Dense(self, T.contiguous(), st), ItemRows(self, NoComm(), n), k, X, tol, max_outer, 24, 1e7,
                                   seed, st, even_lock=False)

    def eigh_top(self, S, r):
        lam, C = self.eigh_psd(S)
        return lam[:r], C[:, :r]

    def chol_rinv(self, G, shift_rel=0.0, info=None):
        g = G.numpy()
        n = g.shape[0]
        out_info = torch.zeros(1, dtype=torch.int32) if info is None else info
        try:
            L = np.linalg.cholesky(g + shift_rel * np.trace(g) * np.eye(n))
            Rinv = np.linalg.inv(L.T)
            out_info[0] = 0
        except np.linalg.LinAlgError:
            Rinv = np.zeros((n, n))
            out_info[0] = 1
        return torch.from_numpy(np.ascontiguousarray(np.triu(Rinv))), out_info

    def axpbypcz(self, alpha, Z, beta=0.0, Y=None, gamma=0.0, X=None, out=None):
        r = alpha * Z
        if Y is not None:
            r = r + beta * Y
        if X is not None:
            r = r + gamma * X
        return r

    def resid_colnorm2(self, Z, X, theta):
        return ((Z - X * theta[None, :]) ** 2).sum(dim=0)

    def small_mm(self, A, B, transA=False, transB=False):
        return ((A.t() if transA else A) @ (B.t() if transB else B)).contiguous()

    def scale_cols(self, X, s):
        X.mul_(s[None, :])
        return X

    def tucker_predict(self, users, items, u, v, w, core, want_scores=False):
        users, items = (np.asarray(a, dtype=np.int64) for a in (users, items))
        u, v, w, core = (np.asarray(a, dtype=np.float64) for a in (u, v, w, core))
        t = np.einsum('abc,hb->hac', core, v[items])
        gu = np.einsum('hac,ha->hc', t, u[users])
        scores = gu @ w.T
        return torch.from_numpy(np.argmax(scores, axis=1).astype(np.int64)), (torch.from_numpy(scores) if want_scores else None)

    def synchronize(self):
        pass

    # ---- K5 double ------------------------------------------------------------------------------
    def ttm(self, plan, idx1, idx2, vals, u, v, n0):
        # plan carries the row of every task; rebuild the per-nnz output row from task ranges
        i1 = idx1.numpy().astype(np.int64)
        i2 = idx2.numpy().astype(np.int64)
        nnz = len(i1)
        rows = np.empty(nnz, dtype=np.int64)
        tr = plan['task_row'].numpy()
        tb = plan['task_begin'].numpy()
        te = plan['task_end'].numpy()
        for r, b, e in zip(tr, tb, te):
            rows[b:e] = r
        vv = np.ones(nnz) if vals is None else vals.numpy()
        contrib = (vv[:, None] * u.numpy()[i1, :])[:, :, None] * v.numpy()[i2, :][:, None, :]
        res = np.zeros((n0, u.shape[1], v.shape[1]))
        np.add.at(res, rows, contrib)
        return torch.from_numpy(res.reshape(n0, -1))

    # ---- K3 doubles -------------------------------------------------------------------------------
    def candidate_capacity(self, topk):
        if topk < 1:
            return 0
        return 16 if topk <= 10 else 32 if topk <= 24 else 64 if topk <= 52 else 0

    def score_splits(self, n_users, KC, prune=False):
        return 1

    def pack_frag(self, M):
        return M.to(torch.float32)

    def pack_frag_bound(self, M, extra=None, extra_scale=0.0):
        ub = self.row_norm_bound(M)
        if extra is not None:
            ub = ub + (extra.double() * extra_scale).to(torch.float32) * (1 + 1e-6)
        return self.pack_frag(M), ub

    def row_norm_bound(self, M):
        return torch.from_numpy((np.linalg.norm(M.numpy(), axis=1) * (1 + 1e-6)).astype(np.float32))

    def tile_norm_bound(self, V):
        nb = (np.linalg.norm(V.numpy(), axis=1) * (1 + 1e-6)).astype(np.float32)
        suf = np.maximum.accumulate(nb[::-1])[::-1]
        return torch.from_numpy(np.ascontiguousarray(suf[::32]))

    def score_candidates(self, Vp, Ep, n_users, n_items, K, seen_ptr, seen_idx, KC, splits=1,
                         user_bound=None, tile_bound=None, seen_tiles=None):
        s = (Ep.numpy() @ Vp.numpy().T).astype(np.float32)
        if seen_ptr is not None:
            sp = seen_ptr.numpy()
            si = seen_idx.numpy()
            for u in range(n_users):
                s[u, si[sp[u]:sp[u + 1]]] = -np.inf
        n_pad = -(-n_users // 32) * 32
        cs = np.full((n_pad, KC), -np.inf, dtype=np.float32)
        ci = np.full((n_pad, KC), -1, dtype=np.int32)
        kk = min(KC, n_items)
        for u in range(n_users):
            order = np.lexsort((np.arange(n_items), -s[u]))[:kk]
            ok = np.isfinite(s[u, order])
            cs[u, :ok.sum()] = s[u, order][ok]
            ci[u, :ok.sum()] = order[ok]
        return torch.from_numpy(cs.ravel()), torch.from_numpy(ci.ravel())

    def score_exit_tiles(self, n_users, splits=1):
        return torch.zeros(splits, -(-n_users // 32), dtype=torch.int64)

    def rescore_topk(self, V, E, n_items, seen_ptr, KC, cs, ci, topk, vmax, want_scores=True, splits=1, out=None,
                     rows=None, n_rows_dev=None, e_err=None, e_exact=False, v32=None):
        n_users, K = E.shape
        if rows is not None:
            # re-do of a user list: compute everybody, keep only the listed users' results
            res = self.rescore_topk(V, E, n_items, seen_ptr, KC, cs, ci, topk, vmax, want_scores, splits, None, None,
                                    None, e_err, e_exact)
            sel = rows.long() if n_rows_dev is None else rows[:int(n_rows_dev[0])].long()
            for dst, src in zip(out, res):
                dst[sel] = src[sel]
            return out
        Vn, En = V.numpy(), E.numpy()
        use32 = v32 is not None and e_err is not None and not e_exact
        if use32:
            Vn = v32.numpy()[:, :K].astype(np.float64)      # scored against the fp32 image; its rounding joins delta
        cs = cs.numpy().reshape(-1, KC)
        ci = ci.numpy().reshape(-1, KC)
        out_idx = np.full((n_users, topk), -1, dtype=np.int64)
        out_s = np.full((n_users, topk), -np.inf)
        flags = np.zeros(n_users, dtype=np.int32)
        for u in range(n_users):
            valid = ci[u] >= 0
            idx = ci[u][valid]
            s = Vn[idx] @ En[u]
            order = np.lexsort((idx, -s))
            idx, s = idx[order], s[order]
            n = min(topk, len(idx))
            out_idx[u, :n] = idx[:n]
            out_s[u, :n] = s[:n]
            n_seen = int(seen_ptr[u + 1] - seen_ptr[u]) if seen_ptr is not None else 0
            if n_items - n_seen < topk:
                flags[u] |= 2
            delta = float(e_err[u]) * 2.0 ** -24 * (1 + 1e-6) * vmax if e_err is not None else 0.0
            if use32:
                delta += np.linalg.norm(En[u]) * 2.0 ** -24 * (1 + 1e-6) * vmax
            if delta > 0 and not e_exact:
                ss = np.r_[s, -np.inf]
                m = min(topk, len(s))
                if m and (ss[:m] - ss[1:m + 1] <= 2 * delta).any():
                    flags[u] |= 4
            if n_items - n_seen < topk:
                pass
            elif valid.all():
                bound = (3 * 2.0 ** -16 + (4 * K + 10) * 2.0 ** -23) * np.linalg.norm(En[u]) * vmax   # rescore.hip: split-bf16 sweep
                tau = float(cs[u, KC - 1]); tau += abs(tau) * 2.0 ** -15
                slack = delta if e_exact else 2 * delta
                if bound > 0 and not (s[topk - 1] - tau > bound + slack):
                    flags[u] |= 4 if (not e_exact and delta > 0 and s[topk - 1] - tau > bound + delta) else 1
        res = torch.from_numpy(out_idx), torch.from_numpy(out_s), torch.from_numpy(flags)
        if out is not None:
            for dst, src in zip(out, res):
                dst.copy_(src)
            return out
        return res

    def flag_compact(self, flags, mask=7):
        hit = torch.nonzero((flags & mask) != 0).flatten().to(torch.int32)
        lst = torch.zeros(max(flags.numel(), 1), dtype=torch.int32)
        lst[:hit.numel()] = hit
        return lst, torch.tensor([hit.numel()], dtype=torch.int32)

    def fold_rows(self, A, lst, cnt, V, E, row_offset=0):
        sel = lst[:int(cnt[0])].long() + row_offset
        if sel.numel():
            E[sel, :V.shape[1]] = torch.from_numpy(np.ascontiguousarray((A.m[sel.numpy()] @ V.numpy())))

    def score_exact_rows(self, rows, V, E, n_items, seen_ptr, seen_idx, topk):
        Vn, En = V.numpy(), E.numpy()
        out_idx = np.full((len(rows), topk), -1, dtype=np.int64)
        out_s = np.full((len(rows), topk), -np.inf)
        for r, u in enumerate(rows.numpy()):
            s = Vn @ En[u]
            cls = np.zeros(n_items, dtype=np.int64)
            if seen_ptr is not None:
                cls[seen_idx.numpy()[int(seen_ptr[u]):int(seen_ptr[u + 1])]] = 1
            order = np.lexsort((np.arange(n_items), -s, cls))[:topk]
            out_idx[r, :len(order)] = order
            out_s[r, :len(order)] = s[order]
        return torch.from_numpy(out_idx), torch.from_numpy(out_s)

    def score_exact_list(self, lst, cnt, V, E, n_items, seen_ptr, seen_idx, topk, out_idx, out_s, n_wg=128):
        rows = lst[:int(cnt)]
        if len(rows):
            ex_idx, ex_s = self.score_exact_rows(rows, V, E, n_items, seen_ptr, seen_idx, topk)
            out_idx[rows.long()] = ex_idx
            out_s[rows.long()] = ex_s

    def eval_ranks(self, recs, hold_row, hold_item):
        r, hr, hi = recs.numpy(), hold_row.numpy(), hold_item.numpy()
        match = r[hr] == hi[:, None]
        return torch.from_numpy(np.where(match.any(1), match.argmax(1) + 1, 0).astype(np.int32))

    def eval_metrics(self, recs, topk, hold_ptr, hold_item, hold_rel=None, hold_pos=None, not_rated_penalty=0.0,
                     switch_positive=0.0, alternative=True):
        """test double of HipOps.eval_metrics: the same 16 sums through the host formulas (polara_amd.evaluation)"""
        from polara_amd import evaluation as ev
        r = recs.numpy()[:, :topk]
        ptr = hold_ptr.numpy()
        users = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
        rel = None if hold_rel is None else hold_rel.numpy()
        pos = None if hold_pos is None else hold_pos.numpy().astype(bool)
        m = ev._Matched(r, users, hold_item.numpy(), rel, pos)
        tp, fp, tn, fn = ev._relevance_counts(m, not_rated_penalty, per_key=True)
        rs = ev.get_relevance_scores(m, not_rated_penalty)
        n = m.n_users
        out = np.zeros(16)
        out[0], out[1], out[3] = tp.sum(), np.sum(fp), fn.sum()
        out[2] = 0.0 if tn is None else tn.sum()
        out[4], out[5], out[8] = rs.precision * n, rs.recall * n, rs.miss_rate * n
        out[6] = 0.0 if rs.fallout is None else rs.fallout * n
        out[7] = 0.0 if rs.specifity is None else rs.specifity * n
        out[9], out[10] = ev.get_arhr_score(m) * n, ev.get_mrr_score(m) * n
        rk = ev.get_ranking_scores(m, m.topk, switch_positive if pos is not None else None, alternative)
        out[11], out[12] = rk.map * n, rk.ndcg * n
        out[13] = 0.0 if rk.ndcl is None else rk.ndcl * n
        out[14], out[15] = m.n_valid_recs.sum(), len(users)
        return out

    def unique_count(self, ids, n_bins):
        return int(len(np.unique(ids.numpy())))

    def dense_scores(self, V, E):
        return E @ V.t()
