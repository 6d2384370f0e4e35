"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in CPU tests).

What is exchanged:
  build   : users are row-sharded (nnz-balanced contiguous blocks); the item-side blocks of the solver are
            row-sharded as well (solver.ItemRows).  Per Gramian step ONE all-gather of the block X
            [n_items x l_active fp64] in front of A_p X and ONE reduce-scatter of Z_p = A_p^T (A_p X) behind it
            (= the volume of the sum all-reduce of the replicated layout), plus l x l all-reduces of the Gram
            matrices.  The l x l kernels (Jacobi eigh, Cholesky) are recomputed identically on every rank.
  scoring : V is already replicated; test users are sharded; no collective in the data path.
            Only the final [n_users x topk] int64 result is gathered (host side).
xGMI is point-to-point, so each exchange is launched as ONE large call per step (tens of MB):
few, fat collectives rather than a bucketed stream.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class TorchComm:
    """Communicator interface used by solver.svd_topk and models (rank, world, allreduce, gather_rows)."""

    def __init__(self, group=None, exercise_collectives=False, split_step=False):
        """exercise_collectives: issue every collective even in a group of ONE rank (the shortcuts below return the local
        buffer instead).  A test switch: on a one-GPU box it is the only way to run the RCCL calls of this class — shapes,
        dtypes, contiguity, in-place rules — against the real library (tests/test_gpu_dist.py).  By itself it selects the
        row-sharded item layout of the solver (collectives inside the orthogonalisation); with `split_step` the build takes the
        DEFAULT form of a user-sharded build instead — the library's step in its two halves with ONE all-reduce of the block
        between them (solver._block_lanczos) — so that this sequence too (our kernels on the current stream, RCCL's on its
        own, the library's buffers as operands) meets the real library on a one-GPU box."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; call init_from_env() first')
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.split_step = bool(split_step)
        self._always = bool(exercise_collectives) or self.split_step
        self.bytes_reduced = 0
        self.n_allreduce = 0
        self.bytes_gathered = self.bytes_scattered = 0
        self.n_allgather = self.n_reduce_scatter = 0
        self.n_panel_exchanges = 0            # exchanges started through the *_start forms (two per product: solver.ItemRows.product)

    def allreduce(self, t):
        if self.world > 1 or self._always:
            if t.is_cuda and dist.get_backend(self.group) == 'gloo':
                # debugging aid: several ranks sharing one GPU cannot use RCCL; stage through the host
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.bytes_reduced += t.numel() * t.element_size()
            self.n_allreduce += 1
        return t

    def all_gather_rows(self, local):
        """[rows x b] per rank -> [world*rows x b] on every rank, rank order = row order (the item-side blocks of the
        solver in front of an SpMM: `solver.ItemRows.full`)."""
        if self.world == 1 and not self._always:
            return local
        out = torch.empty((self.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.group) == 'gloo':
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, local.cpu(), group=self.group)
            out.copy_(h)
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)
        self.bytes_gathered += out.numel() * out.element_size()
        self.n_allgather += 1
        return out

    def reduce_scatter_rows(self, full, rows):
        """sum over the ranks of [world*rows x b] blocks, rank r keeps rows [r*rows, (r+1)*rows) (`ItemRows.product`).
        RCCL: one reduce-scatter; gloo has none, so the CPU tests sum everything and slice."""
        if self.world == 1 and not self._always:
            return full
        assert full.shape[0] == self.world * rows and full.is_contiguous()
        if dist.get_backend(self.group) == 'nccl':
            out = torch.empty((rows, full.shape[1]), dtype=full.dtype, device=full.device)
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = full.cpu() if full.is_cuda else full.clone()      # the caller's buffer is not ours to overwrite
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            out = h[self.rank * rows:(self.rank + 1) * rows].to(full.device).contiguous()
        self.bytes_scattered += full.numel() * full.element_size()
        self.n_reduce_scatter += 1
        return out

    # ---- asynchronous forms: the exchange of one column panel runs next to the products of the next one ------------------
    class _Ready:
        def __init__(self, value):
            self.value = value

        def wait(self):
            return self.value

    class _Pending:
        """an RCCL collective in flight on the library's own stream; wait() makes the CURRENT stream wait for it (the host
        does not block) and hands the result over"""

        def __init__(self, work, value):
            self.work, self.value = work, value

        def wait(self):
            self.work.wait()
            return self.value

    def _is_async(self):
        return dist.get_backend(self.group) == 'nccl' and (self.world > 1 or self._always)

    def allreduce_start(self, t, count=True):
        """`allreduce`, started only: under RCCL the call returns at once and the kernels the caller enqueues next run
        while the blocks travel; `count=False` for the further panels of ONE logical exchange (the collective counters of the
        bench line count exchanges, not panels)."""
        self.n_panel_exchanges += 1
        if not self._is_async():
            before = self.n_allreduce
            out = self.allreduce(t)
            if not count:
                self.n_allreduce = before
            return TorchComm._Ready(out)
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.bytes_reduced += t.numel() * t.element_size()
        self.n_allreduce += 1 if count else 0
        return TorchComm._Pending(work, t)

    def reduce_scatter_rows_start(self, full, rows, count=True):
        """`reduce_scatter_rows`, started only (see allreduce_start)."""
        self.n_panel_exchanges += 1
        if not self._is_async():
            before = self.n_reduce_scatter
            out = self.reduce_scatter_rows(full, rows)
            if not count:
                self.n_reduce_scatter = before
            return TorchComm._Ready(out)
        assert full.shape[0] == self.world * rows and full.is_contiguous()
        out = torch.empty((rows, full.shape[1]), dtype=full.dtype, device=full.device)
        work = dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.bytes_scattered += full.numel() * full.element_size()
        self.n_reduce_scatter += 1 if count else 0
        return TorchComm._Pending(work, out)

    def _exchange_device(self):
        """where collective payloads live: the GPU under RCCL, the host under gloo"""
        if dist.get_backend(self.group) == 'nccl':
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    def gather_rows(self, local, n_total, width, dtype=np.int64):
        """Concatenates per-rank row blocks (rank order = row order) on every rank (host arrays): the block heights
        travel in one small all-gather, the blocks — padded to the tallest — in one `all_gather_into_tensor`; typed
        buffers end to end (no pickling: a [n_users x topk] result is tens of MB to GB)."""
        local = np.ascontiguousarray(local, dtype=dtype).reshape(-1, width)
        if self.world == 1 and not self._always:
            assert local.shape == (n_total, width)
            return local
        dev = self._exchange_device()
        rows = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        all_rows = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_rows, rows, group=self.group)
        heights = [int(x) for x in all_rows.tolist()]
        tallest = max(max(heights), 1)
        send = torch.zeros(tallest, width, dtype=torch.from_numpy(local[:0]).dtype, device=dev)
        send[:local.shape[0]] = torch.from_numpy(local).to(dev)
        recv = torch.empty(self.world * tallest, width, dtype=send.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        recv = recv.cpu().numpy().reshape(self.world, tallest, width)
        out = np.concatenate([recv[r, :heights[r]] for r in range(self.world)], axis=0)
        assert out.shape == (n_total, width), (out.shape, n_total, width)
        return out

    def barrier(self):
        if self.world > 1 or self._always:
            dist.barrier(group=self.group)


def init_from_env(backend=None):
    """Initialises the default process group from torchrun's environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR/PORT) and pins this process to its GPU.  Returns TorchComm."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if backend is None:
        backend = 'nccl' if use_cuda else 'gloo'
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return TorchComm()
