"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in CPU tests).

What is exchanged:
  build   : users are row-sharded (nnz-balanced contiguous blocks).  Per Gramian step ONE sum
            all-reduce of Z_p = A_p^T (A_p Q)  [n_items x l_active fp64] and, per Rayleigh-Ritz,
            one of the l x l Gram matrix.  Everything on the item side (X, V_lock, the Jacobi eigh)
            is replicated and recomputed identically on every rank — no broadcast needed.
  scoring : V is already replicated; test users are sharded; no collective in the data path.
            Only the final [n_users x topk] int64 result is gathered (host side).
xGMI is point-to-point, so the Z all-reduce is launched as ONE large call per step (tens of MB):
few, fat collectives rather than a bucketed stream.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class TorchComm:
    """Communicator interface used by solver.svd_topk and models (rank, world, allreduce, gather_rows)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; call init_from_env() first')
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_reduced = 0
        self.n_allreduce = 0

    def allreduce(self, t):
        if self.world > 1:
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
        if self.world == 1:
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
        if self.world > 1:
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
