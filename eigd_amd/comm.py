"""
Mode sharding across the GPUs of one node.

The per-mode adjoint solves are independent, so rank r of P owns modes r, r+P, r+2P, ...
(block-cyclic: the slower high modes are spread evenly).  A, B, the factor, Phi and B Phi are
replicated; every rank computes them with the same deterministic kernels, so no broadcast is
needed.  The only data-path collective is ONE all-reduce (sum, fp64) of the df/dx vector
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" in the CPU tests).
"""

import numpy as np


class SerialComm:
    rank, size = 0, 1

    def allreduce_sum(self, a):
        return a

    def barrier(self):
        pass


class TorchDistComm:
    """torch.distributed process group as the collective transport (plumbing only)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.device = device if device is not None else ("cuda" if self.backend == "nccl" else "cpu")

    def allreduce_sum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        t = self._torch.from_numpy(a.copy()).to(self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(a.shape)

    def allreduce_max(self, x):
        t = self._torch.tensor([float(x)], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self._dist.barrier()


def mode_columns(N, rank, size):
    """columns (modes) owned by a rank"""
    return np.arange(rank, N, size)
