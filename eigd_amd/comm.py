"""
Mode sharding across the GPUs of one node.

The per-mode adjoint solves are independent, so rank r of P owns modes r, r+P, r+2P, ...
(block-cyclic: the slower high modes are spread evenly).  A, B, the factor, Phi and B Phi are
replicated; every rank computes them with the same deterministic kernels, so no broadcast is
needed.  The only data-path collective is ONE all-reduce (sum, fp64) of the df/dx vector -- the
sum over the modes of reference eigd/eigenvector_derivatives.py:93-134 / 135-180 -- on the device
buffer the partial sums already live in.

``RcclComm``: RCCL over xGMI through the C ABI (``eigd_comm_init`` / ``eigd_allreduce_sum`` of
include/eigd_hip.h; no PyTorch).  One process per GPU; the 128-byte RCCL unique id travels from
rank 0 to the other rank processes through a file rendezvous on the node.
``TorchDistComm``: torch.distributed process group -- kept for the CPU tests only (``gloo``,
world size 2), where no GPU and therefore no RCCL exists.
"""

import ctypes as C
import os
import time

import numpy as np


class SerialComm:
    rank, size = 0, 1

    def allreduce_sum(self, a):
        return a

    def allreduce_max(self, x):
        return x

    def barrier(self):
        pass


def rendezvous_dir():
    """
    Directory through which the rank processes of ONE launch find each other: ``EIGD_COMM_DIR`` if the launcher set
    it (bench.py --gpus N does), else a name made of the launcher's pid and start time (all ranks of a
    ``torch.distributed.run`` launch share the parent process; the start time keeps a recycled pid apart) and the
    rendezvous port.
    """
    d = os.environ.get("EIGD_COMM_DIR")
    if d:
        return d
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat") as fh:
            start = fh.read().rsplit(")", 1)[1].split()[19]
    except OSError:
        start = "0"
    port = os.environ.get("MASTER_PORT", "0")
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"eigd_comm_{ppid}_{start}_{port}")


def exchange_unique_id(rank, size, make_id, tag="uid", timeout=300.0):
    """rank 0 calls ``make_id()`` and publishes the bytes (atomic rename); the other ranks wait for the file"""
    d = rendezvous_dir()
    path = os.path.join(d, f"{tag}.bin")
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        uid = make_id()
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                uid = fh.read()
            if len(uid) > 0:
                return uid
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no unique id from rank 0 under {d} after {timeout:.0f} s")
        time.sleep(0.02)


class RcclComm:
    """
    RCCL communicator of the rank processes of one node, bound through the C ABI.  ``allreduce_sum`` takes a device
    block (in place, on the context's stream -- what the total derivative uses) or a numpy array (uploaded, reduced,
    downloaded: the N x N coefficient matrices of pgmres / pcpg).
    """

    _generation = 0

    def __init__(self, ctx, rank=None, size=None):
        from ._ffi import c_vp, call

        self.ctx = ctx
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.size = int(os.environ.get("WORLD_SIZE", "1")) if size is None else int(size)
        self.backend = "rccl"
        uid = None
        if self.size > 1:
            def make_id():
                buf = C.create_string_buffer(128)
                call("eigd_comm_unique_id", buf)
                return buf.raw

            RcclComm._generation += 1
            uid = exchange_unique_id(self.rank, self.size, make_id, tag=f"uid{RcclComm._generation}")
        h = c_vp()
        call("eigd_comm_init", ctx.h, self.size, self.rank, uid, C.byref(h))
        self.h = h

    def close(self):
        from . import _ffi

        if getattr(self, "h", None) is not None and self.ctx.h is not None:
            _ffi.lib().eigd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce_sum_device(self, blk):
        """in place on a contiguous device block; ordered on the block's context stream"""
        from ._ffi import call

        if blk.ld != blk.k:
            raise ValueError("contiguous block expected")
        call("eigd_allreduce_sum", self.h, blk.ptr, blk.n * blk.k)
        return blk

    def allreduce_sum(self, a):
        from .device import DeviceBlock

        if isinstance(a, DeviceBlock):
            return self.allreduce_sum_device(a)
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.size == 1:
            return a
        blk = self.ctx.from_host(a.reshape(-1, 1))
        self.allreduce_sum_device(blk)
        return blk.get().reshape(a.shape)

    def allreduce_max(self, x):
        from ._ffi import call

        if self.size == 1:
            return float(x)
        blk = self.ctx.from_host(np.array([float(x)]))
        call("eigd_allreduce_max", self.h, blk.ptr, 1)
        return float(blk.get()[0, 0])

    def barrier(self):
        self.allreduce_max(0.0)


class TorchDistComm:
    """torch.distributed process group as the collective transport: the CPU tests' stand-in (gloo) for RcclComm"""

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
