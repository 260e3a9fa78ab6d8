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
    it (bench.py --gpus N does: a fresh mkdtemp), else a name made of the launcher's pid and start time (all ranks of a
    ``torch.distributed.run`` launch share the parent process; the start time keeps a recycled pid apart), the
    rendezvous port and the elastic run id.
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
    run = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "") if ch.isalnum())[:32]
    # (a worker restart of the same agent keeps pid, start time, port and run id: the restart count tells the rounds apart)
    rnd = "".join(ch for ch in os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") if ch.isdigit())[:6] or "0"
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"eigd_comm_{os.getuid()}_{ppid}_{start}_{port}_{run}_{rnd}")


def _process_start_time():
    """wall-clock start of this process (a unique id older than every rank of this launch belongs to another launch)"""
    try:
        with open("/proc/self/stat") as fh:
            ticks = int(fh.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as fh:
            up = float(fh.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except (OSError, ValueError, IndexError):
        return time.time()


_STALE_SLACK_S = 120.0   # ranks of one launch start within this window of each other


def _own_private_dir(d):
    """create the rendezvous directory for this user only; refuse one that somebody else prepared"""
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not private to this user")


def exchange_unique_id(rank, size, make_id, tag="uid", timeout=300.0):
    """
    rank 0 calls ``make_id()`` and publishes the bytes (atomic rename; whatever an earlier launch left under the same
    name is removed first); the other ranks wait for a file that is not older than this launch.  ``retire_unique_id``
    removes it once every rank has used it.
    """
    d = rendezvous_dir()
    path = os.path.join(d, f"{tag}.bin")
    if rank == 0:
        _own_private_dir(d)
        try:
            os.unlink(path)
        except OSError:
            pass
        uid = make_id()
        tmp = path + f".tmp{os.getpid()}"
        try:
            os.unlink(tmp)
        except OSError:
            pass
        # mode 0600 whatever the umask: the readers accept only a file that nobody else can write (under umask 002 a
        # plain open() would publish 0664, which every other rank rejects until its timeout)
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(uid)
        os.chmod(tmp, 0o600)
        os.replace(tmp, path)
        return uid
    t0 = time.monotonic()
    oldest = _process_start_time() - _STALE_SLACK_S
    me = os.getuid()
    while True:
        try:
            sd, sf = os.stat(d), os.stat(path)
            # only what this user published in a directory nobody else can write to (rank 0 creates it 0700) counts
            private = sd.st_uid == me and not (sd.st_mode & 0o077) and sf.st_uid == me and not (sf.st_mode & 0o022)
            if private and sf.st_mtime >= oldest:
                with open(path, "rb") as fh:
                    uid = fh.read()
                if len(uid) > 0:
                    return uid
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no unique id from rank 0 under {d} after {timeout:.0f} s")
        time.sleep(0.02)


def retire_unique_id(rank, tag="uid"):
    """after a collective that every rank has passed: rank 0 removes the published id (nothing stale for a later launch)"""
    if rank != 0:
        return
    d = rendezvous_dir()
    try:
        os.unlink(os.path.join(d, f"{tag}.bin"))
    except OSError:
        pass
    if not os.environ.get("EIGD_COMM_DIR"):
        try:
            os.rmdir(d)
        except OSError:
            pass


def _init_watchdog(rank, size, seconds):
    """
    ncclCommInitRank waits for every rank and has no timeout of its own: if a peer died before it got there this
    process would wait for ever.  A timer thread (the ctypes call releases the GIL) ends the process instead.
    """
    import sys
    import threading

    def fire():
        print(f"[eigd_amd.comm] rank {rank} of {size}: communicator not up after {seconds:.0f} s "
              "(a peer never reached eigd_comm_init?) -- giving up", file=sys.stderr, flush=True)
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


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
        tag = None
        watchdog = None
        if self.size > 1:
            def make_id():
                buf = C.create_string_buffer(128)
                call("eigd_comm_unique_id", buf)
                return buf.raw

            RcclComm._generation += 1
            tag = f"uid{RcclComm._generation}"
            limit = float(os.environ.get("EIGD_COMM_INIT_TIMEOUT", "600"))
            uid = exchange_unique_id(self.rank, self.size, make_id, tag=tag, timeout=min(300.0, limit))
            watchdog = _init_watchdog(self.rank, self.size, limit)
        h = c_vp()
        try:
            call("eigd_comm_init", ctx.h, self.size, self.rank, uid, C.byref(h))
        finally:
            if watchdog is not None:
                watchdog.cancel()
        self.h = h
        if self.size > 1:
            self.barrier()                 # every rank holds the communicator: the published id can go
            retire_unique_id(self.rank, tag)

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
        """in place on a contiguous device block of the communicator's context (ordered on that context's stream)"""
        from ._ffi import call

        if blk.ld != blk.k:
            raise ValueError("contiguous block expected")
        if blk.ctx is not self.ctx:
            # the collective is enqueued on the communicator's context stream: a block of another (forked) context
            # would be reduced without ordering against the work that produced it
            raise ValueError("the block belongs to another context than the communicator")
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
