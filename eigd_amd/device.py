"""
Thin Python objects over the C ABI: device context, device arrays (row-major n x k
blocks and stacks of them), CSR matrices, symbolic analysis and the numeric factor.

Nothing here computes on the host; every method is one call into libeigd_hip.so.
"""

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import c_vp, call, hptr

_default_ctx = None


class Context:
    """One device + one stream."""

    def __init__(self, device=0, _parent=None):
        h = c_vp()
        if _parent is None:
            call("eigd_ctx_create", int(device), C.byref(h))
        else:
            call("eigd_ctx_fork", _parent.h, C.byref(h))
        self.h = h
        self.device = int(device)
        self.parent = _parent

    def fork(self, index=0):
        """child context number `index` on the same device (own stream); created once and kept"""
        kids = self.__dict__.setdefault("_children", {})
        if index not in kids:
            kids[index] = Context(self.device, _parent=self)
        return kids[index]

    def make_current(self):
        """bind the calling host thread to this context's device (worker threads start on device 0)"""
        call("eigd_ctx_make_current", self.h)

    def sync(self):
        call("eigd_sync", self.h)

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        call("eigd_mem_info", self.h, C.byref(f), C.byref(t))
        return f.value, t.value

    def timer_start(self):
        call("eigd_timer_start", self.h)

    def timer_stop_ms(self):
        ms = C.c_double()
        call("eigd_timer_stop_ms", self.h, C.byref(ms))
        return ms.value

    def close(self):
        if self.h is not None and self.h.value:
            self.trim_pool()
            _ffi.lib().eigd_ctx_destroy(self.h)
            self.h = None

    # -- allocation helpers -------------------------------------------------
    def empty(self, n, k=1):
        return DeviceBlock(self, n, k)

    def zeros(self, n, k=1):
        b = DeviceBlock(self, n, k)
        b.zero()
        return b

    def from_host(self, a):
        src = a
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a is src:
            _pinned.note_source(a)      # the caller's own buffer (no conversion copy): worth page-locking if it comes again
        if a.ndim == 1:
            b = DeviceBlock(self, a.shape[0], 1)
        elif a.ndim == 2:
            b = DeviceBlock(self, a.shape[0], a.shape[1])
        else:
            raise ValueError("expected a vector or a matrix")
        b.set(a)
        return b

    def stack(self, ns, n, k=1):
        return DeviceStack(self, ns, n, k)

    # -- device twins of host arrays (the reference's two-call surface hands the same arrays in twice) ---------------
    def twin_upload(self, a):
        """
        Device block holding the host array ``a`` for READ-ONLY use.  The reference's surface hands psi back to
        add_total_derivative / eval_adjoint_residual_norm as solve_adjoint returned it: with ``tuning.host_twins ==
        "returned"`` (the default) the device block such an array was downloaded from is kept and handed out again instead
        of a second PCIe transfer -- as long as the array is still the read-only array the library returned (see
        twin_adopt: an in-place edit raises, a caller that makes it writable again gets a transfer).  Arrays the caller
        owns (Phib) are transferred every time: the library reads what it is given, as the reference does
        (eigenvector_derivatives.py:1830-1870, 2167-2207).  ``tuning.host_twins = True`` also keeps copies of caller-owned
        arrays, validated against a content sample only (1024 rows + first and last page: an edit between the sampled
        rows goes unnoticed -- opt-in); ``False`` keeps nothing.
        """
        from . import tuning

        mode = tuning.host_twins
        big = isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 2 and a.flags.c_contiguous and a.nbytes >= _PIN_MIN_BYTES
        if not (mode and big):
            return self.from_host(a)
        blk = _twins.lookup(self, a, owned_only=(mode != True))  # noqa: E712 (True = every array, "returned" = the library's own)
        if blk is None:
            blk = self.from_host(a)
            if mode is True:
                _twins.remember(self, a, blk)
        return blk

    def twin_adopt(self, a, block):
        """
        ``a`` has just been downloaded from ``block`` (which nobody modifies any more): keep the pair (see twin_upload).
        The array goes back to the caller READ-ONLY: writing into it raises (numpy's "assignment destination is read-only")
        instead of leaving a stale device copy behind; ``a.flags.writeable = True`` (or a copy) gives a writable array, and
        the next call that is handed it transfers it.
        """
        from . import tuning

        if tuning.host_twins and isinstance(a, np.ndarray) and a.ndim == 2 and a.nbytes >= _PIN_MIN_BYTES and block.ld == block.k:
            a.flags.writeable = False
            _twins.remember(self, a, block, owned=True)

    def project_stats(self):
        """(projections measured, updates applied) by project_norm2 since the last call; resets the counters"""
        out = np.zeros(2, dtype=np.int32)
        call("eigd_project_stats", self.h, hptr(out))
        return int(out[0]), int(out[1])

    def fetch_colnorm2(self, k):
        """host copy of the last colnorm2_dev result: waits for that copy only, not for work enqueued since"""
        out = np.empty(k)
        call("eigd_colnorm2_fetch", self.h, hptr(out), int(k))
        return out

    def workspace_stack(self, tag, ns, n, k=1):
        """
        A stack that is kept alive between calls and handed out again for the same (tag, shape): the Krylov
        histories are tens of GB at 1 M dof and hipMalloc / hipFree of such blocks costs up to a second.
        """
        cache = self.__dict__.setdefault("_ws", {})
        st = cache.get(tag)
        if st is None or (st.ns, st.n, st.k) != (int(ns), int(n), int(k)):
            cache.pop(tag, None)
            st = DeviceStack(self, ns, n, k)
            cache[tag] = st
        return st

    def release_workspaces(self):
        self.__dict__.pop("_ws", None)
        self.trim_pool()

    def trim_pool(self):
        """give the pooled (released, reusable) device blocks back to the driver"""
        pool = self.__dict__.get("_pool")
        ptrs = pool.drain() if pool is not None else []
        _pinned.trim()
        _twins.clear()
        if self.h is None:
            return
        for ptr in ptrs:
            _ffi.lib().eigd_free(self.h, c_vp(ptr))


def default_context():
    """Context on the device named by EIGD_DEVICE / LOCAL_RANK (default 0)."""
    import os

    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("EIGD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        cnt = C.c_int()
        call("eigd_device_count", C.byref(cnt))
        _default_ctx = Context(dev % max(cnt.value, 1))
    return _default_ctx


# ---------------------------------------------------------------------------
# page-locked host memory for the numpy call surface
# ---------------------------------------------------------------------------
_PIN_MIN_BYTES = 1 << 20             # smaller blocks: the staging of a pageable copy costs nothing that matters
_PIN_POOL_MAX_BYTES = 8 * 1024**3    # released result buffers kept for reuse (a step returns the same shapes again)
_PIN_KEEP_PER_SIZE = 3
_PIN_FIRST_USE_MAX_BYTES = 32 << 20  # results of this size and more are page-locked from their second appearance on


class _PinnedHost:
    """
    Results that go back to the caller as numpy arrays (``DeviceBlock.get``) live in page-locked memory: the D2H copy
    runs at the direct-DMA rate instead of being staged through the runtime's bounce buffers, the pages are not
    first-touched by the copy, and when the caller hands the same array back (psi from ``solve_adjoint`` into
    ``add_total_derivative``) the H2D copy is a direct DMA as well.  The buffer returns to a pool when the last numpy
    view of it dies.  Arrays the caller allocated itself are page-locked in place (hipHostRegister) from the second time
    the same buffer arrives -- the right-hand sides of a design loop are handed in again every step -- and released when
    the array is garbage collected.
    """

    def __init__(self):
        import threading

        self.free = {}            # nbytes -> [host pointers]
        self.held = 0
        self.seen = {}            # id(array) -> [weak reference, sightings] of a caller-owned array
        self.registered = {}      # address -> nbytes
        self.asked = {}           # nbytes -> results of that size asked for so far
        self.surplus = []         # buffers beyond the pool's budget: freed by the next caller of empty() / trim()
        self._lock = threading.RLock()   # (mode groups on several streams upload and download from their own host threads)

    def empty(self, shape):
        import weakref

        nbytes = int(np.prod(shape)) * 8
        self._free_surplus()
        with self._lock:
            lst = self.free.get(nbytes)
            ptr = lst.pop() if lst else None
            if ptr is not None:
                self.held -= nbytes
            else:
                self.asked[nbytes] = self.asked.get(nbytes, 0) + 1
                first = self.asked[nbytes] < 2
        if ptr is None:
            # page-locking costs about what the staged copy of the same bytes costs twice over (62 ms for 255 MB): it
            # pays from the second result of a size on -- a one-off download (the eigenvectors after a solve) goes to an
            # ordinary array
            if first and nbytes >= _PIN_FIRST_USE_MAX_BYTES:
                return np.empty(shape)
            h = c_vp()
            try:
                call("eigd_host_alloc", nbytes, C.byref(h))
            except _ffi.EigdHipError:
                return np.empty(shape)                     # no page-locked memory left: an ordinary array
            ptr = h.value
        buf = (C.c_char * nbytes).from_address(ptr)
        fin = weakref.finalize(buf, self._release, ptr, nbytes)  # fires when the last view of the array is gone
        fin.atexit = False                                  # (at interpreter exit the process gives the memory back)
        return np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def _release(self, ptr, nbytes):
        """finaliser of a result array (any thread, any time): list moves only, see _DevicePool"""
        with self._lock:
            lst = self.free.setdefault(nbytes, [])
            if len(lst) < _PIN_KEEP_PER_SIZE and self.held + nbytes <= _PIN_POOL_MAX_BYTES:
                lst.append(ptr)
                self.held += nbytes
            else:
                self.surplus.append(ptr)

    def _free_surplus(self):
        with self._lock:
            ptrs, self.surplus = self.surplus, []
        for ptr in ptrs:
            try:
                _ffi.lib().eigd_host_free(c_vp(ptr))
            except Exception:
                pass

    def trim(self):
        """give the pooled page-locked buffers back (Context.trim_pool does this together with the device pool)"""
        with self._lock:
            ptrs = [p for lst in self.free.values() for p in lst]
            self.free.clear()
            self.held = 0
            self.surplus.extend(ptrs)
        self._free_surplus()

    def note_source(self, a):
        """a large caller-owned array is about to be uploaded: page-lock it in place from its second upload on"""
        import weakref

        if a.nbytes < _PIN_MIN_BYTES or not a.flags.owndata:
            return
        with self._lock:
            addr = a.ctypes.data
            if self.registered.get(addr) == a.nbytes:
                return
            # sightings are counted per array OBJECT (a fresh array that happens to land on a freed one's address is a
            # first sighting: a loop that allocates its right-hand sides anew every step never pays for page-locking)
            ent = self.seen.get(id(a))
            if ent is None or ent[0]() is not a:
                if len(self.seen) > 64:
                    self.seen.clear()
                self.seen[id(a)] = [weakref.ref(a), 1]
                return
            ent[1] += 1
            if ent[1] != 2:                                # (second sighting locks; a failed attempt is not repeated)
                return
            if addr in self.registered:                    # an array resized in place left a stale registration here
                self._unregister(addr)
            try:
                call("eigd_host_register", c_vp(addr), a.nbytes)
            except (_ffi.EigdHipError, ValueError):
                return
            self.registered[addr] = a.nbytes
            weakref.finalize(a, self._unregister, addr).atexit = False

    def _unregister(self, addr):
        with self._lock:
            had = self.registered.pop(addr, None) is not None
        if had:
            try:
                _ffi.lib().eigd_host_unregister(c_vp(addr))
            except Exception:
                pass


_pinned = _PinnedHost()


class _HostTwins:
    """the (host array, device block) pairs of Context.twin_upload: weak on the host side, at most four"""

    KEEP = 4

    def __init__(self):
        import threading

        self.items = []           # [weakref to the array, context, block, sampled row indices, their content, owned], newest last
        self._lock = threading.RLock()

    @staticmethod
    def _rows(a):
        n = a.shape[0]
        per_page = max(1, 4096 // max(a.strides[0], 1))
        idx = np.unique(np.concatenate([np.arange(min(per_page, n)), np.linspace(0, n - 1, min(n, 1024)).astype(np.int64),
                                        np.arange(max(0, n - per_page), n)]))
        return idx

    def remember(self, ctx, a, block, owned=False):
        """owned: the library made ``a`` (a download of ``block``) and handed it out read-only"""
        import weakref

        idx = None if owned else self._rows(a)
        with self._lock:
            self.items = [it for it in self.items if it[0]() is not None and it[0]() is not a]
            self.items.append([weakref.ref(a), ctx, block, idx, None if owned else a[idx].copy(), owned])
            del self.items[: max(0, len(self.items) - self.KEEP)]

    def lookup(self, ctx, a, owned_only=False):
        with self._lock:
            for it in self.items:
                if it[0]() is a and it[1] is ctx:
                    blk = it[2]
                    if it[5]:
                        # the library's own array: nobody can have written into it while it stayed read-only
                        ok = (blk.n, blk.k) == a.shape and not a.flags.writeable
                    else:
                        ok = (not owned_only) and (blk.n, blk.k) == a.shape and np.array_equal(a[it[3]], it[4])
                    if ok:
                        return blk
                    self.items.remove(it)               # the array was changed (or made writable, or reshaped): forget the copy
                    return None
        return None

    def forget(self, a):
        """``a`` is about to be modified by the library itself (a psi handed back in as the initial guess)"""
        with self._lock:
            self.items = [it for it in self.items if it[0]() is not None and it[0]() is not a]

    def clear(self):
        with self._lock:
            self.items = []


_twins = _HostTwins()


def writable_result(a):
    """
    the library is about to write into the caller's array ``a`` (a psi updated in place, reference 386-389, 793, 968,
    1184): if it is one of the read-only arrays the library handed out itself (Context.twin_adopt), it becomes writable
    again and its kept device copy is dropped
    """
    if isinstance(a, np.ndarray) and not a.flags.writeable:
        _twins.forget(a)
        a.flags.writeable = True       # (raises for memory that is read-only for good: the caller's business, as in the reference)
    return a


def pinned_empty(shape):
    """numpy float64 array in page-locked memory (uploads and downloads of it run at the direct-DMA rate)"""
    return _pinned.empty(tuple(int(v) for v in np.atleast_1d(shape)))


_POOL_KEEP_PER_SIZE = 16          # blocks of one size kept for reuse (a short-recurrence step holds ten n x k blocks at once, the numpy surface two more)
_POOL_MAX_BYTES = 24 * 1024**3     # and in total (the Krylov workspaces have their own cache)


class _DevicePool:
    """
    Released device blocks of one context, keyed by size.  A block can die on ANY host thread and at any time (the
    garbage collector runs finalisers wherever an allocation happens to trigger it; mode groups on several streams have
    their own host threads): ``give`` therefore only moves pointers between lists under the pool's lock and never calls
    into HIP.  Blocks beyond the pool's budget wait in ``surplus`` and are freed by the next ``take`` / ``trim`` --
    calls made by code that is using the context, on its thread, in stream order.
    """

    def __init__(self):
        import threading

        self.lock = threading.Lock()
        self.free = {}      # nbytes -> [device pointers]
        self.bytes = 0
        self.surplus = []   # pointers to hand back to the driver

    def take(self, nbytes):
        with self.lock:
            lst = self.free.get(nbytes)
            ptr = lst.pop() if lst else None
            if ptr is not None:
                self.bytes -= nbytes
            surplus, self.surplus = self.surplus, []
        return ptr, surplus

    def give(self, ptr, nbytes):
        with self.lock:
            lst = self.free.setdefault(nbytes, [])
            # (small blocks: more of one size may be in flight at once -- hipFree stalls the stream)
            if len(lst) < max(_POOL_KEEP_PER_SIZE, (1 << 30) // max(nbytes, 1)) and self.bytes + nbytes <= _POOL_MAX_BYTES:
                lst.append(ptr)
                self.bytes += nbytes
            else:
                self.surplus.append(ptr)

    def drain(self):
        with self.lock:
            ptrs = [p for lst in self.free.values() for p in lst] + self.surplus
            self.free, self.bytes, self.surplus = {}, 0, []
        return ptrs


def _pool_of(ctx):
    pool = ctx.__dict__.get("_pool")
    if pool is None:
        pool = ctx.__dict__.setdefault("_pool", _DevicePool())   # (dict.setdefault is atomic: one pool per context)
    return pool


class _Buffer:
    """
    Owning device allocation.  Released blocks go to a per-context pool keyed by size and are handed out again:
    a step allocates the same n x k temporaries every time, and hipMalloc / hipFree of 256 MB blocks cost up to
    milliseconds (hipFree also waits for the device).  Stream order makes the reuse safe: all work of a context goes
    through its one stream, and a block is only ever handed out again by the context it came from.
    """

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr, surplus = _pool_of(ctx).take(self.nbytes)
        for p in surplus:
            _ffi.lib().eigd_free(ctx.h, c_vp(p))
        if self.ptr is not None:
            return
        p = c_vp()
        try:
            call("eigd_malloc", ctx.h, self.nbytes, C.byref(p))
        except _ffi.EigdHipError:
            ctx.trim_pool()  # out of memory: give the pooled blocks back and try once more
            call("eigd_malloc", ctx.h, self.nbytes, C.byref(p))
        self.ptr = p.value

    def __del__(self):
        try:
            ptr, self.ptr = getattr(self, "ptr", None), None
            if ptr and self.ctx.h is not None:
                _pool_of(self.ctx).give(ptr, self.nbytes)
        except Exception:
            pass


class DeviceBlock:
    """n x k block of doubles, row-major with leading dimension ld (numpy C order)."""

    def __init__(self, ctx, n, k, buf=None, offset=0, ld=None):
        self.ctx, self.n, self.k = ctx, int(n), int(k)
        self.ld = self.k if ld is None else int(ld)
        if buf is None:
            buf = _Buffer(ctx, 8 * max(self.n * self.ld, 1))
            offset = 0
        self.buf = buf
        self.offset = int(offset)  # in doubles

    @property
    def ptr(self):
        return c_vp(self.buf.ptr + 8 * self.offset)

    @property
    def shape(self):
        return (self.n, self.k)

    def cols(self, c0, c1):
        """view of columns [c0, c1) (same rows, same leading dimension)"""
        return DeviceBlock(self.ctx, self.n, c1 - c0, self.buf, self.offset + c0, self.ld)

    def rows(self, r0, r1):
        """view of rows [r0, r1) (same columns, same leading dimension)"""
        return DeviceBlock(self.ctx, r1 - r0, self.k, self.buf, self.offset + r0 * self.ld, self.ld)

    def zero(self):
        if self.ld == self.k:
            call("eigd_memset", self.ctx.h, self.ptr, 0, 8 * self.n * self.k)
        else:
            self.assign_lincomb([(0.0, self)])

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(self.n, self.k)
        if self.ld == self.k:
            call("eigd_h2d", self.ctx.h, self.ptr, hptr(a), 8 * self.n * self.k)
        else:
            tmp = self.ctx.from_host(a)
            self.copy_from(tmp)

    def get(self):
        if self.ld == self.k:
            nbytes = 8 * self.n * self.k
            out = _pinned.empty((self.n, self.k)) if nbytes >= _PIN_MIN_BYTES else np.empty((self.n, self.k))
            call("eigd_d2h", self.ctx.h, hptr(out), self.ptr, nbytes)
            return out
        tmp = self.ctx.empty(self.n, self.k)
        tmp.copy_from(self)
        return tmp.get()

    def copy_from(self, src):
        if (src.n, src.k) != (self.n, self.k):
            raise ValueError("shape mismatch in copy")
        if src.ld == src.k and self.ld == self.k:
            call("eigd_d2d", self.ctx.h, self.ptr, src.ptr, 8 * self.n * self.k)
        else:
            for c0 in range(0, self.k, 64):
                c1 = min(self.k, c0 + 64)
                call("eigd_copy_block", self.ctx.h, self.n, c1 - c0, src.cols(c0, c1).ptr, src.ld,
                     self.cols(c0, c1).ptr, self.ld)
        return self

    def copy(self):
        return self.ctx.empty(self.n, self.k).copy_from(self)

    # ---- panel operations --------------------------------------------------
    def assign_lincomb(self, terms):
        """self[:, c] = sum_t coef_t[c] * X_t[:, c]; coef scalar or length-k array; <= 4 terms"""
        nt = len(terms)
        k = self.k
        for c0 in range(0, k, 64):
            c1 = min(k, c0 + 64)
            kb = c1 - c0
            coef = np.empty((nt, kb))
            ptrs = (c_vp * nt)()
            lds = (C.c_int * nt)()
            for t, (cf, X) in enumerate(terms):
                coef[t, :] = np.broadcast_to(np.asarray(cf, dtype=np.float64), (k,))[c0:c1]
                ptrs[t] = X.cols(c0, c1).ptr
                lds[t] = X.ld
            call("eigd_lincomb", self.ctx.h, self.n, kb, self.cols(c0, c1).ptr, self.ld, nt, ptrs, lds, hptr(coef))
        return self

    def coldot(self, other):
        out = np.empty(self.k)
        for c0 in range(0, self.k, 64):
            c1 = min(self.k, c0 + 64)
            tmp = np.empty(c1 - c0)
            call("eigd_coldot", self.ctx.h, self.n, c1 - c0, self.cols(c0, c1).ptr, self.ld,
                 other.cols(c0, c1).ptr, other.ld, hptr(tmp))
            out[c0:c1] = tmp
        return out

    def coldot_dd(self, other):
        """column-wise dots in twice the working precision: (hi, lo) arrays with hi + lo = the dot products"""
        hi, lo = np.empty(self.k), np.empty(self.k)
        for c0 in range(0, self.k, 64):
            c1 = min(self.k, c0 + 64)
            tmp = np.empty(2 * (c1 - c0))
            call("eigd_coldot_dd", self.ctx.h, self.n, c1 - c0, self.cols(c0, c1).ptr, self.ld,
                 other.cols(c0, c1).ptr, other.ld, hptr(tmp))
            hi[c0:c1], lo[c0:c1] = tmp[: c1 - c0], tmp[c1 - c0:]
        return hi, lo

    def colnorms(self):
        return np.sqrt(self.coldot(self))

    def colnorm2_dev(self):
        """squared column norms, left on the device (no host synchronisation); ctx.fetch_colnorm2(k) collects the copy"""
        assert self.k <= 64
        out = self.ctx.empty(1, self.k)
        call("eigd_colnorm2_dev", self.ctx.h, self.n, self.k, self.ptr, self.ld, out.ptr)
        return out

    def assign_scaled_inverse(self, src, norm2_dev, skip):
        """self[:, c] = src[:, c] / sqrt(norm2[c]) with norm2 on the device; zero where skip[c] or the norm is zero"""
        assert self.k == src.k and self.k <= 64
        mask = np.ascontiguousarray(np.asarray(skip, dtype=bool)[: self.k], dtype=np.uint8)
        call("eigd_scale_inv_norm", self.ctx.h, self.n, self.k, src.ptr, src.ld, self.ptr, self.ld, norm2_dev.ptr, hptr(mask))
        return self

    def pair_orthonormalise(self, norm2_dev, W1, W2, skip):
        """
        self = [T1 | T2] (n x 2k): W1 = T1/|T1|, T2 <- T2 - (T1.T2/|T1|^2) T1 (in place), W2 = T2/|T2|, all on the device.
        norm2_dev: the 2k squared column norms of self (project_norm2).  Returns the device block
        [|T1|^2 | T1.T2 | |T2|^2 | W1.T2] (4k); ctx.fetch_colnorm2(4k) collects its host copy.
        """
        k = W1.k
        assert self.k == 2 * k and W2.k == k and 4 * k <= 128
        mask = np.ascontiguousarray(np.asarray(skip, dtype=bool)[:k], dtype=np.uint8)
        out = self.ctx.empty(1, 4 * k)
        call("eigd_pair_orthonormalise", self.ctx.h, self.n, k, self.ptr, self.ld, norm2_dev.ptr, W1.ptr, W1.ld, W2.ptr,
             W2.ld, hptr(mask), out.ptr)
        return out

    def tdot(self, X):
        """self^T X  -> host (self.k x X.k)"""
        out = np.empty((self.k, X.k))
        for a0 in range(0, self.k, 64):
            a1 = min(self.k, a0 + 64)
            for b0 in range(0, X.k, 64):
                b1 = min(X.k, b0 + 64)
                tmp = np.empty((a1 - a0, b1 - b0))
                call("eigd_gemm_tn", self.ctx.h, self.n, a1 - a0, b1 - b0, self.cols(a0, a1).ptr, self.ld, 1,
                     X.cols(b0, b1).ptr, X.ld, hptr(tmp))
                out[a0:a1, b0:b1] = tmp
        return out

    def add_product(self, U, Cmat, alpha=1.0, beta=1.0):
        """self = beta * self + alpha * U @ C, C on the host (U.k x self.k)"""
        Cmat = np.ascontiguousarray(Cmat, dtype=np.float64).reshape(U.k, self.k)
        for b0 in range(0, self.k, 64):
            b1 = min(self.k, b0 + 64)
            bb = beta
            for a0 in range(0, U.k, 64):
                a1 = min(U.k, a0 + 64)
                cc = np.ascontiguousarray(Cmat[a0:a1, b0:b1])
                call("eigd_gemm_nn", self.ctx.h, self.n, a1 - a0, b1 - b0, U.cols(a0, a1).ptr, U.ld, 1, hptr(cc),
                     self.cols(b0, b1).ptr, self.ld, float(alpha), float(bb))
                bb = 1.0
        return self

    def project(self, U, V):
        """self <- self - U (V^T self)   (reference _project, eigenvector_derivatives.py:26-30)"""
        if U.k > 128 or self.k > 64:
            t = V.tdot(self)
            return self.add_product(U, t, alpha=-1.0, beta=1.0)
        call("eigd_project", self.ctx.h, self.n, U.k, self.k, U.ptr, U.ld, V.ptr, V.ld, self.ptr, self.ld)
        return self

    def project_to(self, U, V, Cdev, tol=0.0, flag=None, norm2=None):
        """self <- self - U (V^T self) with the coefficients V^T self written to the device block Cdev (U.k x self.k)
        and no host synchronisation; tol > 0: the update is applied only where it matters (measured on the device: some
        coefficient above tol times the norm of its column -- ``norm2``: a 1 x k device block of squared norms, e.g. the
        B-norms from coldot_dev; without it the Euclidean norms as in project_norm2) and flag (a 1 x 1 device block)
        receives 1.0 / 0.0"""
        if U.k > 64 or self.k > 64 or (Cdev.n, Cdev.k) != (U.k, self.k):
            raise ValueError("project_to: panels of at most 64 columns, coefficient block U.k x self.k")
        if norm2 is not None and norm2.n * norm2.k != self.k:
            raise ValueError("project_to: one squared norm per column of the block")
        call("eigd_project_to", self.ctx.h, self.n, U.k, self.k, U.ptr, U.ld, V.ptr, V.ld, self.ptr, self.ld, Cdev.ptr,
             Cdev.ld, float(tol), flag.ptr if flag is not None else None, norm2.ptr if norm2 is not None else None)
        return self

    def coldot_dev(self, other, out):
        """column-wise dots with ``other`` into the device block ``out`` (1 x k), no host synchronisation"""
        if self.k > 64 or (other.n, other.k) != (self.n, self.k) or out.n * out.k != self.k:
            raise ValueError("coldot_dev: blocks of the same shape with at most 64 columns, one result per column")
        call("eigd_coldot_dev", self.ctx.h, self.n, self.k, self.ptr, self.ld, other.ptr, other.ld, out.ptr)
        return out

    def svqb_step(self, BX, Cdev, first, flag, update_bx=True):
        """one SVQB pass of the B-orthonormalisation of this block on the device (no host round trip): self and BX are
        multiplied in place by the transform of their Gram matrix self^T BX, Cdev (p x p device block) <- Cq Cdev with
        self_in = self_out Cq (first: Cdev <- Cq), flag (1 x 1 device block) tells a numerically dependent direction;
        update_bx=False leaves BX as it is (the caller recomputes B X from the finished block)"""
        if self.k > 32 or (BX.n, BX.k) != (self.n, self.k) or (Cdev.n, Cdev.k) != (self.k, self.k) or Cdev.ld != self.k:
            raise ValueError("svqb_step: blocks of at most 32 columns, a contiguous p x p coefficient block")
        call("eigd_svqb_step", self.ctx.h, self.n, self.k, self.ptr, self.ld, BX.ptr, BX.ld, Cdev.ptr, 1 if first else 0,
             flag.ptr, 1 if update_bx else 0)
        return self

    def project_norm2(self, U, V, uscale=0.0, tol=0.0):
        """project(U, V), then the squared column norms of the result: as colnorm2_dev (device block + pinned copy for
        ctx.fetch_colnorm2), but formed while the projection writes the block -- no extra pass over it.  ``uscale``: the
        largest Euclidean column norm of U (makes the measured update independent of the scale of the inner product);
        ``tol``: relative size of an update that matters (0: 1e-13)"""
        if U.k > 128 or self.k > 64:
            return self.project(U, V).colnorm2_dev()
        out = self.ctx.empty(1, self.k)
        call("eigd_project_norm2", self.ctx.h, self.n, U.k, self.k, U.ptr, U.ld, V.ptr, V.ld, self.ptr, self.ld, out.ptr,
             float(uscale), float(tol))
        return out

    def gather_cols(self, cols):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = self.ctx.empty(self.n, len(cols))
        for c0 in range(0, len(cols), 64):
            c1 = min(len(cols), c0 + 64)
            sub = np.ascontiguousarray(cols[c0:c1])
            call("eigd_gather_cols", self.ctx.h, self.n, c1 - c0, self.ptr, self.ld, hptr(sub),
                 out.cols(c0, c1).ptr, out.ld)
        return out

    def gather_cols_into(self, dst, cols):
        """dst[:, j] = self[:, cols[j]] into an existing block (a slab of a narrower Krylov stack)"""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        if (dst.n, dst.k) != (self.n, len(cols)):
            raise ValueError("shape mismatch in column gather")
        for c0 in range(0, len(cols), 64):
            c1 = min(len(cols), c0 + 64)
            sub = np.ascontiguousarray(cols[c0:c1])
            call("eigd_gather_cols", self.ctx.h, self.n, c1 - c0, self.ptr, self.ld, hptr(sub),
                 dst.cols(c0, c1).ptr, dst.ld)
        return dst

    def scatter_cols_into(self, dst, cols):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        for c0 in range(0, len(cols), 64):
            c1 = min(len(cols), c0 + 64)
            sub = np.ascontiguousarray(cols[c0:c1])
            call("eigd_scatter_cols", self.ctx.h, self.n, c1 - c0, self.cols(c0, c1).ptr, self.ld, hptr(sub),
                 dst.ptr, dst.ld)
        return dst


class DeviceStack:
    """ns slabs, each an n x k row-major block with ld = k, contiguous in one allocation."""

    def __init__(self, ctx, ns, n, k=1):
        self.ctx, self.ns, self.n, self.k = ctx, int(ns), int(n), int(k)
        self.slab = self.n * self.k
        self.buf = _Buffer(ctx, 8 * max(self.ns * self.slab, 1))

    def __getitem__(self, j):
        if not 0 <= j < self.ns:
            raise IndexError(j)
        return DeviceBlock(self.ctx, self.n, self.k, self.buf, j * self.slab, self.k)

    def zero(self):
        call("eigd_memset", self.ctx.h, self.ptr, 0, 8 * self.ns * self.slab)
        return self

    @property
    def ptr(self):
        return c_vp(self.buf.ptr)

    def slab_ptr(self, j):
        return c_vp(self.buf.ptr + 8 * j * self.slab)

    def dot(self, T, ns=None, j0=0, c0=0):
        """H[j, c] = S_j[:, c0 + c] . T[:, c]  for j in [j0, j0+ns), c in [0, T.k)"""
        ns = self.ns - j0 if ns is None else ns
        H = np.empty((ns, T.k))
        if ns > 0:
            call("eigd_stack_dot", self.ctx.h, self.n, T.k, ns, c_vp(self.buf.ptr + 8 * (j0 * self.slab + c0)),
                 self.slab, self.k, T.ptr, T.ld, hptr(H))
        return H

    def axpy_into(self, T, H, alpha=1.0, j0=0, c0=0):
        """T[:, c] += alpha * sum_j S_{j0+j}[:, c0 + c] H[j, c]"""
        H = np.ascontiguousarray(H, dtype=np.float64).reshape(-1, T.k)
        ns = H.shape[0]
        step = max(1, (60 * 1024) // (8 * T.k))
        for a in range(0, ns, step):
            b = min(ns, a + step)
            call("eigd_stack_axpy", self.ctx.h, self.n, T.k, b - a, c_vp(self.buf.ptr + 8 * ((j0 + a) * self.slab + c0)),
                 self.slab, self.k, hptr(np.ascontiguousarray(H[a:b])), T.ptr, T.ld, float(alpha))
        return T

    def cgs2(self, T, ns, c0=0, tol=1e-13):
        """one Gram-Schmidt step of T against slabs [0, ns) (columns c0..): returns (coefficients, passes over the stack)"""
        H = np.empty((ns, T.k))
        passes = C.c_int(0)
        call("eigd_stack_cgs2", self.ctx.h, self.n, T.k, ns, c_vp(self.buf.ptr + 8 * c0), self.slab, self.k, T.ptr, T.ld,
             float(tol), hptr(H), C.byref(passes))
        return H, passes.value

    def cgs2_pair(self, T, ns, c0=0, tol=1e-13):
        """Gram-Schmidt step of the pair T = [T1 | T2] (n x 2k) against columns c0..c0+k-1 of slabs [0, ns): both blocks in
        the same two passes over the stack.  Returns (coefficients ns x 2k, passes)"""
        k = T.k // 2
        H = np.empty((ns, T.k))
        passes = C.c_int(0)
        call("eigd_stack_cgs2_pair", self.ctx.h, self.n, k, ns, c_vp(self.buf.ptr + 8 * c0), self.slab, self.k, T.ptr, T.ld,
             float(tol), hptr(H), C.byref(passes))
        return H, passes.value

    def axpy_dot_into(self, T, H1, alpha=1.0, j0=0, c0=0):
        """T += alpha * sum_j S_j H1[j]; returns H2[j, c] = S_j[:, c] . T[:, c] on the updated T (ns <= 32)"""
        H1 = np.ascontiguousarray(H1, dtype=np.float64).reshape(-1, T.k)
        ns = H1.shape[0]
        H2 = np.empty((ns, T.k))
        call("eigd_stack_axpy_dot", self.ctx.h, self.n, T.k, ns, c_vp(self.buf.ptr + 8 * (j0 * self.slab + c0)),
             self.slab, self.k, hptr(H1), T.ptr, T.ld, float(alpha), hptr(H2))
        return H2

    # k == 1 stacks double as column-major n x ns matrices (the Lanczos basis)
    def tdot_block(self, X, ns=None):
        """V[:, :ns]^T X -> host (ns x X.k); only for k == 1 stacks"""
        assert self.k == 1
        ns = self.ns if ns is None else ns
        out = np.empty((ns, X.k))
        for a0 in range(0, ns, 64):
            a1 = min(ns, a0 + 64)
            for b0 in range(0, X.k, 64):
                b1 = min(X.k, b0 + 64)
                tmp = np.empty((a1 - a0, b1 - b0))
                call("eigd_gemm_tn", self.ctx.h, self.n, a1 - a0, b1 - b0, self.slab_ptr(a0), 1, self.slab,
                     X.cols(b0, b1).ptr, X.ld, hptr(tmp))
                out[a0:a1, b0:b1] = tmp
        return out

    def times_into(self, X, Cmat, ns=None, alpha=1.0, beta=0.0):
        """X = beta X + alpha V[:, :ns] @ C ; only for k == 1 stacks"""
        assert self.k == 1
        ns = self.ns if ns is None else ns
        Cmat = np.ascontiguousarray(Cmat, dtype=np.float64).reshape(ns, X.k)
        for b0 in range(0, X.k, 64):
            b1 = min(X.k, b0 + 64)
            bb = beta
            for a0 in range(0, ns, 64):
                a1 = min(ns, a0 + 64)
                cc = np.ascontiguousarray(Cmat[a0:a1, b0:b1])
                call("eigd_gemm_nn", self.ctx.h, self.n, a1 - a0, b1 - b0, self.slab_ptr(a0), 1, self.slab, hptr(cc),
                     X.cols(b0, b1).ptr, X.ld, float(alpha), float(bb))
                bb = 1.0
        return X


class DevicePanels:
    """
    A tall basis of up to ncols columns stored as row-major panels of 64 columns (n x 64, leading dimension 64): the
    layout the tall-skinny MFMA kernels stream best (a row of a panel is 512 contiguous bytes; eigd_gemm_tn /
    eigd_gemm_nn take their operand fragments straight from it).  The restarted block Lanczos keeps V and B V this way:
    its Gram-Schmidt step against c basis vectors is ceil(c / 64) products per pass instead of c dependent axpys.
    Same small interface as a k = 1 DeviceStack where the adjoint stage reads the basis (tdot_block / times_into).
    """

    PW = 64

    def __init__(self, ctx, ncols, n):
        self.ctx, self.n = ctx, int(n)
        self.npanels = max(1, -(-int(ncols) // self.PW))
        self.ncols = self.npanels * self.PW
        self.buf = _Buffer(ctx, 8 * self.npanels * self.n * self.PW)

    def view(self, j0, j1):
        """columns [j0, j1) of ONE panel as a strided block"""
        q = j0 // self.PW
        assert 0 <= j0 < j1 <= self.ncols and (j1 - 1) // self.PW == q
        return DeviceBlock(self.ctx, self.n, j1 - j0, self.buf, q * self.n * self.PW + (j0 - q * self.PW), self.PW)

    def pieces(self, j0, j1):
        """[(view, a, b)]: the panels' shares of columns [j0, j1); a, b count from j0"""
        out = []
        j = j0
        while j < j1:
            e = min(j1, (j // self.PW + 1) * self.PW)
            out.append((self.view(j, e), j - j0, e - j0))
            j = e
        return out

    def tdot_block(self, X, ns=None, j0=0):
        """V[:, j0:j0+ns]^T X -> host (ns x X.k)"""
        ns = self.ncols - j0 if ns is None else ns
        out = np.empty((ns, X.k))
        for blk, a, b in self.pieces(j0, j0 + ns):
            out[a:b] = blk.tdot(X)
        return out

    def times_into(self, X, Cmat, ns=None, alpha=1.0, beta=0.0, j0=0):
        """X = beta X + alpha V[:, j0:j0+ns] @ C"""
        ns = self.ncols - j0 if ns is None else ns
        Cmat = np.ascontiguousarray(Cmat, dtype=np.float64).reshape(ns, X.k)
        bb = beta
        for blk, a, b in self.pieces(j0, j0 + ns):
            X.add_product(blk, Cmat[a:b], alpha=alpha, beta=bb)
            bb = 1.0
        return X

    def times_panels(self, out, Cmat, ns):
        """out[:, :kx] = V[:, :ns] @ C (ns x kx) with ``out`` another DevicePanels of the same n or a DeviceBlock (n x kx):
        every panel of V is read once per 80 output columns and the result written once (eigd_panels_times; at most 192
        basis columns per call)"""
        Cmat = np.ascontiguousarray(Cmat, dtype=np.float64)
        panels = isinstance(out, DevicePanels)
        if Cmat.ndim != 2 or Cmat.shape[0] != ns or out.n != self.n or ns > self.ncols or ns > 192 or (
                Cmat.shape[1] > out.ncols if panels else Cmat.shape[1] != out.k):
            raise ValueError("shape mismatch in the product of a panel basis with a coefficient matrix")
        if out.buf is self.buf:
            raise ValueError("the result must not overlap the basis")
        if panels:
            call("eigd_panels_times", self.ctx.h, self.n, int(ns), int(Cmat.shape[1]), c_vp(self.buf.ptr), self.n * self.PW,
                 hptr(Cmat), c_vp(out.buf.ptr), out.n * out.PW, out.PW, out.PW)
        else:
            call("eigd_panels_times", self.ctx.h, self.n, int(ns), int(Cmat.shape[1]), c_vp(self.buf.ptr), self.n * self.PW,
                 hptr(Cmat), out.ptr, 0, 1 << 30, out.ld)
        return out

    def get_block(self, j0, p, out=None):
        """contiguous n x p copy of columns [j0, j0+p)"""
        X = out if out is not None else self.ctx.empty(self.n, p)
        for blk, a, b in self.pieces(j0, j0 + p):
            X.cols(a, b).copy_from(blk)
        return X

    def set_block(self, j0, X):
        for blk, a, b in self.pieces(j0, j0 + X.k):
            blk.copy_from(X.cols(a, b))

    def to_host(self, m):
        out = np.empty((self.n, m))
        for blk, a, b in self.pieces(0, m):
            out[:, a:b] = blk.get()
        return out

    def from_host(self, V):
        V = np.asarray(V, dtype=np.float64)
        for blk, a, b in self.pieces(0, V.shape[1]):
            blk.set(np.ascontiguousarray(V[:, a:b]))

    def as_stack(self, m):
        """the first m columns as a k = 1 stack (column-major n x m): for consumers written against that layout"""
        st = DeviceStack(self.ctx, m, self.n, 1)
        for blk, a, b in self.pieces(0, m):
            for q in range(b - a):
                st[a + q].copy_from(blk.cols(q, q + 1))
        return st

    def swap(self, other):
        self.buf, other.buf = other.buf, self.buf


class CSRMatrix:
    """device copy of a scipy CSR matrix"""

    def __init__(self, ctx, A):
        from scipy import sparse

        A = sparse.csr_matrix(A)
        if A.dtype != np.float64:
            A = A.astype(np.float64)
        self.ctx = ctx
        self.shape = A.shape
        self.n, self.ncols = A.shape  # rectangular: the gather / averaging / filter maps of the design-variable chain
        self.nnz = int(A.nnz)
        ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        dv = np.ascontiguousarray(A.data, dtype=np.float64)
        h = c_vp()
        call("eigd_csr_upload_rect", ctx.h, self.n, self.ncols, self.nnz, hptr(ip), hptr(ix), hptr(dv), C.byref(h))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None and self.ctx.h is not None:
                _ffi.lib().eigd_mat_free(self.h)
                self.h = None
        except Exception:
            pass

    def apply(self, X, Y=None, alpha=1.0, beta=0.0):
        """Y = alpha A X + beta Y"""
        if Y is None:
            Y = X.ctx.empty(self.n, X.k)
        if X.n != self.ncols or (Y.n, Y.k) != (self.n, X.k):
            raise ValueError("shape mismatch in SpMM")
        call("eigd_spmm_on", X.ctx.h, self.h, X.ptr, X.ld, Y.ptr, Y.ld, X.k, float(alpha), float(beta))  # on X's stream
        return Y

    dtype = np.dtype(np.float64)

    def matvec(self, x):
        """numpy in, numpy out: lets scipy's aslinearoperator wrap a matrix that lives on the device"""
        x = np.asarray(x, dtype=np.float64)
        out = self.apply(self.ctx.from_host(x.reshape(self.ncols, -1))).get()
        return out[:, 0] if x.ndim == 1 else out

    matmat = matvec

    def update_values_device(self, vals):
        """new values (device block of nnz doubles, in this matrix's CSR order), same sparsity"""
        if vals.n * vals.k < self.nnz:
            raise ValueError("value count does not match the matrix")
        call("eigd_csr_update_values_dev", self.h, vals.ptr)

    def spmv_bytes(self, k=1):
        """algorithmic bytes of one product"""
        if k == 1:
            return 12.0 * self.nnz + 20.0 * self.n
        return 12.0 * self.nnz + 4.0 * self.n + 16.0 * self.n * k


_I32 = ("perm", "iperm", "f_c0", "f_ns", "f_bs", "f_parent", "f_level", "f_slot", "f_npanels", "border", "rel",
        "lvl_ptr", "lvl_fronts", "lvl_nsteps", "v_src")
_I64 = ("f_bptr", "f_foff", "f_voff", "f_ioff", "a_src", "a_dst")


class Symbolic:
    """Host ordering + symbolic multifrontal analysis of a symmetric sparse pattern."""

    SIZE_NAMES = ("n", "nfronts", "nlevels", "nnzL", "front_doubles", "sumd", "maxd", "border_len", "nlower",
                  "nlaunch_steps", "flops", "maxns")

    def __init__(self, A, leaf_size=0, panel_width=0, coords=None):
        from scipy import sparse

        A = sparse.csr_matrix(A)
        A.sort_indices()
        self.n = A.shape[0]
        ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.nnz = int(A.nnz)
        self.pattern_key = self.pattern_fingerprint(ip, ix)
        h = c_vp()
        if coords is None:
            call("eigd_symbolic_create", self.n, hptr(ip), hptr(ix), int(leaf_size), int(panel_width), C.byref(h))
        else:
            xy = np.ascontiguousarray(coords, dtype=np.float64).reshape(self.n, -1)
            call("eigd_symbolic_create_geom", self.n, hptr(ip), hptr(ix), int(leaf_size), int(panel_width),
                 xy.shape[1], hptr(xy), C.byref(h))
        self.h = h
        sz = np.zeros(12, dtype=np.int64)
        call("eigd_symbolic_sizes", self.h, hptr(sz), 12)
        self.sizes = dict(zip(self.SIZE_NAMES, (int(v) for v in sz)))

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None:
                _ffi.lib().eigd_symbolic_free(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def pattern_fingerprint(indptr, indices):
        import hashlib

        hsh = hashlib.blake2b(digest_size=16)
        hsh.update(np.ascontiguousarray(indptr, dtype=np.int32).tobytes())
        hsh.update(np.ascontiguousarray(indices, dtype=np.int32).tobytes())
        return hsh.hexdigest()

    def check_pattern(self, A):
        """a reused analysis must belong to the matrix: the numeric phase scatters the values through its index maps"""
        if A.shape[0] != self.n or int(A.nnz) != self.nnz or self.pattern_fingerprint(A.indptr, A.indices) != self.pattern_key:
            raise ValueError("the symbolic analysis was made for a different sparsity pattern")

    def array(self, name):
        s = self.sizes
        nf = s["nfronts"]
        lens = {
            "perm": s["n"], "iperm": s["n"], "border": s["border_len"], "rel": s["border_len"],
            "lvl_ptr": s["nlevels"] + 1, "lvl_fronts": nf, "lvl_nsteps": s["nlevels"], "v_src": s["sumd"],
            "f_bptr": nf + 1, "a_src": s["nlower"], "a_dst": s["nlower"],
        }
        ln = lens.get(name, nf)
        if name in _I32:
            out = np.empty(max(ln, 1), dtype=np.int32)
            call("eigd_symbolic_get_i32", self.h, name.encode(), hptr(out), ln)
        elif name in _I64:
            out = np.empty(max(ln, 1), dtype=np.int64)
            call("eigd_symbolic_get_i64", self.h, name.encode(), hptr(out), ln)
        else:
            raise KeyError(name)
        return out[:ln]


class Factor:
    """Numeric LL^T factor of a symmetric positive definite CSR matrix on the device."""

    def __init__(self, ctx, A, symbolic=None, leaf_size=0, panel_width=0, coords=None):
        from scipy import sparse

        A = sparse.csr_matrix(A)
        if A.dtype != np.float64:
            A = A.astype(np.float64)
        A.sort_indices()
        self.ctx = ctx
        self.n = A.shape[0]
        if symbolic is not None:
            symbolic.check_pattern(A)
        self.symbolic = symbolic if symbolic is not None else Symbolic(A, leaf_size, panel_width, coords)
        data = np.ascontiguousarray(A.data, dtype=np.float64)
        h = c_vp()
        call("eigd_factor_create", ctx.h, self.symbolic.h, hptr(data), C.byref(h))
        self.h = h
        if self.stats()["static_pivots"] > 0:
            self.verify_static_pivots(CSRMatrix(ctx, A))

    def __del__(self):
        try:
            for h, _ in self.__dict__.get("_lanes", {}).values():
                _ffi.lib().eigd_factor_lane_free(h)
            if getattr(self, "h", None) is not None and self.ctx.h is not None:
                _ffi.lib().eigd_factor_free(self.h)
                self.h = None
        except Exception:
            pass

    def refactor(self, A):
        from scipy import sparse

        A = sparse.csr_matrix(A)
        A.sort_indices()
        self.symbolic.check_pattern(A)
        data = np.ascontiguousarray(A.data, dtype=np.float64)
        call("eigd_factor_refactor", self.h, hptr(data))
        if self.stats()["static_pivots"] > 0:
            self.verify_static_pivots(CSRMatrix(self.ctx, A))

    def refine(self, mat_dev, B, X, alpha=1.0, steps=1):
        """X <- X + mat^{-1} (alpha B - mat X), ``steps`` times: iterative refinement of X ~ alpha mat^{-1} B on device blocks"""
        R = X.ctx.empty(X.n, X.k)
        for _ in range(steps):
            mat_dev.apply(X, R)
            R.assign_lincomb([(alpha, B), (-1.0, R)])
            self.solve_to(R, R, 1.0)
            X.assign_lincomb([(1.0, X), (1.0, R)])
        return X

    STATIC_PIVOT_REFINEMENTS = 3

    def verify_static_pivots(self, mat_dev):
        """
        The factorisation replaced pivots that were singular inside their panel block by +-sqrt(eps) |A| (static pivots,
        see ldlt_bk_inv_kernel): it is the factor of a nearby matrix, good as a preconditioner of a few refinement steps
        -- unless the matrix itself is singular to working precision, which partial pivoting over whole columns
        (SuperLU, reference 13) would have reported.  One refined solve of a random system tells the two apart.
        """
        b = self.ctx.from_host(np.random.default_rng(0).uniform(-1.0, 1.0, size=(self.n, 1)))
        x = self.solve_to(b, self.ctx.empty(self.n, 1))
        self.refine(mat_dev, b, x, steps=self.STATIC_PIVOT_REFINEMENTS)
        r = mat_dev.apply(x)
        r.assign_lincomb([(1.0, r), (-1.0, b)])
        rel = float(r.colnorms()[0] / b.colnorms()[0])
        if not rel < 1e-8:
            raise _ffi.NotPositiveDefiniteError(
                f"the (shifted) matrix is singular to working precision ({self.stats()['static_pivots']} static pivots, "
                f"relative residual {rel:.1e} after refinement): move the shift away from an eigenvalue")

    def refactor_device(self, vals):
        """numeric refactorisation from CSR values that already live on the device (ElementAssembler.assemble)"""
        if vals.n * vals.k < self.symbolic.nnz:
            raise ValueError("fewer values than the analysed pattern has entries")
        call("eigd_factor_refactor_dev", self.h, vals.ptr)

    def solve_inplace(self, X, alpha=1.0):
        if X.n != self.n:
            raise ValueError("shape mismatch in factor solve")
        if X.ctx is not self.ctx:
            return self.solve_to(X, X, alpha)
        call("eigd_factor_solve", self.h, X.ptr, X.ld, X.k, float(alpha))
        return X

    def solve_to(self, Xin, Xout, alpha=1.0):
        """Xout <- alpha * M^{-1} Xin (Xin untouched); runs on the stream of Xout's context"""
        if Xin.n != self.n or (Xout.n, Xout.k) != (Xin.n, Xin.k):
            raise ValueError("shape mismatch in factor solve")
        if Xout.ctx is self.ctx:
            call("eigd_factor_solve_to", self.h, Xin.ptr, Xin.ld, Xout.ptr, Xout.ld, Xin.k, float(alpha))
        else:
            call("eigd_factor_lane_solve_to", self._lane(Xout.ctx), Xin.ptr, Xin.ld, Xout.ptr, Xout.ld, Xin.k,
                 float(alpha))
        return Xout

    def _lane(self, ctx):
        lanes = self.__dict__.setdefault("_lanes", {})
        key = id(ctx)
        if key not in lanes:
            h = c_vp()
            call("eigd_factor_lane_create", self.h, ctx.h, C.byref(h))
            lanes[key] = (h, ctx)
        return lanes[key][0]

    def stats(self):
        out = np.zeros(7)
        call("eigd_factor_stats", self.h, hptr(out), 7)
        return {"nnzL": int(out[0]), "device_bytes": int(out[1]), "flops": float(out[2]), "nfronts": int(out[3]),
                "negative_pivots": int(out[4]), "static_pivots": int(out[5]), "workspace_planes": int(out[6])}

    def solve_bytes(self, k):
        b = C.c_double()
        call("eigd_factor_solve_bytes", self.h, int(k), C.byref(b))
        return b.value


class ElementBilinear:
    """
    Device-side derivative callback: ``cb(W, V) -> numpy (nelem,)`` with
    out[e] = scale[e] * sum_c w_e(:, c)^T M_e v_e(:, c).  Marked ``device = True`` so
    add_total_derivative hands it device blocks (no D2H of the n x N weight matrices).
    """

    device = True

    def __init__(self, ctx, elem_dofs, Me, scale=None, etype=None):
        elem_dofs = np.ascontiguousarray(elem_dofs, dtype=np.int32)
        self.ctx = ctx
        self.nelem, self.nd = elem_dofs.shape
        self._dofs = _Buffer(ctx, elem_dofs.nbytes)
        call("eigd_h2d", ctx.h, c_vp(self._dofs.ptr), hptr(elem_dofs), elem_dofs.nbytes)
        self._etype = None
        if etype is not None:  # Me holds one matrix per element TYPE, etype[e] selects
            etype = np.ascontiguousarray(etype, dtype=np.int32)
            Me = np.ascontiguousarray(Me, dtype=np.float64)
            if etype.shape != (self.nelem,) or Me.ndim != 3 or Me.shape[1:] != (self.nd, self.nd) or (
                    self.nelem and (etype.min() < 0 or etype.max() >= Me.shape[0])):
                raise ValueError("element types do not match the matrices")
            self._etype = _Buffer(ctx, etype.nbytes)
            call("eigd_h2d", ctx.h, c_vp(self._etype.ptr), hptr(etype), etype.nbytes)
            self.per_elem = 2
            self._Me = _Buffer(ctx, Me.nbytes)
            call("eigd_h2d", ctx.h, c_vp(self._Me.ptr), hptr(Me), Me.nbytes)
        elif isinstance(Me, DeviceBlock):  # per-element matrices made on the device (ElementLinearMatrices)
            if Me.n * Me.k != self.nelem * self.nd * self.nd or Me.ld != Me.k:
                raise ValueError("element matrix block does not match the dof list")
            self.per_elem, self._Me = 1, Me
        else:
            Me = np.ascontiguousarray(Me, dtype=np.float64)
            self.per_elem = 1 if Me.ndim == 3 else 0
            if Me.shape[-2:] != (self.nd, self.nd) or (self.per_elem and Me.shape[0] != self.nelem):
                raise ValueError("element matrix shape does not match the dof list")
            self._Me = _Buffer(ctx, Me.nbytes)
            call("eigd_h2d", ctx.h, c_vp(self._Me.ptr), hptr(Me), Me.nbytes)
        self._scale = None
        if isinstance(scale, DeviceBlock):
            if scale.n * scale.k != self.nelem:
                raise ValueError("one scale factor per element expected")
            self._scale = scale
        elif scale is not None:
            scale = np.ascontiguousarray(scale, dtype=np.float64)
            self._scale = _Buffer(ctx, scale.nbytes)
            call("eigd_h2d", ctx.h, c_vp(self._scale.ptr), hptr(scale), scale.nbytes)

    @classmethod
    def from_device(cls, ctx, elem_dofs, Me, scale):
        """element matrices (device block or numpy) and scale factors (device block) that already live on the device"""
        return cls(ctx, elem_dofs, Me, scale)

    @staticmethod
    def _p(obj):
        return obj.ptr if isinstance(obj, DeviceBlock) else c_vp(obj.ptr)

    def accumulate(self, W, V, out, alpha=1.0):
        """out (device, nelem x 1) += alpha * contraction"""
        if (W.n, W.k) != (V.n, V.k):
            raise ValueError("shape mismatch")
        sp = self._p(self._scale) if self._scale is not None else c_vp(None)
        for c0 in range(0, W.k, 64):
            c1 = min(W.k, c0 + 64)
            call("eigd_elem_bilinear", self.ctx.h, self.nelem, self.nd, c_vp(self._dofs.ptr), self._p(self._Me),
                 self.per_elem, c_vp(self._etype.ptr) if self._etype is not None else c_vp(None), sp, W.cols(c0, c1).ptr, W.ld, V.cols(c0, c1).ptr, V.ld, c1 - c0, float(alpha), out.ptr)
        return out

    def __call__(self, W, V):
        if not isinstance(W, DeviceBlock):  # numpy in, numpy out (vector or tensor form of the reference callbacks)
            W2 = np.asarray(W, dtype=np.float64).reshape(W.shape[0], -1)
            V2 = np.asarray(V, dtype=np.float64).reshape(V.shape[0], -1)
            W, V = self.ctx.from_host(W2), self.ctx.from_host(V2)
        out = self.ctx.zeros(self.nelem, 1)
        self.accumulate(W, V, out)
        return out.get()[:, 0]


class GroupedElementDerivative:
    """
    Device callback for design variables that scale GROUPS of elements (wall thicknesses of panels, examples/crm.py
    style): ``cb(W, V)[g] = sum_{e in g} sum_parts part(W, V)[e]`` with ElementBilinear parts; ``accumulate`` adds into a
    device vector of ngroups entries (fixed summation order: a CSR product with the group incidence matrix).
    """

    device = True

    def __init__(self, ctx, parts, group_of_element, ngroups):
        from scipy import sparse

        self.ctx, self.parts = ctx, list(parts)
        group_of_element = np.asarray(group_of_element, dtype=np.int64)
        self.nel = len(group_of_element)
        if any(p.nelem != self.nel for p in self.parts):
            raise ValueError("parts and groups disagree on the number of elements")
        self.nout = self.nelem = int(ngroups)
        self._map = CSRMatrix(ctx, sparse.csr_matrix((np.ones(self.nel), (group_of_element, np.arange(self.nel))),
                                                     shape=(self.nout, self.nel)))

    def accumulate(self, W, V, out, alpha=1.0):
        tmp = self.ctx.zeros(self.nel, 1)
        for part in self.parts:
            part.accumulate(W, V, tmp)
        self._map.apply(tmp, out, alpha=float(alpha), beta=1.0)
        return out

    def __call__(self, W, V):
        if not isinstance(W, DeviceBlock):
            W = self.ctx.from_host(np.asarray(W, dtype=np.float64).reshape(np.shape(W)[0], -1))
            V = self.ctx.from_host(np.asarray(V, dtype=np.float64).reshape(np.shape(V)[0], -1))
        return self.accumulate(W, V, self.ctx.zeros(self.nout, 1)).get()[:, 0]


class ElementAssembler:
    """
    Device-side assembly of element matrices into CSR values (the COO -> CSR loops of the reference's
    harnesses, examples/buckling.py:152-176, 220-255).  The dof lists are analysed once; ``pattern()`` gives the scipy
    CSR pattern, ``assemble(Me, scale)`` the values on the device, in that pattern's order and with a fixed summation
    order (bitwise reproducible).  ``values_into(A, ...)`` / ``refactor(factor, ...)`` feed a CSRMatrix / Factor built on
    the same pattern without a host round trip.
    """

    def __init__(self, ctx, elem_dofs, n):
        elem_dofs = np.ascontiguousarray(elem_dofs, dtype=np.int32)
        self.ctx = ctx
        self.n = int(n)
        self.nelem, self.nd = elem_dofs.shape
        h = c_vp()
        call("eigd_assembler_create", ctx.h, self.n, self.nelem, self.nd, hptr(elem_dofs), C.byref(h))
        self.h = h
        nnz = C.c_int64()
        call("eigd_assembler_nnz", self.h, C.byref(nnz))
        self.nnz = int(nnz.value)
        self._cache = {}

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None and self.ctx.h is not None:
                _ffi.lib().eigd_assembler_free(self.h)
                self.h = None
        except Exception:
            pass

    def pattern(self):
        """scipy CSR matrix with the assembled sparsity (values zero)"""
        from scipy import sparse

        ip = np.empty(self.n + 1, dtype=np.int32)
        ix = np.empty(max(self.nnz, 1), dtype=np.int32)
        call("eigd_assembler_pattern", self.h, hptr(ip), hptr(ix))
        return sparse.csr_matrix((np.zeros(self.nnz), ix[: self.nnz], ip), shape=(self.n, self.n))

    def _upload(self, key, arr, dtype=np.float64, hold=None):
        """
        element matrices / scale factors are uploaded once per distinct host array.  The cache is least-recently-used
        (a hit moves the entry to the end); ``hold`` collects the buffers of the call in progress so that none of them
        is released -- by an eviction or by a replaced entry -- before the launch that reads it has been enqueued.
        """
        ent = self._cache.pop(key, None)
        if ent is None or ent[0] is not arr:
            a = np.ascontiguousarray(arr, dtype=dtype)
            buf = _Buffer(self.ctx, max(a.nbytes, 8))
            call("eigd_h2d", self.ctx.h, c_vp(buf.ptr), hptr(a), a.nbytes)
            if ent is not None and hold is not None:
                hold.append(ent[1])                        # (the replaced buffer may be an operand of this very call)
            ent = (arr, buf)
        self._cache[key] = ent                             # most recently used: last
        if hold is not None:
            hold.append(ent[1])
        while len(self._cache) > 8:  # a handful of distinct operands per assembler (K, G, their parts): drop the oldest
            old_key = next(iter(self._cache))
            old = self._cache.pop(old_key)
            if hold is not None:
                hold.append(old[1])
        return ent[1]

    def assemble(self, Me, scale=None, out=None, etype=None):
        """
        CSR values (device block nnz x 1) of sum_e scale[e] P_e^T Me P_e; Me: (nd, nd) shared or (nelem, nd, nd) numpy
        array, or a device block of nelem * nd * nd doubles (per-element matrices made on the device)
        """
        hold = []  # every cached operand of this call stays alive until its launch is enqueued
        if isinstance(Me, DeviceBlock):
            if Me.n * Me.k != self.nelem * self.nd * self.nd:
                raise ValueError("element matrix block does not match the dof list")
            sp = scale.ptr if isinstance(scale, DeviceBlock) else (
                c_vp(None) if scale is None else c_vp(self._upload("scale", scale, hold=hold).ptr))
            vals = out if out is not None else self.ctx.empty(max(self.nnz, 1), 1)
            call("eigd_assemble", self.h, Me.ptr, 1, c_vp(None), sp, vals.ptr)
            return vals
        Me = np.asarray(Me, dtype=np.float64)
        per_elem = 1 if Me.ndim == 3 else 0
        tp = c_vp(None)
        if etype is not None:  # one matrix per element type
            et = np.asarray(etype)
            if Me.ndim != 3 or et.shape != (self.nelem,) or et.min() < 0 or et.max() >= Me.shape[0]:
                raise ValueError("element types do not match the matrices")
            per_elem = 2
            tp = c_vp(self._upload("etype", etype, np.int32, hold=hold).ptr)
        if Me.shape[-2:] != (self.nd, self.nd) or (per_elem == 1 and Me.shape[0] != self.nelem):
            raise ValueError("element matrix shape does not match the dof list")
        if isinstance(scale, DeviceBlock):
            sp = scale.ptr
        elif scale is None:
            sp = c_vp(None)
        else:
            if np.shape(scale) != (self.nelem,):
                raise ValueError("one scale factor per element expected")
            sp = c_vp(self._upload("scale", scale, hold=hold).ptr)
        vals = out if out is not None else self.ctx.empty(max(self.nnz, 1), 1)
        mp = c_vp(self._upload(("Me", id(Me)), Me, hold=hold).ptr)
        call("eigd_assemble", self.h, mp, per_elem, tp, sp, vals.ptr)
        del hold
        return vals

    def values_to_host(self, vals):
        return vals.get()[: self.nnz, 0]


class ElementLinearMatrices:
    """
    Element matrices that are linear in the element's dof values, ``Me[e] = sum_m (L[m] . u_e) Q[m]`` (device):
    the stress stiffness of a linear pre-buckling state (examples/buckling.py:220-255).  ``full_dofs``: nelem x nd
    indices into the FULL dof vector; ``L``: (nterms, nd); ``Q``: (nterms, nd, nd).
    """

    def __init__(self, ctx, full_dofs, L, Q):
        full_dofs = np.ascontiguousarray(full_dofs, dtype=np.int32)
        L = np.ascontiguousarray(L, dtype=np.float64)
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        self.ctx = ctx
        self.nelem, self.nd = full_dofs.shape
        self.nterms = L.shape[0]
        if L.shape != (self.nterms, self.nd) or Q.shape != (self.nterms, self.nd, self.nd):
            raise ValueError("L / Q shapes do not match the dof list")
        self._dofs = _Buffer(ctx, full_dofs.nbytes)
        call("eigd_h2d", ctx.h, c_vp(self._dofs.ptr), hptr(full_dofs), full_dofs.nbytes)
        self._L = _Buffer(ctx, L.nbytes)
        call("eigd_h2d", ctx.h, c_vp(self._L.ptr), hptr(L), L.nbytes)
        self._Q = _Buffer(ctx, Q.nbytes)
        call("eigd_h2d", ctx.h, c_vp(self._Q.ptr), hptr(Q), Q.nbytes)

    def __call__(self, u_full):
        """u_full: device block (nfull x 1) -> device block of nelem * nd * nd element-matrix entries"""
        out = self.ctx.empty(self.nelem * self.nd * self.nd, 1)
        call("eigd_elem_linear_matrices", self.ctx.h, self.nelem, self.nd, c_vp(self._dofs.ptr), u_full.ptr, self.nterms,
             c_vp(self._L.ptr), c_vp(self._Q.ptr), out.ptr)
        return out

    def adjoint(self, elem_dofs_dev, W, V, scale=None, alpha=1.0):
        """
        d/du_e of sum_c w_c^T Me(u) v_c per element (device block nelem * nd x 1, entry e*nd + a belongs to
        full_dofs[e, a]): the tensor form of examples/buckling.py:283-316.  ``elem_dofs_dev`` is the device dof list
        (a _Buffer, -1 = constrained) through which the n x k blocks W, V are gathered; ``scale`` a device block or None.
        """
        if (W.n, W.k) != (V.n, V.k):
            raise ValueError("shape mismatch")
        out = self.ctx.zeros(self.nelem * self.nd, 1)
        sp = scale.ptr if scale is not None else c_vp(None)
        tmp = out if W.k <= 64 else self.ctx.empty(self.nelem * self.nd, 1)
        for c0 in range(0, W.k, 64):
            c1 = min(W.k, c0 + 64)
            dst = out if c0 == 0 else tmp
            call("eigd_elem_linear_adjoint", self.ctx.h, self.nelem, self.nd, c_vp(elem_dofs_dev.ptr), self.nterms,
                 c_vp(self._L.ptr), c_vp(self._Q.ptr), sp, W.cols(c0, c1).ptr, W.ld, V.cols(c0, c1).ptr, V.ld, c1 - c0,
                 float(alpha), dst.ptr)
            if c0 > 0:
                out.assign_lincomb([(1.0, out), (1.0, tmp)])
        return out
