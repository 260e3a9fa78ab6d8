"""
Host logic of the device twins (eigd_amd/device.py: _HostTwins), which lets add_total_derivative reuse the device copies
of Phib and of a returned psi: validation against the host array itself -- identity, shape, a content sample -- runs on
the CPU; no GPU call is made here (the blocks are stand-ins).
"""
import gc
from types import SimpleNamespace

import numpy as np

from eigd_amd.device import _HostTwins


def _block(a):
    return SimpleNamespace(n=a.shape[0], k=a.shape[1], ld=a.shape[1])


def test_lookup_needs_the_same_object_context_shape_and_sampled_content():
    tw = _HostTwins()
    ctx, other = object(), object()
    a = np.random.default_rng(0).uniform(size=(50_000, 8))
    blk = _block(a)
    tw.remember(ctx, a, blk)
    assert tw.lookup(ctx, a) is blk
    assert tw.lookup(other, a) is None                       # another context: its own copy
    assert tw.lookup(ctx, a.copy()) is None                  # an equal array is not the same array
    a *= 1.0 + 1e-12                                         # every entry moved in the last digits
    assert tw.lookup(ctx, a) is None and tw.lookup(ctx, a) is None   # seen, and the stale pair is gone for good
    tw.remember(ctx, a, blk)
    a[:, 3] = 0.0                                            # one column: every sampled row sees it
    assert tw.lookup(ctx, a) is None
    tw.remember(ctx, a, blk)
    a[0, 0] += 1.0                                           # first page
    assert tw.lookup(ctx, a) is None
    tw.remember(ctx, a, blk)
    a[-1, -1] += 1.0                                         # last page
    assert tw.lookup(ctx, a) is None
    tw.remember(ctx, a, blk)
    idx = tw.items[-1][3]
    a[idx[len(idx) // 2], 1] += 1e-9                         # a sampled row in the middle
    assert tw.lookup(ctx, a) is None


def test_the_registry_is_weak_and_bounded():
    tw = _HostTwins()
    ctx = object()
    arrays = [np.full((2000, 4), float(i)) for i in range(6)]
    for a in arrays:
        tw.remember(ctx, a, _block(a))
    assert len(tw.items) == tw.KEEP
    assert tw.lookup(ctx, arrays[0]) is None and tw.lookup(ctx, arrays[-1]) is not None
    del arrays, a
    gc.collect()
    b = np.zeros((2000, 4))
    tw.remember(ctx, b, _block(b))                           # dead entries are purged when the next pair comes
    assert len(tw.items) == 1
    tw.clear()
    assert tw.lookup(ctx, b) is None


def test_a_sparse_edit_between_the_sampled_rows_is_only_missed_by_the_opt_in_sampled_copies():
    """what a content sample cannot see -- single entries of rows that are not sampled -- concerns ``tuning.host_twins =
    True`` alone: the default keeps no copy of a caller-owned array at all (``owned_only``), so the edit IS seen"""
    tw = _HostTwins()
    ctx = object()
    a = np.zeros((200_000, 4))
    blk = _block(a)
    tw.remember(ctx, a, blk)
    idx = set(tw.items[-1][3].tolist())
    row = next(r for r in range(1000, 200_000) if r not in idx)
    a[row, 2] = 1.0
    assert tw.lookup(ctx, a, owned_only=True) is None        # the default: nothing kept for the caller's own arrays
    tw.remember(ctx, a, blk)
    assert tw.lookup(ctx, a) is blk                          # the opt-in mode's blind spot


def test_an_array_the_library_returned_is_read_only_and_its_copy_dies_with_that():
    """Context.twin_adopt hands psi out read-only: an in-place edit raises; once the caller makes the array writable the
    kept device block is not handed out again (and the library itself does the same before it writes into such an array)"""
    from eigd_amd.device import Context, _twins, writable_result

    ctx = Context.__new__(Context)                           # no device: the registry is host logic
    a = np.random.default_rng(1).uniform(size=(40_000, 8))   # 2.5 MB: above the twin threshold
    blk = _block(a)
    Context.twin_adopt(ctx, a, blk)
    assert not a.flags.writeable
    try:
        a[123, 4] = 0.0
        raise AssertionError("the edit went through")
    except ValueError:
        pass
    assert _twins.lookup(ctx, a, owned_only=True) is blk and _twins.lookup(ctx, a, owned_only=True) is blk
    a.flags.writeable = True
    a[123, 4] = 0.0                                          # the caller's edit, an unsampled row or not
    assert _twins.lookup(ctx, a, owned_only=True) is None
    Context.twin_adopt(ctx, a, blk)
    writable_result(a)[:] = 1.0                              # the library updating a psi in place (reference 386-389)
    assert a.flags.writeable and _twins.lookup(ctx, a, owned_only=True) is None
    _twins.clear()


def test_released_device_blocks_never_call_the_driver_from_the_releasing_thread():
    """a block may die on any thread (garbage collector): the pool only moves pointers under its lock; blocks over budget
    wait in ``surplus`` for the next allocation of the owning context"""
    import threading

    from eigd_amd import device as dev

    pool = dev._DevicePool()
    for i in range(dev._POOL_KEEP_PER_SIZE + 3):
        pool.give(1000 + i, 1 << 28)                         # 256 MB blocks: at most _POOL_KEEP_PER_SIZE are kept
    assert len(pool.free[1 << 28]) == dev._POOL_KEEP_PER_SIZE and len(pool.surplus) == 3
    ptr, surplus = pool.take(1 << 28)
    assert ptr is not None and len(surplus) == 3 and pool.surplus == []
    errs = []

    def hammer(base):
        try:
            for i in range(2000):
                pool.give(base + i, 4096)
                pool.take(4096)
        except Exception as exc:                             # pragma: no cover
            errs.append(exc)

    ts = [threading.Thread(target=hammer, args=(10_000_000 * (t + 1),)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    ptrs = pool.drain()
    assert pool.bytes == 0 and len(set(ptrs)) == len(ptrs)
