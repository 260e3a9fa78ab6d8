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


def test_a_sparse_edit_between_the_sampled_rows_is_the_documented_blind_spot():
    """what the sample cannot see: single entries of rows that are not sampled -- tuning.host_twins = False
    is the switch for callers that patch entries in place"""
    tw = _HostTwins()
    ctx = object()
    a = np.zeros((200_000, 4))
    blk = _block(a)
    tw.remember(ctx, a, blk)
    idx = set(tw.items[-1][3].tolist())
    row = next(r for r in range(1000, 200_000) if r not in idx)
    a[row, 2] = 1.0
    assert tw.lookup(ctx, a) is blk
