#!/usr/bin/env python3
"""development aid: k-column sweep times of TWO BUILDS of the library in ONE process (plain ctypes on the C ABI, two
factors of the C3 matrix per build, interleaved, alternating) -- sweep times move by +-2 % with where a process's
allocations landed, so builds can only be compared inside one process and over several factors:
    python tools/lib_ab_probe.py build/a/libeigd_hip.so build/b/libeigd_hip.so"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd._ffi import _SIGNATURES  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

vp = C.c_void_p


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.eigd_last_error.restype = C.c_char_p
    for name, args in _SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = C.c_int, args
    return L


def chk(L, rc):
    if rc:
        raise RuntimeError(L.eigd_last_error().decode())


def hp(a):
    return a.ctypes.data_as(vp)


col = BucklingColumn(706, 706, seed=0)
K = col.stiffness().tocsr()
K.sort_indices()
n = K.shape[0]
ip, ix = np.ascontiguousarray(K.indptr, dtype=np.int32), np.ascontiguousarray(K.indices, dtype=np.int32)
data = np.ascontiguousarray(K.data, dtype=np.float64)
xy = np.ascontiguousarray(col.dof_coords(), dtype=np.float64).reshape(n, -1)
paths = sys.argv[1:]
nfac = int(os.environ.get("NFAC", "2"))
builds = []
for path in paths:
    L = load(path)
    ctx, sym = vp(), vp()
    chk(L, L.eigd_ctx_create(0, C.byref(ctx)))
    chk(L, L.eigd_symbolic_create_geom(n, hp(ip), hp(ix), 0, 0, xy.shape[1], hp(xy), C.byref(sym)))
    builds.append([path, L, ctx, sym, []])
for _ in range(nfac):                                   # interleaved: a factor of every build in turn
    for b in builds:
        f = vp()
        chk(b[1], b[1].eigd_factor_create(b[2], b[3], hp(data), C.byref(f)))
        b[4].append(f)
rng = np.random.default_rng(0)
for k in tuple(int(v) for v in os.environ.get("WIDTHS", "8,32").split(",")):
    Bh = np.ascontiguousarray(rng.normal(size=(n, k)))
    bufs = []
    for b in builds:
        L, ctx = b[1], b[2]
        dB, dX = vp(), vp()
        chk(L, L.eigd_malloc(ctx, 8 * n * k, C.byref(dB)))
        chk(L, L.eigd_malloc(ctx, 8 * n * k, C.byref(dX)))
        chk(L, L.eigd_h2d(ctx, dB, hp(Bh), 8 * n * k))
        bufs.append((dB, dX))
    res = {(i, j): [] for i in range(len(builds)) for j in range(nfac)}
    for rep in range(5):
        for j in range(nfac):
            for i, b in enumerate(builds):
                L, ctx, f = b[1], b[2], b[4][j]
                dB, dX = bufs[i]
                for _ in range(2):
                    chk(L, L.eigd_factor_solve_to(f, dB, k, dX, k, k, 1.0))
                chk(L, L.eigd_sync(ctx))
                t0 = time.perf_counter()
                for _ in range(20):
                    chk(L, L.eigd_factor_solve_to(f, dB, k, dX, k, k, 1.0))
                chk(L, L.eigd_sync(ctx))
                res[(i, j)].append((time.perf_counter() - t0) / 20 * 1e3)
    outs = []
    for i, b in enumerate(builds):
        Xh = np.empty((n, k))
        chk(b[1], b[1].eigd_d2h(b[2], hp(Xh), bufs[i][1], 8 * n * k))
        outs.append(Xh)
        meds = [float(np.median(res[(i, j)])) for j in range(nfac)]
        print(f"k={k:2d} {b[0]:40s} factors: " + " ".join(f"{m:.4f}" for m in meds) + f"  mean {np.mean(meds):.4f} ms", flush=True)
    print(f"k={k:2d} results bitwise equal: {all(np.array_equal(outs[0], o) for o in outs[1:])}", flush=True)
    for i, b in enumerate(builds):
        b[1].eigd_free(b[2], bufs[i][0])
        b[1].eigd_free(b[2], bufs[i][1])
