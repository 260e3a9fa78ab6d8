#!/usr/bin/env python3
"""development aid: does page-locking still work after an Arnoldi-form solve (bench: the numpy leg behind the Arnoldi leg
ran its transfers at the pageable rate)"""
import ctypes as C
import os
import resource
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigd_amd as eg  # noqa: E402
from eigd_amd import _ffi  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

warnings.simplefilter("ignore")
print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK))
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
u = col.full_vector(eg.SpLuOperator(K, check_symmetry=False, coords=col.dof_coords())(col.f[col.reduced]))
G = col.geometric_stiffness(u)
sigma, N = 1.0971, 32
fac = eg.SpLuOperator((K + sigma * G).tocsr(), coords=col.dof_coords(), check_symmetry=False)
ctx = fac.ctx
s = eg.IRAM(N=N, m=65, mode="buckling")
s.solve(G, K, fac, sigma)
Phib = np.random.default_rng(1).uniform(size=(K.shape[0], N))


def try_pin(tag):
    h = C.c_void_p()
    try:
        _ffi.call("eigd_host_alloc", Phib.nbytes, C.byref(h))
        _ffi.lib().eigd_host_free(h)
        a = "alloc ok"
    except Exception as e:  # noqa: BLE001
        a = f"alloc FAILED {e}"
    x = np.empty_like(Phib)
    try:
        _ffi.call("eigd_host_register", C.c_void_p(x.ctypes.data), x.nbytes)
        _ffi.lib().eigd_host_unregister(C.c_void_p(x.ctypes.data))
        b = "register ok"
    except Exception as e:  # noqa: BLE001
        b = f"register FAILED {e}"
    print(tag, a, b, "free device GB %.1f" % (ctx.mem_info()[0] / 2**30), flush=True)


def abi_times(P):
    from eigd_amd import adjoint as _a
    from eigd_amd import device as _d
    ms, orig = {}, _ffi.call

    def timed(name, *a):
        t = time.perf_counter()
        try:
            return orig(name, *a)
        finally:
            ms[name] = ms.get(name, 0.0) + 1e3 * (time.perf_counter() - t)

    for m in (_ffi, _d, _a):
        if hasattr(m, "call"):
            m.call = timed
    try:
        np.multiply(P, 1.0 + 1e-9, out=P)
        ctx.sync()
        s.solve_adjoint(P, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        ctx.sync()
    finally:
        for m in (_ffi, _d, _a):
            if hasattr(m, "call"):
                m.call = orig
    print("    ", {k: round(v, 2) for k, v in sorted(ms.items(), key=lambda kv: -kv[1])[:7]}, flush=True)


def numpy_steps(tag):
    ts = []
    P = Phib.copy()
    for _ in range(5):
        np.multiply(P, 1.0 + 1e-9, out=P)
        ctx.sync()
        t0 = time.perf_counter()
        psi, data = s.solve_adjoint(P, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(tag, "numpy solve_adjoint ms", np.round(ts, 1), flush=True)
    abi_times(P)


try_pin("start:")
numpy_steps("before:")
eg.tuning.recurrence = "arnoldi"
d = ctx.from_host(Phib)
for _ in range(2):
    dpsi, data = s.solve_adjoint(d, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
eg.tuning.recurrence = "auto"
try_pin("after two Arnoldi solves:")
numpy_steps("after Arnoldi:")
da, ds = dpsi.get(), d.get()
del da, ds
ctx.release_workspaces()
try_pin("after release_workspaces:")
from eigd_amd import device as dev  # noqa: E402
P = Phib.copy()
for it in range(6):
    np.multiply(P, 1.0 + 1e-9, out=P)
    ctx.sync()
    t0 = time.perf_counter()
    psi, data = s.solve_adjoint(P, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    ctx.sync()
    print("call", it, "%.1f ms" % (1e3 * (time.perf_counter() - t0)), "P registered:", P.ctypes.data in dev._pinned.registered,
          "seen:", [(k == id(P), v[1]) for k, v in dev._pinned.seen.items()], "psi pinned-pool sizes:",
          {k: len(v) for k, v in dev._pinned.free.items()}, "asked:", dict(dev._pinned.asked), flush=True)
abi_times(P)
