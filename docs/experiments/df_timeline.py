#!/usr/bin/env python3
"""timeline of the two dataflow launches of a sweep on the C3 factor, per tree level: when the level's tasks were taken,
how long they waited, when they finished (microseconds from the launch's first stamp).  Usage: df_timeline.py [k ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigd_amd.device import Factor, default_context  # noqa: E402
from eigd_amd.problems import BucklingColumn  # noqa: E402

ctx = default_context()
col = BucklingColumn(706, 706, seed=0)
K = col.stiffness()
F = Factor(ctx, K, coords=col.dof_coords())
rng = np.random.default_rng(0)
for k in ([int(a) for a in sys.argv[1:]] or [4, 32]):
    B = ctx.from_host(rng.normal(size=(K.shape[0], k)))
    for _ in range(3):
        F.solve_inplace(B)
    st, lev, nf = F.sweep_trace(B)
    st = st.astype(np.float64) / 100.0  # microseconds
    for name, sl in (("forward", slice(0, nf)), ("backward", slice(nf, None))):
        s, l = st[sl], lev[sl]
        t0 = s[:, 0].min()
        fin = np.where(s[:, 3] > 0, s[:, 3], s[:, 2])   # (groups of a split chain that are not the last leave early)
        print(f"k={k} {name}: {len(s)} tasks, span {fin.max() - t0:.1f} us")
        print("  level tasks   taken(first..last)    wait-end(first..last)   done(first..last)   mean wait   mean work after wait")
        for lv in (sorted(set(l)) if name == "forward" else sorted(set(l), reverse=True)):
            m = l == lv
            a = s[m]
            w = a[:, 1] > 0
            waited = (a[w, 2] - a[w, 1]).mean() if w.any() else 0.0
            fz = fin[m]
            done = fz[fz > 0]
            work = (a[w & (a[:, 3] > 0), 3] - a[w & (a[:, 3] > 0), 2]).mean() if (w & (a[:, 3] > 0)).any() else 0.0
            we = a[w, 2] - t0 if w.any() else np.zeros(1)
            extra = ""
            if (a[:, 4] > 0).any():  # strip tasks: taken -> staged x, behind the wait -> v1 complete, strips, signal
                q = a[a[:, 4] > 0]
                extra = (f"   pre {np.mean(q[:, 1] - q[:, 0]):5.2f} planes {np.mean(q[:, 4] - q[:, 2]):5.2f} strips {np.mean(q[:, 5] - q[:, 4]):5.2f}"
                         f" signal {np.mean(q[:, 3] - q[:, 5]):5.2f}")
            print(f"  {lv:5d} {m.sum():5d}   {a[:, 0].min() - t0:8.1f} ..{a[:, 0].max() - t0:8.1f}    {we.min():8.1f} ..{we.max():8.1f}"
                  f"     {done.min() - t0:8.1f} ..{done.max() - t0:8.1f}   {waited:8.2f}   {work:8.2f}{extra}")
F.sweep_check()
