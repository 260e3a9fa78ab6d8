import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import eigd_amd as eg
from eigd_amd.device import default_context
ctx = default_context()
n, k = 998284, 32
a = np.random.default_rng(0).uniform(size=(n, k))
for rep in range(5):
    np.multiply(a, 1.0 + 1e-9, out=a)
    ctx.sync(); t0 = time.perf_counter()
    b = ctx.twin_upload(a)
    ctx.sync(); t1 = time.perf_counter()
    h = b.get()
    ctx.sync(); t2 = time.perf_counter()
    ctx.twin_adopt(h, b)
    t3 = time.perf_counter()
    b2 = ctx.twin_upload(h)
    t4 = time.perf_counter()
    print(f"rep {rep}: twin_upload(new content) {1e3*(t1-t0):.2f} ms, get {1e3*(t2-t1):.2f} ms, adopt {1e3*(t3-t2):.2f} ms, lookup {1e3*(t4-t3):.2f} ms same={b2 is b}")
