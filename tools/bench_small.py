#!/usr/bin/env python3
"""Latency of the reference's own test shapes through the Python surface (numpy in, numpy out): tests/array.py:26-146,
tests/bindings.py:165-230.  python tools/bench_small.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import flacarray_amd as fa  # noqa: E402

rng = np.random.default_rng(1)
cases = [
    ("int32 (4,3,1000)", rng.integers(-2**20, 2**20, (4, 3, 1000)).astype(np.int32), {}),
    ("int32 (10000,)", rng.integers(-2**20, 2**20, (10000,)).astype(np.int32), {}),
    ("int64 (4,3,1000)", rng.integers(-2**40, 2**40, (4, 3, 1000)).astype(np.int64), {}),
    ("float32 (4,3,1000) quanta 1e-5", rng.normal(0, 1, (4, 3, 1000)).astype(np.float32), {"quanta": 1e-5}),
    ("float64 (4,3,10000) precision 10", rng.normal(0, 1, (4, 3, 10000)), {"precision": 10}),
    ("float32 (1000,100000) quanta 1e-7", rng.normal(0, 1, (1000, 100000)).astype(np.float32), {"quanta": 1e-7}),
]
for name, x, kw in cases:
    f = fa.FlacArray.from_array(x, **kw)
    f.to_array()
    tc, td = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        f = fa.FlacArray.from_array(x, **kw)
        t1 = time.perf_counter()
        y = f.to_array()
        t2 = time.perf_counter()
        tc.append(t1 - t0)
        td.append(t2 - t1)
    print(f"{name:36s} from_array {min(tc) * 1e3:8.3f} ms   to_array {min(td) * 1e3:8.3f} ms   ({f.nbytes / x.nbytes:.3f} of the raw size)")
# small reads from a host-resident store (the reference's usage: array.py:409-449) and from the same store resident in HBM
x = rng.normal(0, 1, (1000, 100000)).astype(np.float32)
f = fa.FlacArray.from_array(x, quanta=1e-7)
keep = np.zeros(1000, dtype=bool)
keep[500] = True
for name, fn in (("host store: f[500, 1000:2000]", lambda: f[500, 1000:2000]), ("host store: f[500]  (whole stream)", lambda: f[500]),
                 ("host store: to_array(keep=one stream)", lambda: f.to_array(keep=keep))):
    fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        y = fn()
        ts.append(time.perf_counter() - t0)
    print(f"{name:44s} {np.median(ts) * 1e3:8.3f} ms")
f.to_device()
for name, fn in (("resident store: f[500, 1000:2000]", lambda: f[500, 1000:2000]), ("resident store: f[500]  (whole stream)", lambda: f[500])):
    fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        y = fn()
        ts.append(time.perf_counter() - t0)
    print(f"{name:44s} {np.median(ts) * 1e3:8.3f} ms")
