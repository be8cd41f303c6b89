#!/usr/bin/env python3
"""Single small reads from a resident int64 store (two-channel frames): the latency decoder (default) against the
throughput decoder (FLACARRAY_HIP_LATENCY=0).  python tools/bench_reads_i64.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402

n_ch, n = 64, 1 << 20
rng = np.random.default_rng(5)
t = np.arange(n)
x = np.rint(2.0**40 * np.sin(2 * np.pi * 5 * t / n)[None, :] * rng.random((n_ch, 1)) + rng.normal(0, 2.0**20, (n_ch, n))).astype(np.int64)
store = fa.FlacArray.from_device_array(torch.from_numpy(x).cuda())
ch, first, cnt = bench.slice_requests(n_ch, n, 600)
for mode in ("auto", "0"):
    if mode == "0":
        os.environ["FLACARRAY_HIP_LATENCY"] = "0"
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(300):
            y = store[int(ch[i]), int(first[i]) : int(first[i] + cnt[i])]
        dt = (time.perf_counter() - t0) / 300
    assert np.array_equal(y, x[ch[299], first[299] : first[299] + cnt[299]])
    t0 = time.perf_counter()
    for r in range(10):
        store.read_slices(ch[:100], first[:100], cnt[:100])
    d100 = (time.perf_counter() - t0) / 10
    print(f"int64 store, decoder {'K7L where it applies' if mode == 'auto' else 'K7 only'}: single read {dt * 1e6:.1f} us, batch of 100 {d100 * 1e6:.1f} us "
          f"({100 / d100:.0f} slices/s), {store.nbytes / x.nbytes:.3f} of the raw size")
import ctypes
from flacarray_amd import _lib
L = _lib.lib()
os.environ.pop("FLACARRAY_HIP_LATENCY", None)
L.fa_profile_enable(1)
for nb in (1, 10, 100):
    ks = []
    for r in range(5):
        store.read_slices(ch[r * nb : r * nb + nb], first[r * nb : r * nb + nb], cnt[r * nb : r * nb + nb], as_tensor=True)
        ms = (ctypes.c_float * 3)()
        L.fa_profile_last(ms)
        ks.append(ms[2] * 1e3)
    print(f"K7L kernel time, batch of {nb}: {np.median(ks):.1f} us")
