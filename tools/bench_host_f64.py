#!/usr/bin/env python3
"""array_compress on float64 input (the reference's default dtype): one trip over PCIe (fa_encode_f64_host) against the
two-call form it replaces (float_to_int: up, int64 down; encode_flac: int64 up, bytes down).  python tools/bench_host_f64.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import flacarray_amd as fa  # noqa: E402

rng = np.random.default_rng(3)
n_ch, n = 1000, 100000
t = np.arange(n)
x = (rng.random((n_ch, 1)) * 6 * np.sin(2 * np.pi * 5 * t / n) + rng.normal(0, 1, (n_ch, n))).astype(np.float64)
fa.array_compress(x[:8], precision=10)
for name, fn in (("one trip (array_compress)", lambda: fa.array_compress(x, precision=10)),
                 ("two calls (float_to_int + encode_flac)", lambda: (lambda r: fa.encode_flac(r[0], 5) + (r[1], r[2]))(fa.float_to_int(x, precision=10)))):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name}: {best * 1e3:.1f} ms for {x.nbytes / 1e6:.0f} MB of float64 ({x.size / best / 1e6:.0f} Msamples/s), {out[0].nbytes / x.nbytes:.3f} of the raw size")
comp, st, nb, off, gain = fa.array_compress(x, precision=10)
for name, fn in (("one trip (array_decompress)", lambda: fa.array_decompress(comp, n, st, nb, stream_offsets=off, stream_gains=gain, is_int64=True)),
                 ("two calls (decode_flac + int_to_float)", lambda: fa.int_to_float(fa.decode_flac(comp, st, nb, n, is_int64=True), off, gain))):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name}: {best * 1e3:.1f} ms back to {out.nbytes / 1e6:.0f} MB of float64 ({out.size / best / 1e6:.0f} Msamples/s)")
