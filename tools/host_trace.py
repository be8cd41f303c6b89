#!/usr/bin/env python3
"""Per-chunk timeline of the host-pointer encode / decode (FLACARRAY_HIP_HOST_TRACE=1 python tools/host_trace.py)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flacarray_amd as fa
from tests.conftest import sinusoid_noise_i32
n = 1 << 20
x = np.tile(sinusoid_noise_i32(64, n, seed=3), (8, 1))
fa.encode_flac(x[:8], 5)
for rep in range(2):
    print("--- encode", rep, file=sys.stderr); t0 = time.perf_counter()
    comp, st, nb = fa.encode_flac(x, 5)
    print("encode_flac %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
    c = np.asarray(comp)
    print("--- decode", rep, file=sys.stderr); t0 = time.perf_counter()
    y = fa.decode_flac(c, st, nb, n)
    print("decode_flac %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
    del comp, c, y
