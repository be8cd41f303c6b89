#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped host API (numpy in -> numpy out through encode_i32 / decode_i32)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flacarray_amd as fa
from tests.conftest import sinusoid_noise_i32

n_ch, n = int(os.environ.get("KB_CH", "512")), 1 << 20
x = np.tile(sinusoid_noise_i32(64, n, seed=3), (n_ch // 64, 1))
fa.encode_flac(x[:8], 5)  # warm-up (library load, tables)
res = {}
comp = y = None
for rep in range(2):
    comp = y = None  # release the previous outputs outside the timed region (munmap of GBs is not free)
    t0 = time.perf_counter()
    comp, st, nb = fa.encode_flac(x, 5)
    t1 = time.perf_counter()
    y = fa.decode_flac(np.asarray(comp), st, nb, n)
    t2 = time.perf_counter()
    res = {"channels": n_ch, "encode_s": round(t1 - t0, 3), "decode_s": round(t2 - t1, 3),
           "encode_Msamples_per_s": round(x.size / (t1 - t0) / 1e6, 1), "decode_Msamples_per_s": round(x.size / (t2 - t1) / 1e6, 1),
           "roundtrip_Msamples_per_s": round(x.size / (t2 - t0) / 1e6, 1), "encode_GBps_in": round(x.nbytes / (t1 - t0) / 1e9, 2)}
assert np.array_equal(y, x)
print(json.dumps(res))
# where the time goes: the C entry point alone (no Python wrapper work), and a plain copy of the same size
import ctypes
from flacarray_amd import _lib
L = _lib.lib()
flat = np.ascontiguousarray(x).reshape(-1)
starts = np.empty(n_ch, dtype=np.int64)
for rep in range(2):
    nbv, raw = ctypes.c_int64(0), ctypes.c_void_p(None)
    t0 = time.perf_counter()
    err = L.encode_i32(flat.ctypes.data, n_ch, n, 5, ctypes.byref(nbv), starts.ctypes.data, ctypes.byref(raw))
    t1 = time.perf_counter()
    _lib.libc_free(raw.value)
print("C encode_i32 alone: %.3f s (err %d), %d output bytes" % (t1 - t0, err, nbv.value))
t0 = time.perf_counter(); z = flat.copy(); print("numpy copy of the input: %.3f s" % (time.perf_counter() - t0))
