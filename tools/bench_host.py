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
for rep in range(2):
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
