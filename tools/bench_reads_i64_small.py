#!/usr/bin/env python3
"""Single small reads from a resident int64 store of SMALL values of both signs (the encoder writes side + right frames
for these): the latency decoder (default) against the throughput decoder.  python tools/bench_reads_i64_small.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402

n_ch, n = 64, 1 << 20
rng = np.random.default_rng(6)
x = np.rint(rng.normal(0, 8.0, (n_ch, n)) + 20 * np.sin(2 * np.pi * 7 * np.arange(n) / n)[None, :]).astype(np.int64)
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
    print(f"small-valued int64 store, decoder {'K7L where it applies' if mode == 'auto' else 'K7 only'}: single read {dt * 1e6:.1f} us, "
          f"{store.nbytes / x.nbytes:.3f} of the raw size")
