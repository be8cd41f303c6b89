#!/usr/bin/env python3
"""Where the latency decoder (K7L, wave per frame) stops paying against the throughput decoder (K7, lane per frame):
batches of scattered slices left on the device, FLACARRAY_HIP_LATENCY=1 (K7L up to 65535 frames) against =0 (K7)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402

n_ch, n = 1024, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
store = fa.FlacArray.from_device_array(x)
ch, first, cnt = bench.slice_requests(n_ch, n, 20000)
for nb in (500, 1000, 2000, 4000, 8000, 16000):
    row = []
    for mode in ("1", "0"):
        os.environ["FLACARRAY_HIP_LATENCY"] = mode
        store.read_slices(ch[:nb], first[:nb], cnt[:nb], as_tensor=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            store.read_slices(ch[:nb], first[:nb], cnt[:nb], as_tensor=True)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 5 * 1e6)
    print(f"{nb:6d} slices: K7L {row[0]:8.1f} us   K7 {row[1]:8.1f} us")
