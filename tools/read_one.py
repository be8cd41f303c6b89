#!/usr/bin/env python3
"""200 single-slice reads from a resident store (for `rocprofv3 --kernel-trace --stats -- python tools/read_one.py`)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402

n_ch, n = 256, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
store = fa.FlacArray.from_device_array(x)
ch, first, cnt = bench.slice_requests(n_ch, n, 400)
for nb in (1, 64):
    store.read_slices(ch[:nb], first[:nb], cnt[:nb], as_tensor=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        store.read_slices(ch[i : i + nb], first[i : i + nb], cnt[i : i + nb], as_tensor=True)
    torch.cuda.synchronize()
    print(nb, "slices per call:", (time.perf_counter() - t0) / 200 * 1e6, "us per call")
