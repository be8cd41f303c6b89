#!/usr/bin/env python3
"""Latency of small reads from a resident store, the reference's usage pattern (array.py:409-449: one decode call per
key): FlacArray.__getitem__ of one (channel, sample range) at a time, and read_slices batches of growing size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    import flacarray_amd as fa

    n_ch, n = 1024, 1 << 20
    x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
    store = fa.FlacArray.from_device_array(x)
    ch, first, cnt = bench.slice_requests(n_ch, n, 2000)
    xs = x.cpu().numpy()
    for i in range(5):  # warm up + check
        assert np.array_equal(store[int(ch[i]), int(first[i]) : int(first[i] + cnt[i])], xs[ch[i], first[i] : first[i] + cnt[i]])
    for _ in range(3):  # third of three passes: the first opens the size classes of the pinned result pool (~10 ms each,
        # once), and a process' first seconds run at lower clocks (the same loop measures 160 us early and 121 us late)
        t0 = time.perf_counter()
        for i in range(1000):
            store[int(ch[i]), int(first[i]) : int(first[i] + cnt[i])]
        dt = (time.perf_counter() - t0) / 1000
    print(f"__getitem__ one slice per call: {dt*1e6:8.1f} us/call  {1/dt:9.0f} reads/s")
    for nb in (1, 10, 100, 1000):
        reps = max(2000 // nb, 3)
        t0 = time.perf_counter()
        for r in range(reps):
            store.read_slices(ch[:nb], first[:nb], cnt[:nb])
        dt = (time.perf_counter() - t0) / reps
        print(f"read_slices batch of {nb:5d}: {dt*1e6:9.1f} us/call  {nb/dt:10.0f} slices/s")
        t0 = time.perf_counter()
        for r in range(reps):
            store.read_slices(ch[:nb], first[:nb], cnt[:nb], as_tensor=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"   ... left on the device     : {dt*1e6:9.1f} us/call  {nb/dt:10.0f} slices/s")


if __name__ == "__main__":
    main()
