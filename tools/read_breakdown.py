#!/usr/bin/env python3
"""Where a single small read spends its time: Python key handling / wrapper / the C call / the kernel.
   python tools/read_breakdown.py   (one MI355X)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402
from flacarray_amd import _lib  # noqa: E402
from flacarray_amd.libflacarray import _dp, _stream_ptr  # noqa: E402

L = _lib.lib()
n_ch, n = 256, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
store = fa.FlacArray.from_device_array(x)
ix = store._index()
ch, first, cnt = bench.slice_requests(n_ch, n, 400)
R = 300


def timed(fn):
    for i in range(R):  # warm pass (opens the size classes of the pinned result pool)
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R * 1e6


a = timed(lambda i: store[int(ch[i]), int(first[i]) : int(first[i] + cnt[i])])
b = timed(lambda i: ix.decode_slices(ch[i : i + 1], first[i : i + 1], cnt[i : i + 1], to_host=True))
# the bare C call: everything preallocated
out = torch.empty(16384, dtype=torch.int32, device="cuda")
host = np.empty(16384, dtype=np.int32)
zero = np.zeros(1, dtype=np.int64)
sp = _stream_ptr()


def bare(i):
    rc = L.fa_decode_indexed_host(ix._h, 1, ctypes.c_void_p(ch[i : i + 1].ctypes.data), ctypes.c_void_p(first[i : i + 1].ctypes.data),
                                  ctypes.c_void_p(cnt[i : i + 1].ctypes.data), ctypes.c_void_p(zero.ctypes.data), _dp(out), None, None, None,
                                  ctypes.c_void_p(host.ctypes.data), int(cnt[i]) * 4, sp, 0)
    assert rc == 0


ch = np.ascontiguousarray(ch, dtype=np.int64)
first = np.ascontiguousarray(first, dtype=np.int64)
cnt = np.ascontiguousarray(cnt, dtype=np.int64)
c = timed(bare)
L.fa_profile_enable(1)
ks = []
for i in range(R):
    bare(i)
    ms = (ctypes.c_float * 3)()
    L.fa_profile_last(ms)
    ks.append(ms[2] * 1e3)
L.fa_profile_enable(0)
print(f"store[ch, a:b] {a:.1f} us | DeviceDecodeIndex.decode_slices(to_host) {b:.1f} us | fa_decode_indexed_host alone {c:.1f} us | "
      f"K7L kernel (HIP events) median {np.median(ks):.1f} us, mean {np.mean(ks):.1f} us")
