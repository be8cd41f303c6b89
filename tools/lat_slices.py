#!/usr/bin/env python3
"""K7L kernel time over random small reads of the benchmark data (HIP events): python tools/lat_slices.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, flacarray_amd as fa
from flacarray_amd import _lib
L = _lib.lib()
n_ch, n = 256, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
comp, st, nb = fa.encode_flac_device(x, level=5, compact=True)
ix = fa.DeviceDecodeIndex(comp, st, nb, n)
ch, first, cnt = bench.slice_requests(n_ch, n, 600)
L.fa_profile_enable(1)
ks = []
for i in range(600):
    ix.decode_slices(ch[i : i + 1], first[i : i + 1], cnt[i : i + 1])
    ms = (ctypes.c_float * 3)(); L.fa_profile_last(ms)
    ks.append(ms[2] * 1e3)
ks = np.array(ks[50:])
print(os.path.basename(_lib.LIB_PATH), f"mean {ks.mean():.1f} median {np.median(ks):.1f} p90 {np.percentile(ks, 90):.1f} max {ks.max():.1f} us")
