#!/usr/bin/env python3
"""K7L kernel time on whole-frame single reads (in-library HIP events): python tools/lat_time.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, flacarray_amd as fa
from flacarray_amd import _lib
L = _lib.lib()
n_ch, n = 8, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
comp, st, nb, info = fa.encode_flac_device(x, level=5, return_info=True, compact=True)
info = info.cpu().numpy().reshape(n_ch, 256, 8)
ix = fa.DeviceDecodeIndex(comp, st, nb, n)
L.fa_profile_enable(1)
by = {}
for ch in range(n_ch):
    for f in range(0, 256, 4):
        out, _ = ix.decode_slices([ch], [f * 4096], [4096])
        ms = (ctypes.c_float * 3)(); L.fa_profile_last(ms)
        by.setdefault((int(info[ch, f, 0]), int(info[ch, f, 1])), []).append(ms[2] * 1e3)
print(os.path.basename(_lib.LIB_PATH), {k: round(float(np.median(v)), 1) for k, v in sorted(by.items())}, "us (median K7L time per whole-frame read, by (type, order))")
