#!/usr/bin/env python3
"""Diagnostic: which reads make the latency decoder (K7L) give up?  FLACARRAY_HIP_LATENCY_DEBUG=1 python tools/lat_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, flacarray_amd as fa
n_ch, n = 8, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
comp, st, nb, info = fa.encode_flac_device(x, level=5, return_info=True, compact=True)
info = info.cpu().numpy().reshape(n_ch, 256, 8)
ix = fa.DeviceDecodeIndex(comp, st, nb, n)
xs = x.cpu().numpy()
rng = np.random.default_rng(3)
for it in range(600):
    ch = int(rng.integers(0, n_ch)); f = int(rng.integers(0, 254)); a = int(rng.integers(0, 4096)); c = int(rng.integers(1, 8193))
    sys.stderr.write(f"Q ch {ch} f {f} a {a} c {c} type {info[ch,f,0]} order {info[ch,f,1]} bytes {info[ch,f,6]}\n")
    out, _ = ix.decode_slices([ch], [f * 4096 + a], [c])
    assert np.array_equal(out.cpu().numpy(), xs[ch, f * 4096 + a : f * 4096 + a + c])
print("done")
