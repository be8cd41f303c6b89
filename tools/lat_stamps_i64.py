#!/usr/bin/env python3
"""K7L phase stamps for two-channel frames (small-valued int64: side + right frames).  Needs the diagnostic build without
-DFA_DEV_MINIMAL (the two-channel kernels):  python -m flacarray_amd.build --variant stl64 -DFA_LAT_STAMPS=1"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, flacarray_amd as fa
n_ch, n = 8, 1 << 18
rng = np.random.default_rng(6)
x = np.rint(rng.normal(0, 8.0, (n_ch, n)) + 20 * np.sin(2 * np.pi * 7 * np.arange(n) / n)[None, :]).astype(np.int64)
store = fa.FlacArray.from_device_array(torch.from_numpy(x).cuda())
for f in (8, 12, 16):
    y = store[1, f * 4096 : f * 4096 + 4096]
    assert np.array_equal(y, x[1, f * 4096 : f * 4096 + 4096])
    y = store[1, f * 4096 + 100 : f * 4096 + 600]
