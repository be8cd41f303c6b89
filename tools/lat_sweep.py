#!/usr/bin/env python3
"""K7L time per whole-frame read against the amplitude of the data (Rice parameter) and the partition order the
encoder chose: python tools/lat_sweep.py   (FLACARRAY_HIP_LIB selects a variant build)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import flacarray_amd as fa  # noqa: E402
from flacarray_amd import _lib  # noqa: E402

L = _lib.lib()
n_ch, n = 4, 1 << 18
rng = np.random.default_rng(11)
t = np.arange(n)
rows = []
for bits in (1, 4, 8, 12, 16, 20):
    for burst in (False, True):
        sig = rng.normal(0, 2.0**bits, (n_ch, n))
        if burst:  # changing variance: the encoder picks higher partition orders
            sig *= (1.0 + 7.0 * (np.sin(2 * np.pi * t / 700.0) > 0.6))[None, :]
        x = np.rint(sig + 2.0 ** (bits + 3) * np.sin(2 * np.pi * 3 * t / n)).astype(np.int32)
        xd = torch.from_numpy(x).cuda()
        comp, st, nb, info = fa.encode_flac_device(xd, level=5, return_info=True, compact=True)
        info = info.cpu().numpy().reshape(n_ch, -1, 8)
        ix = fa.DeviceDecodeIndex(comp, st, nb, n)
        L.fa_profile_enable(1)
        ks = []
        for ch in range(n_ch):
            for f in range(0, info.shape[1], 4):
                out, _ = ix.decode_slices([ch], [f * 4096], [4096])
                ms = (ctypes.c_float * 3)()
                L.fa_profile_last(ms)
                ks.append(ms[2] * 1e3)
        assert torch.equal(out, xd[n_ch - 1, f * 4096 : f * 4096 + 4096])
        L.fa_profile_enable(0)
        ix.close()
        po = info[:, :, 2].reshape(-1)
        rows.append(f"noise 2^{bits:<2d} {'bursts' if burst else 'steady'}: partition order median {int(np.median(po))} (max {int(po.max())}), "
                    f"K7L {np.median(ks):6.1f} us per frame (p90 {np.percentile(ks, 90):6.1f})")
print(os.path.basename(_lib.LIB_PATH))
print("\n".join(rows))
