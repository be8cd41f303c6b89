#!/usr/bin/env python3
"""Where K3F's sequence overtakes K3G's on whole-frame int32 arrays: encode_flac_device of n x 65536 (16 frames per stream)
with FLACARRAY_HIP_PLACED_BELOW=0 (always K3F) and =10^9 (always K3G), median ms per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import flacarray_amd as fa
from flacarray_amd.libflacarray import EncodeWorkspace

dev = torch.device("cuda", 0)
ws = EncodeWorkspace()
big = bench.make_data(torch, 1024, 65536, 5, dev)


def timed(x, reps=30):
    ts = []
    for i in range(reps + 2):
        t0 = time.perf_counter()
        out = fa.encode_flac_device(x, level=5, workspace=ws)
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(time.perf_counter() - t0)
        del out
    return float(np.median(ts)) * 1e3


tails = bench.make_data(torch, 2048, 10000, 5, dev)  # streams of 2 full frames + a short one
for n in (256, 512, 768, 1024, 1365, 2048):
    x = tails[:n].contiguous()
    os.environ["FLACARRAY_HIP_PLACED_BELOW"] = "0"
    t_f = timed(x)
    os.environ["FLACARRAY_HIP_PLACED_BELOW"] = "1000000000"
    t_g = timed(x)
    del os.environ["FLACARRAY_HIP_PLACED_BELOW"]
    print(f"{n * 3:6d} frames ({n} x 10000, short last frames): K3F + detour {t_f:.3f} ms, K3G {t_g:.3f} ms", flush=True)
for n in (32, 64, 96, 128, 160, 192, 256, 384, 512):
    x = big[:n].contiguous()
    os.environ["FLACARRAY_HIP_PLACED_BELOW"] = "0"
    t_f = timed(x)
    os.environ["FLACARRAY_HIP_PLACED_BELOW"] = "1000000000"
    t_g = timed(x)
    del os.environ["FLACARRAY_HIP_PLACED_BELOW"]
    print(f"{n * 16:6d} frames: K3F {t_f:.3f} ms, K3G {t_g:.3f} ms", flush=True)
