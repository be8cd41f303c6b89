#!/usr/bin/env python3
"""int64 / float64 arrays (two-channel streams) on one MI355X: encode + decode of n_ch x 1Msamp."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import flacarray_amd as fa
from flacarray_amd.libflacarray import EncodeWorkspace

n_ch, n = int(os.environ.get("KB_CH", "1024")), 1 << 20
dev = torch.device("cuda", 0)
# 45-bit signals: the int32 workload scaled by 2^13 plus low-order noise, so the low word is busy
x32 = bench.make_data(torch, n_ch, n, 5, dev)
x = x32.to(torch.int64) * 8192 + torch.randint(-4096, 4096, (n_ch, n), device=dev, dtype=torch.int64)
del x32
ws = EncodeWorkspace()


def step():
    comp, st, nb = fa.encode_flac_device(x, level=5, workspace=ws)
    y = fa.decode_flac_device(comp, st, nb, n, is_int64=True)
    return comp, y


step()
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    comp = y = None
    comp, y = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
assert torch.equal(y, x)
t1 = time.perf_counter()
comp2, st, nb = fa.encode_flac_device(x, level=5, workspace=ws)
torch.cuda.synchronize()
t2 = time.perf_counter()
y = fa.decode_flac_device(comp2, st, nb, n, is_int64=True)
torch.cuda.synchronize()
t3 = time.perf_counter()
print(json.dumps({"int64_Msamples_per_s": round(n_ch * n / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 2),
                  "encode_ms": round((t2 - t1) * 1e3, 2), "decode_ms": round((t3 - t2) * 1e3, 2),
                  "bytes_per_sample": round(comp.numel() / (n_ch * n), 4), "channels": n_ch}))
