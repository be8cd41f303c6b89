#!/usr/bin/env python3
"""K3G (the placing encoder: every geometry K3F does not take, in a single pass) against the slot sequence it retires
(K3 + K4 + K5, forced with FLACARRAY_HIP_SLOTS) on one MI355X: device-resident encode, median of the repeats."""
import json
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


def timed(x, level, reps):
    fa.encode_flac_device(x, level=level, workspace=ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fa.encode_flac_device(x, level=level, workspace=ws)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del out
    return float(np.median(ts))


big = bench.make_data(torch, 1024, 1 << 20, 5, dev)
cases = [
    ("int32 (12, 1000) level 5  [tests/bindings.py shape]", big[:12, :1000].contiguous(), 5, 50),
    ("int32 (1, 10000) level 5  [cookbook stream]", big[:1, :10000].contiguous(), 5, 50),
    ("int32 (1000, 10000) level 5", big[:1000, :10000].contiguous(), 5, 20),
    ("int32 (4, 8192) level 5  [whole frames]", big[:4, :8192].contiguous(), 5, 50),
    ("int32 (64, 65536) level 5  [whole frames, 1024 of them]", big[:64, :65536].contiguous(), 5, 20),
    ("int32 (256, 65536) level 5  [whole frames, 4096 of them]", big[:256, :65536].contiguous(), 5, 20),
    ("int32 (1024, 65536) level 5  [whole frames, 16384 of them]", big[:1024, :65536].contiguous(), 5, 20),
    ("int32 (1024, 65532) level 5  [16 frames + tail per stream]", big[:1024, :65532].contiguous(), 5, 20),
    ("int32 (1024, 2^20) level 1  [1152-sample blocks]", big, 1, 5),
    ("int32 (1024, 2^20 - 3) level 5  [odd length]", big[:, : (1 << 20) - 3].contiguous(), 5, 5),
    ("int64 (12, 1000) level 5", (big[:12, :1000].to(torch.int64) << 13).contiguous(), 5, 50),
    ("int64 (1024, 2^20) level 5  [tools/bench_i64.py workload]",
     big.to(torch.int64) * 8192 + torch.randint(-4096, 4096, big.shape, device=dev, dtype=torch.int64), 5, 5),
]
for name, x, level, reps in cases:
    t_new = timed(x, level, reps)
    os.environ["FLACARRAY_HIP_SLOTS"] = "1"
    try:
        t_old = timed(x, level, reps)
    finally:
        del os.environ["FLACARRAY_HIP_SLOTS"]
    print(json.dumps({"case": name, "single_pass_ms": round(t_new * 1e3, 3), "slot_sequence_ms": round(t_old * 1e3, 3),
                      "ratio": round(t_old / t_new, 2), "Msamples_per_s": round(x.numel() / t_new / 1e6, 1)}), flush=True)
    del x
