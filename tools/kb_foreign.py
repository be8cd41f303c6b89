#!/usr/bin/env python3
"""Decode of streams WITHOUT a seek table (what libFLAC writes through the reference):
parallel sync-code scan vs. the serial per-stream walk (FLACARRAY_HIP_NO_SYNC_SCAN=1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import flacarray_amd as fa
from tests.conftest import strip_seektable

n_ch, n = int(os.environ.get("KB_CH", "512")), 1 << 20
dev = torch.device("cuda", 0)
x = bench.make_data(torch, n_ch, n, 7, dev)
comp, st, nb = fa.encode_flac_device(x, level=5)
b2, s2, n2 = strip_seektable(comp.cpu().numpy(), st.cpu().numpy(), nb.cpu().numpy())
tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (b2, s2, n2))


def timed(label):
    for r in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fa.decode_flac_device(tb, ts, tn, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.equal(y, x)
    print(f"{label:28s} {dt*1e3:9.2f} ms  {n_ch*n/dt/1e6:10.1f} Msamples/s")


torch.cuda.synchronize()
t0 = time.perf_counter()
y = fa.decode_flac_device(comp, st, nb, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
y = fa.decode_flac_device(comp, st, nb, n)
torch.cuda.synchronize()
print(f"{'own streams (seek table)':28s} {(time.perf_counter()-t0)*1e3:9.2f} ms")
timed("no seek table: sync scan")
os.environ["FLACARRAY_HIP_NO_SYNC_SCAN"] = "1"
timed("no seek table: serial walk")
