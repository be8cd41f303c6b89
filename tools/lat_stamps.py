#!/usr/bin/env python3
"""Where a frame's time goes inside the latency decoder K7L, and how many parse rounds it took: needs the diagnostic build
    python -m flacarray_amd.build --variant stl -DFA_DEV_MINIMAL -DFA_LAT_STAMPS=1
    FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_stl.so python tools/lat_stamps.py
(lane 0 of task 0 prints its s_memrealtime stamps; profiles/r03_reads_stamps.log)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench, flacarray_amd as fa
n_ch, n = 8, 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
comp, st, nb = fa.encode_flac_device(x, level=5, compact=True)
ix = fa.DeviceDecodeIndex(comp, st, nb, n)
for f in (8, 12, 16, 20, 24, 28):
    out, _ = ix.decode_slices([1], [f * 4096], [4096])
    torch.cuda.synchronize()
ix.close()
