import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, flacarray_amd as fa
from flacarray_amd.libflacarray import EncodeWorkspace
dev = torch.device("cuda", 0); ws = EncodeWorkspace()
x = bench.make_data(torch, 1024, 1 << 20, 5, dev)
for lvl in (1, 0):
    fa.encode_flac_device(x, level=lvl, workspace=ws); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o = fa.encode_flac_device(x, level=lvl, workspace=ws); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); del o
    print(f"level {lvl}: {np.median(ts)*1e3:.3f} ms")
