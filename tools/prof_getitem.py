#!/usr/bin/env python3
"""Where FlacArray.__getitem__ spends its time around DeviceDecodeIndex.decode_slices (one MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, flacarray_amd as fa
n_ch = int(os.environ.get("N_CH", "1024")); n = 1 << 20
x = bench.make_data(torch, n_ch, n, 7, torch.device("cuda", 0))
store = fa.FlacArray.from_device_array(x)
ch, first, cnt = bench.slice_requests(n_ch, n, 2000)
ix = store._index()
acc = {"plan": 0.0, "sel": 0.0, "dec": 0.0, "rest": 0.0}
def one(i):
    t0 = time.perf_counter()
    shape, keep, f, l = store._plan_selection((int(ch[i]), slice(int(first[i]), int(first[i] + cnt[i]))))
    t1 = time.perf_counter()
    sel = np.flatnonzero(np.asarray(keep).reshape(-1))
    a1 = np.full(sel.size, f, np.int64); a2 = np.full(sel.size, l - f, np.int64)
    t2 = time.perf_counter()
    flat, _ = ix.decode_slices(sel, a1, a2, to_host=True)
    t3 = time.perf_counter()
    out = flat.reshape(sel.size, l - f).reshape(shape)
    t4 = time.perf_counter()
    acc["plan"] += t1 - t0; acc["sel"] += t2 - t1; acc["dec"] += t3 - t2; acc["rest"] += t4 - t3
for rep in range(2):
    for k in acc: acc[k] = 0.0
    for i in range(1000): one(i)
print("n_ch", n_ch, {k: round(v * 1e3, 1) for k, v in acc.items()}, "us per call")
t0 = time.perf_counter()
for i in range(1000): store[int(ch[i]), int(first[i]) : int(first[i] + cnt[i])]
print("store[...] us", (time.perf_counter() - t0) / 1000 * 1e6)
t0 = time.perf_counter()
for i in range(1000): ix.decode_slices(ch[i:i+1], first[i:i+1], cnt[i:i+1], to_host=True)
print("decode_slices us", (time.perf_counter() - t0) / 1000 * 1e6)
