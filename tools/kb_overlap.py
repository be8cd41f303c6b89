#!/usr/bin/env python3
"""Experiment: does the compaction of one channel chunk (K5, HBM bound) hide under the frame
encode of the next chunk (K3, VALU bound) when the two run on different HIP streams?
Uses the two-phase C entry points as they are: finish(A) is asynchronous, begin(B) blocks."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

import bench  # noqa: E402
from flacarray_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
NCH = int(os.environ.get("KB_CH", "2048"))
NCHUNK = int(os.environ.get("KB_CHUNKS", "4"))
SS = 1 << 20
x = bench.make_data(torch, NCH, SS, 1, dev)
per = NCH // NCHUNK
chunks = [x[i * per:(i + 1) * per] for i in range(NCHUNK)]
wsb = L.fa_encode_workspace_bytes(per, SS, 5)
wss = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(NCHUNK)]
starts = [torch.empty(per, dtype=torch.int64, device=dev) for _ in range(NCHUNK)]
nbytes = [torch.empty(per, dtype=torch.int64, device=dev) for _ in range(NCHUNK)]
s1 = torch.cuda.Stream(device=dev)
s2 = torch.cuda.Stream(device=dev, priority=int(os.environ.get("KB_PRIO", "0")))


def p(t):
    return ctypes.c_void_p(t.data_ptr())


def begin(i, st):
    tot = ctypes.c_int64(0)
    rc = L.fa_encode_i32_device_begin(p(chunks[i]), per, SS, 5, p(wss[i]), wss[i].numel(), p(starts[i]), p(nbytes[i]),
                                      ctypes.byref(tot), None, ctypes.c_void_p(st.cuda_stream))
    assert rc == 0, rc
    return tot.value


def finish(i, out, st):
    rc = L.fa_encode_i32_device_finish(per, SS, 5, p(wss[i]), p(starts[i]), p(out), ctypes.c_void_p(st.cuda_stream))
    assert rc == 0, rc


tots = [begin(i, s1) for i in range(NCHUNK)]
outs = [torch.empty(t, dtype=torch.uint8, device=dev) for t in tots]
refs = []
for i in range(NCHUNK):
    finish(i, outs[i], s1)
torch.cuda.synchronize()
refs = [o.clone() for o in outs]


def run_seq():
    for i in range(NCHUNK):
        begin(i, s1)
        finish(i, outs[i], s1)
    torch.cuda.synchronize()


def run_ovl():
    begin(0, s1)
    for i in range(1, NCHUNK):
        finish(i - 1, outs[i - 1], s2)
        begin(i, s1)
    finish(NCHUNK - 1, outs[NCHUNK - 1], s2)
    torch.cuda.synchronize()


def run_whole():
    tot = ctypes.c_int64(0)
    rc = L.fa_encode_i32_device_begin(p(x), NCH, SS, 5, p(wsw), wsw.numel(), p(stw), p(nbw), ctypes.byref(tot), None,
                                      ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0
    rc = L.fa_encode_i32_device_finish(NCH, SS, 5, p(wsw), p(stw), p(outw), ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()


wsw = torch.empty(L.fa_encode_workspace_bytes(NCH, SS, 5), dtype=torch.uint8, device=dev)
stw = torch.empty(NCH, dtype=torch.int64, device=dev)
nbw = torch.empty(NCH, dtype=torch.int64, device=dev)
outw = torch.empty(sum(tots), dtype=torch.uint8, device=dev)

for name, fn in (("whole", run_whole), ("seq", run_seq), ("ovl", run_ovl), ("whole", run_whole), ("seq", run_seq), ("ovl", run_ovl)):
    fn()
    ts = []
    for r in range(5):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ok = all(torch.equal(a, b) for a, b in zip(outs, refs)) if name != "whole" else True
    print(f"{name}: chunks={NCHUNK} ch={NCH} min {min(ts):.3f} ms  med {sorted(ts)[2]:.3f} ms  identical={ok}", flush=True)
