#!/usr/bin/env python3
"""Kernel micro-bench: encode+decode N channels x 1Msamp, per-kernel ms, parity spot check.

FLACARRAY_HIP_LIB selects the .so (e.g. the -DFA_STAMPS diagnostic build, whose per-phase
cycle shares of encode_frames_kernel are then printed)."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["P0 load/wasted", "P2 fixed loop", "P2 reduce", "(polls per frame)", "P3 autocorr loop", "P3 butterfly", "P3 levinson+quant",
          "P4 lpc residual", "P4 rice search", "emit prep", "preamble", "rows", "tail", "tail: offset wait", "(flush: offset wait)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch

    import bench
    import flacarray_amd as fa
    from flacarray_amd import _lib
    from flacarray_amd.libflacarray import EncodeWorkspace

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    x = bench.make_data(torch, args.channels, args.samples, 123456789, dev)
    ws = EncodeWorkspace()
    L.fa_profile_enable(1)
    res = []
    for r in range(args.reps + 1):
        comp, st, nb = fa.encode_flac_device(x, level=args.level, workspace=ws)
        y = fa.decode_flac_device(comp, st, nb, args.samples)
        ms = (ctypes.c_float * 3)()
        L.fa_profile_last(ms)
        if r > 0:
            res.append(list(ms))
    assert torch.equal(x, y)
    res = np.array(res)
    n = x.numel()
    c = comp.numel() / n
    enc, cmp_, dec = np.median(res, axis=0)
    print(f"lib={os.path.basename(_lib.LIB_PATH)} ch={args.channels} c={c:.4f} B/sample")
    print(f"encode_frames {enc:8.3f} ms  {(4+c)*n/enc/1e6:8.1f} GB/s   compact {cmp_:7.3f} ms   decode_frames {dec:8.3f} ms  {(4+c)*n/dec/1e6:8.1f} GB/s"
          f"   (median of {len(res)}; encode min {res[:, 0].min():.3f} max {res[:, 0].max():.3f})")
    if hasattr(L, "fa_debug_stamps"):
        buf = (ctypes.c_ulonglong * 64)()
        L.fa_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        L.fa_debug_stamps(buf, 1)
        tot = sum(buf[: len(PHASES)])
        nfr = max(buf[16], 1)
        print(f"  sampled frames {nfr}, {tot/nfr:.0f} stamped cycles/frame")
        for i, name in enumerate(PHASES):
            print(f"  {name:22s} {buf[i]/max(tot,1)*100:6.2f} %   {buf[i]/nfr:10.0f} cyc/frame")
        for x in range(8):
            n_ = max(buf[32 + 4 * x], 1)
            print(f"  XCD {x}: frames {buf[32 + 4 * x]:6d}  offset wait {buf[33 + 4 * x] / n_:8.0f}  lifetime {buf[34 + 4 * x] / n_:8.0f}  start->publish {buf[35 + 4 * x] / n_:8.0f}")
        print(f"  (of rows: flush calls   {buf[15]/max(tot,1)*100:6.2f} %   {buf[15]/nfr:10.0f} cyc/frame)")
    if args.check:
        from oracle import oracle as O

        xs = x[:4].cpu().numpy()
        bo, so, no = O.encode_i32(xs, args.level)
        cg, sg, ng = fa.encode_flac_device(x[:4].contiguous(), level=args.level)
        print("parity vs oracle:", np.array_equal(cg.cpu().numpy(), bo))


if __name__ == "__main__":
    main()
