#!/usr/bin/env python3
"""K3G timing experiments: one encode workload per library variant inside one GPU session.
   python tools/kb_placed.py name ...   (name '' / shipped = the shipped library; variants: build.py --variant)
A -DFA_STAMPS variant also prints the share of each phase of K3G's loop (ticket, frame body, table load, wait for the
offset, placement)."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import time

    import numpy as np
    import torch

    import bench
    import flacarray_amd as fa
    from flacarray_amd.libflacarray import EncodeWorkspace

    dev = torch.device("cuda", 0)
    ws = EncodeWorkspace()
    x = bench.make_data(torch, 1024, 1 << 20, 5, dev)[:, : (1 << 20) - 3].contiguous()
    for label, env in (("single pass", None), ("slot sequence", "1")):
        if env:
            os.environ["FLACARRAY_HIP_SLOTS"] = env
        try:
            fa.encode_flac_device(x, level=5, workspace=ws)
        except RuntimeError as e:  # (a variant that writes no offsets may fail its own checks)
            print(label, "failed:", e)
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            out = fa.encode_flac_device(x, level=5, workspace=ws)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del out
        print(f"   {label:14s} {np.median(ts) * 1e3:8.3f} ms", flush=True)
        L = fa._lib.lib()
        import ctypes

        L.fa_profile_enable(1)
        out = fa.encode_flac_device(x, level=5, workspace=ws)
        torch.cuda.synchronize()
        ms = (ctypes.c_float * 6)()
        L.fa_profile_read(ms, 6)
        L.fa_profile_enable(0)
        del out
        print("   HIP events (ms): frame kernel %.3f, K5 %.3f, whole sequence %.3f" % (ms[0], ms[1], ms[3]), flush=True)
        if hasattr(L, "fa_debug_stamps"):
            import ctypes

            buf = (ctypes.c_ulonglong * 64)()
            L.fa_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
            L.fa_debug_stamps(buf, 0)
            nfr = max(1, buf[16])
            if env:
                print(f"   slot kernel: most workgroups alive at once {buf[41]}", flush=True)
            print("   frame body, ticks per frame by phase: " + " ".join(f"{buf[i] / nfr:.0f}" for i in range(14)) + f"  ({nfr} frames)", flush=True)
        if not env and hasattr(L, "fa_debug_stamps"):  # (-DFA_STAMPS build: K3G's cycles per phase, summed over the frames of every 64th workgroup)
            import ctypes

            buf = (ctypes.c_ulonglong * 64)()
            L.fa_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
            L.fa_debug_stamps(buf, 1)
            v = [buf[20 + i] for i in (0, 1, 2, 3, 4, 6)]
            tot = float(sum(v)) or 1.0
            print(f"   workgroups that encoded at least one frame (6 calls): {buf[30]}, most frames in one workgroup {buf[31]}", flush=True)
            print("   workgroup starts by 2.5 ms bin after the first one (all calls): " + " ".join(str(buf[32 + i]) for i in range(8)), flush=True)
            print(f"   frames stamped {buf[25]} (6 calls), s_memtime ticks per frame {tot / max(1, buf[25]):.0f} (100 MHz: x 0.01 us)", flush=True)
            print("   phase shares of a wave's loop: " + ", ".join(f"{n} {100 * c / tot:.1f} %" for n, c in zip(("ticket", "frame body", "tables", "wait for offset", "placement copy", "headers + index"), v)), flush=True)
    sys.exit(0)
for spec in sys.argv[1:]:
    env = dict(os.environ)
    spec, _, envs = spec.partition(":")  # name[:ENV=VAL,...]
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
        print(f"    {k}={v}")
    lib = "libflacarray_hip.so" if spec in ("", "shipped") else f"libflacarray_hip_{spec}.so"
    env["FLACARRAY_HIP_LIB"] = os.path.join(root, "flacarray_amd", "lib", lib)
    print(f"--- {spec}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
