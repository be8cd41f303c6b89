#!/usr/bin/env python3
"""Run tools/kbench.py once per library variant inside one GPU session (box-to-box variance is several per cent):
   python tools/kb_variants.py [--channels N] name[:ENV=VAL,...] ...   (name '' = the shipped library)
Each variant is flacarray_amd/lib/libflacarray_hip_<name>.so (python -m flacarray_amd.build --variant <name> -D...)."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
channels = "2048"
reps = "15"
while args and args[0] in ("--channels", "--reps"):
    if args[0] == "--channels":
        channels = args[1]
    else:
        reps = args[1]
    args = args[2:]
for rnd in range(2):
    for spec in args:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        lib = "libflacarray_hip.so" if name in ("", "shipped") else f"libflacarray_hip_{name}.so"
        env["FLACARRAY_HIP_LIB"] = os.path.join(root, "flacarray_amd", "lib", lib)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        print(f"--- {spec}", flush=True)
        subprocess.run([sys.executable, os.path.join(root, "tools", "kbench.py"), "--channels", channels, "--reps", reps], env=env, check=False)
