#!/usr/bin/env python3
"""A/B of two builds of the library inside one GPU session (box-to-box variance is +-10%):
   python tools/kb_ab.py libA.so libB.so   -> runs tools/kb_encode_only.py / kb_decode_only.py as child processes."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = os.environ.get("KB_AB", "encode")
script = os.path.join(root, "tools", "kb_encode_only.py" if which == "encode" else "kb_decode_only.py")
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, FLACARRAY_HIP_LIB=os.path.join(root, lib))
        subprocess.run([sys.executable, script], env=env, check=True)
