import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, flacarray_amd as fa
from flacarray_amd import _lib
L=_lib.lib()
x = bench.make_data(torch, 2048, 1<<20, 1, torch.device("cuda",0))
comp, st, nb = fa.encode_flac_device(x, level=5)
L.fa_profile_enable(1)
for r in range(3):
    y = fa.decode_flac_device(comp, st, nb, 1<<20)
    ms=(ctypes.c_float*3)(); L.fa_profile_last(ms)
print(os.path.basename(_lib.LIB_PATH), "decode ms", ms[2], "equal", bool(torch.equal(x,y)))
