import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, flacarray_amd as fa
from flacarray_amd import _lib
from flacarray_amd.libflacarray import EncodeWorkspace
L=_lib.lib()
x = bench.make_data(torch, 2048, 1<<20, 1, torch.device("cuda",0))
ws = EncodeWorkspace()
L.fa_profile_enable(1)
for r in range(3):
    comp, st, nb = fa.encode_flac_device(x, level=5, workspace=ws)
    ms=(ctypes.c_float*3)(); L.fa_profile_last(ms)
print(os.path.basename(_lib.LIB_PATH), "encode ms", ms[0], "compact", ms[1])
