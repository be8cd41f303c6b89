import sys, os, ctypes, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, flacarray_amd as fa
from flacarray_amd import _lib
L=_lib.lib()
dev=torch.device("cuda",0)
n=1<<20
for name, x in (("noise32", torch.randint(-2**31, 2**31-1, (1024, n), device=dev, dtype=torch.int32)),
                ("sinus", bench.make_data(torch, 1024, n, 1, dev))):
    comp, st, nb = fa.encode_flac_device(x, level=5)
    L.fa_profile_enable(1)
    for r in range(3):
        y = fa.decode_flac_device(comp, st, nb, n)
        ms=(ctypes.c_float*3)(); L.fa_profile_last(ms)
    print(name, "mono decode ms", round(ms[2],3), "equal", bool(torch.equal(x,y)), "B/sample", comp.numel()/x.numel())
    del x, y, comp
# stereo: low word noise, high word small
x64 = (torch.randint(-2**31, 2**31-1, (512, n), device=dev, dtype=torch.int64) & 0xFFFFFFFF) | (torch.randint(-3, 4, (512, n), device=dev, dtype=torch.int64) << 32)
comp, st, nb = fa.encode_flac_device(x64, level=5)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    y = fa.decode_flac_device(comp, st, nb, n, is_int64=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    ms=(ctypes.c_float*3)(); L.fa_profile_last(ms)
print("stereo lownoise decode K7 ms", round(ms[2],3), "total", round(dt*1e3,2), bool(torch.equal(x64,y)))
del x64, y
# stereo: both words smooth (compressible)
a = bench.make_data(torch, 512, n, 3, dev).to(torch.int64)
x64 = (a & 0xFFFFFFFF) | ((a >> 3) << 32)
comp, st, nb = fa.encode_flac_device(x64, level=5)
for r in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    y = fa.decode_flac_device(comp, st, nb, n, is_int64=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    ms=(ctypes.c_float*3)(); L.fa_profile_last(ms)
print("stereo smooth decode K7 ms", round(ms[2],3), "total", round(dt*1e3,2), bool(torch.equal(x64,y)), "B/sample", comp.numel()/x64.numel())
