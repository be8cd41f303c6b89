import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import flacarray_amd as fa
rng = np.random.default_rng(0)
x = rng.integers(-30000, 30000, (12, 1000)).astype(np.int32)
comp, st, nb = fa.encode_flac(x, 5); c = np.asarray(comp)
fa.decode_flac(c, st, nb, 1000)
for rep in range(3):
    print("--- encode", file=sys.stderr); t0 = time.perf_counter(); comp, st, nb = fa.encode_flac(x, 5); t1 = time.perf_counter()
    print("encode_flac %.3f ms" % ((t1 - t0) * 1e3), file=sys.stderr)
    print("--- decode", file=sys.stderr); t0 = time.perf_counter(); y = fa.decode_flac(c, st, nb, 1000); t1 = time.perf_counter()
    print("decode_flac %.3f ms" % ((t1 - t0) * 1e3), file=sys.stderr)
