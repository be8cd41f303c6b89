"""Decode of the benchmark store with and without the frame CRC-16 check (K9): beside K7 on a side stream (default) and
after it (FLACARRAY_HIP_VERIFY_AFTER=1).  Wall-clock per call, stream synchronised.  python tools/kb_verify.py [channels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import flacarray_amd as fa  # noqa: E402

n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 1 << 20
dev = torch.device("cuda", 0)
x = bench.make_data(torch, n_ch, n, 123456789, dev)
comp, st, nb = fa.encode_flac_device(x, level=5)


def run(verify, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fa.decode_flac_device(comp, st, nb, n, verify=verify)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del y
    return min(ts[1:]) * 1e3


print("channels", n_ch)
print("decode, no check        %.3f ms" % run(False))
print("decode + K9 beside K7   %.3f ms  (side stream priority: %s)" % (run(True), os.environ.get("FLACARRAY_HIP_VERIFY_PRIO", "lowest")))
os.environ["FLACARRAY_HIP_VERIFY_AFTER"] = "1"
print("decode + K9 after K7    %.3f ms" % run(True))
