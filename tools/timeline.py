#!/usr/bin/env python3
"""Where a frame's wait for its byte offset comes from (diagnostic build -DFA_TIMELINE: the FrameInfo record carries
100 MHz timestamps).  FLACARRAY_HIP_LIB=.../libflacarray_hip_tl.so python tools/timeline.py [channels]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    import flacarray_amd as fa

    n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    x = bench.make_data(torch, n_ch, 1 << 20, 123456789, torch.device("cuda", 0))
    for rep in range(2):
        comp, st, nb, info = fa.encode_flac_device(x, level=5, return_info=True)
    t = info.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    start, pub, off, req, got = t[:, 3], t[:, 4], t[:, 2], t[:, 5], t[:, 7]
    t0 = start.min()
    start, pub, off, req, got = (v - t0 for v in (start, pub, off, req, got))
    n = len(start)
    lo, hi = n // 10, n - n // 10  # steady state
    front = np.maximum.accumulate(pub)  # when every frame up to g has published
    tick = 10.0  # ns
    def show(name, v):
        v = v[lo:hi] * tick / 1000.0
        print(f"  {name:58s} mean {v.mean():8.2f} us   p10 {np.percentile(v, 10):8.2f}   median {np.median(v):8.2f}   p90 {np.percentile(v, 90):8.2f}")
    print(f"{n} frames, kernel span {(got.max()) * tick / 1e6:.3f} ms")
    show("start -> size published", pub - start)
    show("size published -> offset asked for", req - pub)
    show("offset asked for -> offset seen (the wait)", got - req)
    show("own publish -> all predecessors published (stragglers)", front - pub)
    show("all predecessors published -> offset stored (scanner)", off - front)
    show("offset stored -> seen by the frame (poll), where it waited", np.where(off > req, got - off, 0))
    order = np.argsort(start, kind="stable")
    print("  frames whose start order differs from ticket order:", int((np.diff(start) < 0).sum()))


if __name__ == "__main__":
    main()
