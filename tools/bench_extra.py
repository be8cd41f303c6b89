#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json on one MI355X (not the headline bench line):
  cfg 3: 4096ch x 1Msamp float32 with per-channel quanta (quantise + encode, decode + fused dequantise)
  cfg 5 (one-GPU analogue): 10 000 scattered (channel, sample-range) slices from a 4096ch store
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--slices", type=int, default=10000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch

    import bench
    import flacarray_amd as fa
    from flacarray_amd.libflacarray import EncodeWorkspace

    dev = torch.device("cuda", 0)
    n_ch, n = args.channels, args.samples
    out = {}

    # ---- cfg 3: float32 ----
    xi = bench.make_data(torch, n_ch, n, 123456789, dev)
    xf = xi.to(torch.float32) * (1.0 / 65536.0)  # the field before rint, amplitude 1 (approximately: exact multiples of 2^-16)
    xf += (torch.rand(xf.shape, device=dev) - 0.5) * (2.0**-17)
    del xi
    q = (2.0**-16 * (1 + torch.arange(n_ch, device=dev) % 4)).to(torch.float32)
    ws = EncodeWorkspace()

    def f32_step():
        idata, off, gain = fa.float32_to_int32_device(xf, q)
        comp, st, nb = fa.encode_flac_device(idata, level=5, workspace=ws)
        del idata
        y = fa.decode_flac_device(comp, st, nb, n, offsets=off, gains=gain)
        return comp, off, y

    f32_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        comp, off, y = None, None, None
        comp, off, y = f32_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tol = 0.5 * q[:, None] + 4 * torch.finfo(torch.float32).eps * (xf.abs() + off.abs()[:, None])
    ok = bool(((y - xf).abs() <= tol).all())
    out["cfg3_float32"] = {"Msamples_per_s": round(n_ch * n / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 2),
                           "bytes_per_sample": round(comp.numel() / (n_ch * n), 4), "within_half_quantum": ok}
    del xf, y, comp

    # ---- cfg 5 analogue: scattered slices ----
    x = bench.make_data(torch, n_ch, n, 42, dev)
    comp, st, nb = fa.encode_flac_device(x, level=5, workspace=ws)
    rng = np.random.default_rng(987654321)
    ns = args.slices
    ch = rng.integers(0, n_ch, ns)
    cnt = rng.integers(1, 8193, ns)
    first = np.array([rng.integers(0, n - c + 1) for c in cnt])
    fa.decode_slices_device(comp, st, nb, n, ch[:16], first[:16], cnt[:16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, off = fa.decode_slices_device(comp, st, nb, n, ch, first, cnt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # verify a sample of the slices
    good = True
    for i in rng.integers(0, ns, 200):
        good = good and bool(torch.equal(res[off[i] : off[i] + cnt[i]], x[ch[i], first[i] : first[i] + cnt[i]]))
    out["cfg5_slices"] = {"slices": ns, "seconds": round(dt, 4), "slices_per_s": round(ns / dt, 1),
                          "Msamples_per_s": round(cnt.sum() / dt / 1e6, 1), "verified": good}
    # CPU side of cfg 5: the same requests as single-stream decode calls at the C level of the
    # CPU oracle (a port; it seeks through the SEEKTABLE as libFLAC's seek_absolute would), on a subset
    # of the streams copied to the host; scaled to the request count
    if not args.no_cpu:
        from oracle import oracle as O

        O.lib().oracle_set_threads(1)
        n_host = 64
        hb_st, hb_nb = st[:n_host].cpu().numpy(), nb[:n_host].cpu().numpy()
        hb = comp[: int(hb_st[-1] + hb_nb[-1])].cpu().numpy()
        xs = x[:n_host].cpu().numpy()
        m = min(ns, 1000)
        t0 = time.perf_counter()
        okc = True
        for i in range(m):
            c = int(ch[i]) % n_host
            y = O.decode_i32(hb, hb_st[c : c + 1], hb_nb[c : c + 1], n, int(first[i]), int(first[i] + cnt[i]))
            okc = okc and bool(np.array_equal(y[0], xs[c, first[i] : first[i] + cnt[i]]))
        dtc = time.perf_counter() - t0
        out["cfg5_slices"]["cpu_port_1thread"] = {"slices": m, "seconds": round(dtc, 3), "slices_per_s": round(m / dtc, 1),
                                                   "verified": okc, "through": "ctypes call per slice"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
