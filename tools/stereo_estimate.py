#!/usr/bin/env python3
"""What would stereo decorrelation buy on the reference's two-channel streams?  (CPU only, numpy.)

The reference stores an int64 sample as two 32-bit FLAC channels, low word and high word (utils.c:96-123); libFLAC then
picks, per frame, the cheapest of left/right, left/side, side/right and mid/side.  This script estimates the bits of
the four candidate channels (left = low word, right = high word, mid, side) with the same kind of estimate an encoder
uses -- best fixed predictor of order 0..4 by sum of magnitudes, Rice parameter from the mean -- on the kinds of int64
data the tests and benches use, and prints which assignment wins.  It is the measurement behind DESIGN.md section 7's
"not built" for the int64 encoder's stereo decision."""
import numpy as np


def rice_bits(res):
    """bits of a partition coded with the best single Rice parameter (RFC 9639 9.2.7)"""
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.float64)
    best = None
    for k in range(0, 40):
        b = float(np.sum(np.floor(u / 2.0**k))) + (k + 1) * u.size
        best = b if best is None or b < best else best
    return best


def channel_bits(x, frame=4096):
    """estimated bits of one channel (python ints in an object array would be slow: int64 is enough for 33-bit values)"""
    total = 0.0
    for f0 in range(0, x.size, frame):
        s = x[f0 : f0 + frame].astype(np.int64)
        if np.all(s == s[0]):
            total += 40
            continue
        best = None
        e = s.copy()
        for order in range(5):
            if order:
                e = np.diff(e)
            b = rice_bits(e) + order * 33
            best = b if best is None or b < best else best
        total += best
    return total


def report(name, v):
    lo = (v & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)  # channel 0: low word, read as signed (utils.c:104)
    hi = (v >> 32).astype(np.int64)                                            # channel 1: high word
    side = lo - hi
    mid = (lo + hi) >> 1
    bl, br, bm, bs = (channel_bits(c) for c in (lo, hi, mid, side))
    cand = {"left/right": bl + br, "left/side": bl + bs, "side/right": bs + br, "mid/side": bm + bs}
    win = min(cand, key=cand.get)
    n = v.size
    print(f"{name:34s} " + "  ".join(f"{k} {b / n:6.2f}" for k, b in cand.items()) + f"  bits/sample -> {win}")


def main():
    rng = np.random.default_rng(5)
    n = 1 << 16
    t = np.arange(n)
    report("noise, sigma 2^10, both signs", np.rint(rng.normal(0, 2**10, n)).astype(np.int64))
    report("noise, sigma 2^20, both signs", np.rint(rng.normal(0, 2**20, n)).astype(np.int64))
    report("sinusoid + noise, 2^16 (bench)", np.rint(65536 * (3 * np.sin(2 * np.pi * t / 20000.0) + rng.normal(0, 1, n))).astype(np.int64))
    report("positive only, < 2^20", rng.integers(0, 2**20, n).astype(np.int64))
    report("tiny: -2..2", rng.integers(-2, 3, n).astype(np.int64))
    report("wide: sigma 2^40", np.rint(rng.normal(0, 2.0**40, n)).astype(np.int64))
    report("counter (ramp) near 2^32", (2**32 - 3000 + 7 * t).astype(np.int64))
    report("full-range int64", rng.integers(-(2**62), 2**62, n).astype(np.int64))


if __name__ == "__main__":
    main()
