#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: HIP encoder output byte-identical to the oracle's, and decode(encode(x)) == x,
over random shapes, levels, amplitudes and signal classes (int32 and int64).  Diagnostic companion of
tests/test_gpu_parity.py (same checks, many more random cases):  python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flacarray_amd as fa  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)

LENGTHS = [1, 2, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1151, 1152, 1153, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 20000, 65536]


def signal(rng, kind, n_stream, n, i64):
    t = np.arange(n, dtype=np.float64)
    hi = 62 if i64 else 31
    bits = int(rng.integers(1, hi + 1))
    amp = float(2.0 ** bits - 1)
    if kind == 0:  # uniform noise of a random width
        x = rng.integers(-int(amp), int(amp) + 1, size=(n_stream, n), dtype=np.int64)
    elif kind == 1:  # sinusoids + gaussian noise
        f = rng.uniform(0.0005, 0.2, size=(n_stream, 1))
        x = np.rint(0.5 * amp * np.sin(2 * np.pi * f * t) + rng.normal(0, max(amp / 2 ** rng.integers(1, 12), 0.5), size=(n_stream, n)))
    elif kind == 2:  # sparse spikes on a quiet floor (long unary runs, escapes, row cap)
        x = rng.integers(-3, 4, size=(n_stream, n), dtype=np.int64)
        m = rng.random((n_stream, n)) < 0.003
        x = np.where(m, rng.integers(-int(amp), int(amp) + 1, size=(n_stream, n)), x)
    elif kind == 3:  # piecewise constant / ramps
        seg = max(1, int(rng.integers(1, 3000)))
        x = np.repeat(rng.integers(-int(amp), int(amp) + 1, size=(n_stream, n // seg + 1)), seg, axis=1)[:, :n] + (t * rng.integers(-3, 4)).astype(np.int64)
    elif kind == 4:  # random walk (strongly predictable)
        x = np.cumsum(rng.integers(-int(min(amp, 2 ** 20)), int(min(amp, 2 ** 20)) + 1, size=(n_stream, n)), axis=1)
    else:  # wasted bits
        w = int(rng.integers(1, 12))
        x = rng.integers(-int(amp) >> w, (int(amp) >> w) + 1, size=(n_stream, n), dtype=np.int64) << w
    lim = 2 ** (63 if i64 else 31)
    if x.dtype.kind == "f":
        x = np.clip(x, -(lim - 1024), lim - 1024)
    else:
        x = np.clip(x, -lim, lim - 1)
    x = x.astype(np.int64 if i64 else np.int32)
    if rng.random() < 0.15 and n > 1:  # extreme values
        x[0, 0] = -lim
        x[0, -1] = lim - 1
    return np.ascontiguousarray(x)


def run(n_cases, seed, verbose=True, single_pass_only=False):
    """returns (number of mismatching cases, samples checked); single_pass_only: int32 streams whose length is a
    multiple of 4096 -- or a multiple of 4 beyond 8192 -- at levels 3-8, i.e. only geometries the single-pass encoder takes"""
    rng = np.random.default_rng(seed)
    bad = 0
    tot = 0
    t0 = time.time()
    for c in range(n_cases):
        i64 = rng.random() < 0.25
        level = int(rng.choice([0, 1, 3, 4, 5, 5, 5, 6, 8]))
        n = int(rng.choice(LENGTHS)) if rng.random() < 0.7 else int(rng.integers(1, 30000))
        n_stream = int(rng.integers(1, 6))
        kind = int(rng.integers(0, 6))
        if single_pass_only:
            i64 = False
            level = int(rng.choice([3, 4, 5, 5, 5, 6, 7, 8]))
            n = 4096 * int(rng.integers(1, 6))
            if rng.random() < 0.5:  # a short last frame (the slot encoder's, placed by the single-pass scanner)
                n = 4096 * int(rng.integers(2, 6)) + 4 * int(rng.integers(1, 1024))
            n_stream = int(rng.integers(1, 12))
        x = signal(rng, kind, n_stream, n, i64)
        comp, starts, nbytes = fa.encode_flac(x, level)
        ob, os_, on = (O.encode_i64 if i64 else O.encode_i32)(x, level)
        same = np.array_equal(np.asarray(comp), ob) and np.array_equal(starts.reshape(-1), os_.reshape(-1))
        y = fa.decode_flac(comp, starts, nbytes, n, is_int64=i64)
        rt = np.array_equal(y.reshape(x.shape), x)
        tot += x.size
        if not (same and rt):
            bad += 1
            print(f"MISMATCH case {c} seed {seed}: i64={i64} level={level} shape={x.shape} kind={kind} bytes_equal={same} roundtrip={rt}", flush=True)
        if verbose and c % 500 == 499:
            print(f"{c + 1} cases, {tot / 1e6:.1f} Msamples, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    return bad, tot


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
    bad, tot = run(n_cases, seed, single_pass_only="--single-pass" in sys.argv)
    print(f"done: {n_cases} cases, {tot / 1e6:.1f} Msamples, mismatches: {bad}")
    sys.exit(1 if bad else 0)
