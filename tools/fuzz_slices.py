#!/usr/bin/env python3
"""Randomised check of small reads (the latency decoder's territory): random arrays of every signal class of
tools/fuzz_parity.py, encoded by the HIP encoder, read back as random batches of slices through a decode index -- left on
the device and as numpy arrays (pinned landing / pooled blocks) -- and compared with the source.
python tools/fuzz_slices.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import flacarray_amd as fa  # noqa: E402
from tools.fuzz_parity import LENGTHS, signal  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(seed)
bad = tot = 0
t0 = time.time()
for c in range(n_cases):
    i64 = rng.random() < 0.25
    level = int(rng.choice([0, 1, 3, 5, 5, 5, 8]))
    n = int(rng.choice(LENGTHS)) if rng.random() < 0.5 else int(rng.integers(1, 40000))
    n_stream = int(rng.integers(1, 5))
    x = signal(rng, int(rng.integers(0, 6)), n_stream, n, i64)
    xd = torch.from_numpy(x).cuda()
    comp, st, nb = fa.encode_flac_device(xd, level=level, compact=True)
    ix = fa.DeviceDecodeIndex(comp, st, nb, n, is_int64=i64)
    try:
        for rep in range(3):
            k = int(rng.integers(1, 12))
            ch = rng.integers(0, n_stream, k)
            cnt = rng.integers(1, min(n, 9000) + 1, k)
            first = np.array([rng.integers(0, n - cc + 1) for cc in cnt])
            want = np.concatenate([x[a, f : f + cc] for a, f, cc in zip(ch, first, cnt)])
            dev, _ = ix.decode_slices(ch, first, cnt)
            host, _ = ix.decode_slices(ch, first, cnt, to_host=True)
            ok = np.array_equal(dev.cpu().numpy(), want) and np.array_equal(np.asarray(host), want)
            tot += want.size
            if not ok:
                bad += 1
                print(f"MISMATCH case {c} seed {seed}: i64={i64} level={level} shape={x.shape}", flush=True)
    finally:
        ix.close()
    if c % 200 == 199:
        print(f"{c + 1} cases, {tot / 1e6:.1f} Msamples read, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {tot / 1e6:.1f} Msamples read, mismatches: {bad}")
sys.exit(1 if bad else 0)
