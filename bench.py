#!/usr/bin/env python3
"""bench.py -- Msamples/s encode+decode of a 4096ch x 1Msamp int32 array per MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: encode the HBM-resident int32 matrix
(K3F: analyse, size, place and write every frame in one kernel, then the stream headers) into the reference's
(compressed, starts, nbytes) triple, then decode that triple back to int32 (K6 index, K7 decode).
Weak scaling: every rank owns its own 4096 channels (contiguous leading-axis shard, the
reference's mpi.py:84-90 distribution); no data-path collective -- only the per-stream byte
counts are all-gathered to form the global starts (the analogue of mpi.py:177).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_data(torch, n_ch, n_samp, seed, device):
    """S2 of SURVEY.md 8(d) generated on the device in channel tiles:
    rint(2^16 * (dc_c + s_c*(2 sin(2pi 3f t) + 6 sin(2pi f t)) + N(0,1))), f = 5/T."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_ch, n_samp), dtype=torch.int32, device=device)
    t = torch.arange(n_samp, device=device, dtype=torch.float32)
    f = 5.0 / n_samp
    wave = 2.0 * torch.sin(2 * np.pi * 3 * f * t) + 6.0 * torch.sin(2 * np.pi * f * t)
    tile = 64
    for c0 in range(0, n_ch, tile):
        c1 = min(n_ch, c0 + tile)
        dc = 5.0 * (torch.rand((c1 - c0, 1), generator=g, device=device) - 0.5)
        sc = torch.rand((c1 - c0, 1), generator=g, device=device)
        x = dc + sc * wave + torch.randn((c1 - c0, n_samp), generator=g, device=device)
        out[c0:c1] = torch.round(x * 65536.0).to(torch.int32)
    return out


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the most recent committed PMC measurement that has this kernel
    (profiles/r*_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, corrected as MI355X_MICROARCH.md prescribes); None if there is none."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            data = json.load(open(path))
            for name, rec in data["kernels"].items():
                if kernel in name:
                    return round(rec["hbm_bytes"] / 1e9, 3)
        except Exception:
            continue
    return None


def measured_mix(kernel):
    """Instructions per wave of `kernel` from the newest committed PMC pass (profiles/r*_traffic.json, key
    "instruction_mix"); None if there is none."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            for name, rec in json.load(open(path)).get("instruction_mix", {}).items():
                if kernel in name:
                    return dict(rec, source=os.path.basename(path))
        except Exception:
            continue
    return None


def measured_issue(kernel):
    """Issue-side picture of `kernel` from the newest committed PMC pass that has one (profiles/r*_traffic.json,
    key "issue": share of a wave's life spent issuing vector instructions x waves per SIMD); None if there is none."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            for name, rec in json.load(open(path)).get("issue", {}).items():
                if kernel in name:
                    return dict(rec, source=os.path.basename(path))
        except Exception:
            continue
    return None


def cpu_baseline(n_samp, seconds_budget=25.0):
    """Time the CPU oracle (a port, not libFLAC: libFLAC is absent from this image) on a bounded
    sample of the same workload with every host core."""
    from oracle import oracle as O

    # what this process may actually use: the scheduler affinity mask and the cgroup's CPU quota (a one-GPU box gives a
    # share of the host, whatever os.cpu_count() says); the baseline runs on that many threads
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cpu_max = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                cpu_max = fh.read().strip()
            if path.endswith("cfs_quota_us"):
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    cpu_max = cpu_max + " " + fh.read().strip()
            break
        except OSError:
            continue
    quota_cpus = None
    if cpu_max:
        parts = cpu_max.split()
        if len(parts) == 2 and parts[0] not in ("max", "-1"):
            quota_cpus = max(1, int(float(parts[0]) / float(parts[1]) + 0.5))
    usable = min(affinity, quota_cpus) if quota_cpus else affinity
    threads = int(os.environ.get("FA_BENCH_CPU_THREADS", "0")) or usable
    threads = max(1, min(threads, O.lib().oracle_num_threads(), 64))
    O.lib().oracle_set_threads(threads)
    n_ch = 16 * threads  # ~15-20 s of CPU work whatever the thread count (16 channels of 2^20 samples per thread)
    rng = np.random.default_rng(123456789)
    t = np.arange(n_samp)
    f = 5.0 / n_samp
    wave = 2.0 * np.sin(2 * np.pi * 3 * f * t) + 6.0 * np.sin(2 * np.pi * f * t)
    x = np.rint(65536.0 * (5.0 * (rng.random((n_ch, 1)) - 0.5) + rng.random((n_ch, 1)) * wave + rng.normal(0, 1, (n_ch, n_samp)))).astype(np.int32)
    t0 = time.perf_counter()
    blob, st, nb = O.encode_i32(x, 5, use_threads=True)
    t1 = time.perf_counter()
    y = O.decode_i32(blob, st, nb, n_samp, use_threads=True)
    t2 = time.perf_counter()
    assert np.array_equal(x, y)
    # SURVEY 8(d) / BASELINE.md 2: the one-thread figure beside the all-cores one (a 16-channel slice of the same sample)
    O.lib().oracle_set_threads(1)
    x1 = x[:16]
    s0 = time.perf_counter()
    b1, st1, nb1 = O.encode_i32(x1, 5, use_threads=False)
    s1 = time.perf_counter()
    y1 = O.decode_i32(b1, st1, nb1, n_samp, use_threads=False)
    s2 = time.perf_counter()
    assert np.array_equal(x1, y1)
    O.lib().oracle_set_threads(threads)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    cflags = "unknown"
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            cflags = next((ln.split("=", 1)[1].strip() for ln in f if ln.startswith("CFLAGS")), "unknown")
    except OSError:
        pass
    # SURVEY 8(d), cfg 5 on the CPU: single-stream decode(first, last) calls at the C level, one thread
    rch, rfirst, rcnt = slice_requests(16, n_samp, 200, seed=24680)
    r0 = time.perf_counter()
    for i in range(200):
        c = int(rch[i])
        O.decode_i32(b1, st1[c : c + 1].copy(), nb1[c : c + 1].copy(), n_samp, int(rfirst[i]), int(rfirst[i] + rcnt[i]))
    single_read_us = (time.perf_counter() - r0) / 200 * 1e6
    out = {
        "value": round(x.size / (t2 - t0) / 1e6, 2),
        "unit": "Msamples/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"{n_ch}ch x {n_samp} int32 sinusoid+noise, level 5, encode {x.size/(t1-t0)/1e6:.1f} + decode {x.size/(t2-t1)/1e6:.1f} Msamples/s",
        "cpu_model": cpu_model,
        "host_cores_visible": os.cpu_count(),
        "affinity_cpus": affinity,
        "cgroup_cpu_max": cpu_max,
        "threads_1": {"value": round(x1.size / (s2 - s0) / 1e6, 2), "unit": "Msamples/s",
                      "sample": f"16ch x {n_samp}, one thread: encode {x1.size/(s1-s0)/1e6:.1f} + decode {x1.size/(s2-s1)/1e6:.1f} Msamples/s"},
        "compiler": "gcc " + cflags,
        "single_read_us": round(single_read_us, 1),  # one scattered (channel, range<=8192) slice per call, one thread (cfg 5 on the CPU)
    }
    # a system libFLAC, if this box has one, gives the reference's own engine (one thread, through the
    # ctypes harness of oracle/libflac_harness.py) and a cross-decode of the port's streams
    try:
        from oracle import libflac_harness as H

        if H.available():
            xs = x[:4]
            t3 = time.perf_counter()
            lb, ls, ln = H.encode_i32(xs, 5)
            t4 = time.perf_counter()
            ys = H.decode_i32(lb, ls, ln, n_samp)
            t5 = time.perf_counter()
            cross = bool(np.array_equal(ys, xs)) and bool(np.array_equal(H.decode_i32(blob, st[:4], nb[:4], n_samp), xs))
            out["libflac"] = {"version": H.version(), "threads": 1, "encode_Msamples_per_s": round(xs.size / (t4 - t3) / 1e6, 2),
                              "decode_Msamples_per_s": round(xs.size / (t5 - t4) / 1e6, 2), "cross_decode_ok": cross,
                              "bytes_libflac": int(lb.size), "bytes_port": int(nb[:4].sum())}
        else:
            out["libflac"] = "unavailable on this box (ctypes.util.find_library('FLAC') is None): parity with libFLAC unpinned"
    except Exception as e:  # the harness must never take the bench down
        out["libflac"] = f"harness error: {type(e).__name__}: {e}"[:200]
    return out


def make_float_data(torch, n_ch, n_samp, seed, device):
    """S3 of SURVEY.md 8(d): the same field before rint, float32, amplitude 1 (configuration 3)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_ch, n_samp), dtype=torch.float32, device=device)
    t = torch.arange(n_samp, device=device, dtype=torch.float32)
    f = 5.0 / n_samp
    wave = 2.0 * torch.sin(2 * np.pi * 3 * f * t) + 6.0 * torch.sin(2 * np.pi * f * t)
    for c0 in range(0, n_ch, 64):
        c1 = min(n_ch, c0 + 64)
        dc = 5.0 * (torch.rand((c1 - c0, 1), generator=g, device=device) - 0.5)
        sc = torch.rand((c1 - c0, 1), generator=g, device=device)
        out[c0:c1] = dc + sc * wave + torch.randn((c1 - c0, n_samp), generator=g, device=device)
    return out


def slice_requests(n_ch, n_samp, n, seed=987654321):
    """S4 of SURVEY.md 8(d): n scattered (channel, first, length) requests, length 1..8192."""
    rng = np.random.default_rng(seed)
    ch = rng.integers(0, n_ch, n)
    cnt = rng.integers(1, 8193, n)
    first = (rng.random(n) * (n_samp - cnt + 1)).astype(np.int64)
    return ch.astype(np.int64), first, cnt.astype(np.int64)


def profile_read(L, n=6):
    ms = (ctypes.c_float * n)()
    L.fa_profile_read(ms, n)
    return [float(v) for v in ms]


def bench_cfg3(torch, fa, L, n_ch, n_samp, level, dev, steps=3):
    """Configuration 3 on one GPU: float32 in -> range pre-pass -> encode with the quantisation (per-channel quanta)
    fused into the staging load -> decode with the fused restore -> float32 out.  Reported beside the headline, never
    as `value`."""
    from flacarray_amd.libflacarray import EncodeWorkspace

    xf = make_float_data(torch, n_ch, n_samp, 31337, dev)
    q = (2.0**-16 * (1 + torch.arange(n_ch, device=dev) % 4)).to(torch.float32)
    ws = EncodeWorkspace()

    def step():
        comp, st, nb, off, gain = fa.encode_flac_device_f32(xf, q, level=level, workspace=ws)
        return fa.decode_flac_device(comp, st, nb, n_samp, offsets=off, gains=gain), comp

    y, comp = step()
    torch.cuda.synchronize()
    L.fa_profile_enable(1)
    k1 = []
    t0 = time.perf_counter()
    for _ in range(steps):
        y = comp = None
        y, comp = step()
        k1.append(profile_read(L)[5])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    L.fa_profile_enable(0)
    err = float(((y - xf).abs() / (0.5 * q[:, None])).max())  # in units of half a quantum (<= 1 + float32 rounding)
    # the tolerance north_star states (tests/array.py:251-260: |x' - x| <= quanta / 2), plus the float32 rounding of the
    # restore's multiply and add at the magnitude of the data: 4 eps |x| in units of half a quantum
    bound = 1.0 + 4.0 * float(np.finfo(np.float32).eps) * float(xf.abs().max()) / float(0.5 * q.min())
    assert err <= bound, f"configuration 3: worst error {err} half quanta exceeds {bound}"
    return {
        "workload": f"{n_ch}ch x {n_samp} float32, per-channel quanta 2^-16 (1 + c mod 4), level {level}: quantise, encode, decode + restore",
        "ms_per_step": round(dt * 1e3, 3),
        "Msamples_per_s": round(n_ch * n_samp / dt / 1e6, 1),
        "range_prepass_ms": round(float(np.mean(k1)), 3),  # K1a + K1b (min / max per stream); the quantisation itself runs inside the encoder
        "compressed_bytes_per_sample": round(comp.numel() / xf.numel(), 4),
        "max_abs_err_in_half_quanta": round(err, 4),
        "max_abs_err_bound": round(bound, 4),
    }


def bench_end_to_end(fa, n_ch=512, n_samp=1 << 20, level=5):
    """numpy in -> numpy out through the reference-shaped host API (encode_i32 / decode_i32 behind encode_flac /
    decode_flac): PCIe both ways included.  2 GiB of int32, second of two repetitions.  Never `value`."""
    rng = np.random.default_rng(7)
    t = np.arange(n_samp)
    f = 5.0 / n_samp
    wave = 2.0 * np.sin(2 * np.pi * 3 * f * t) + 6.0 * np.sin(2 * np.pi * f * t)
    base = np.rint(65536.0 * (rng.random((64, 1)) * wave + rng.normal(0, 1, (64, n_samp)))).astype(np.int32)
    x = np.tile(base, (n_ch // 64, 1))
    fa.encode_flac(x[:8], level)  # library tables, staging buffers
    comp = y = None
    for _ in range(2):
        comp = y = None
        t0 = time.perf_counter()
        comp, st, nb = fa.encode_flac(x, level)
        t1 = time.perf_counter()
        y = fa.decode_flac(np.asarray(comp), st, nb, n_samp)
        t2 = time.perf_counter()
    assert np.array_equal(y, x)
    # the same decode into an array the caller already owns (its pages exist): what is left is PCIe, not the population
    # of fresh host memory (profiles/r03_host_abi.md)
    t3 = t4 = None
    try:
        from flacarray_amd.libflacarray import decode_flac_into

        y[:] = 0
        t3 = time.perf_counter()
        decode_flac_into(np.asarray(comp), st, nb, n_samp, y)
        t4 = time.perf_counter()
        assert np.array_equal(y, x)
    except ImportError:
        pass
    return {
        "workload": f"{n_ch}ch x {n_samp} int32 ({x.nbytes / 2**30:.0f} GiB) numpy -> encode_flac -> numpy -> decode_flac -> numpy, PCIe included",
        "decode_into_existing_array_Gsamples_per_s": round(x.size / (t4 - t3) / 1e9, 2) if t3 is not None else None,
        "encode_Gsamples_per_s": round(x.size / (t1 - t0) / 1e9, 2),
        "decode_Gsamples_per_s": round(x.size / (t2 - t1) / 1e9, 2),
        "encode_s": round(t1 - t0, 4),
        "decode_s": round(t2 - t1, 4),
        "host_input_GBs": round(x.nbytes / (t1 - t0) / 1e9, 1),
    }


def bench_cookbook(fa):
    """The one workload the reference publishes timings for (docs/docs/cookbook.ipynb cells 4-8: hardware not stated,
    OMP_NUM_THREADS=4): create_fake_data((1000, 100000), float32), FlacArray.from_array(arr, quanta=1e-7) and
    to_array(), numpy in and numpy out -- PCIe, the NaN scan and the float conversion included, as in the notebook."""
    from tests.golden import reference_published as P

    arr = P.fake_data((1000, 100000), np.float32)
    fa.FlacArray.from_array(arr[:8], quanta=1.0e-7).to_array()  # library tables, staging buffers
    best_c = best_d = None
    f = None
    for _ in range(2):
        t0 = time.perf_counter()
        f = fa.FlacArray.from_array(arr, quanta=1.0e-7)
        t1 = time.perf_counter()
        back = f.to_array()
        t2 = time.perf_counter()
        best_c = (t1 - t0) if best_c is None else min(best_c, t1 - t0)
        best_d = (t2 - t1) if best_d is None else min(best_d, t2 - t1)
    assert float(np.abs(back - arr).max()) <= 0.5e-7 + 4 * float(np.finfo(np.float32).eps) * float(np.abs(arr).max())
    return {
        "workload": "create_fake_data((1000, 100000), float32), FlacArray.from_array(quanta=1e-7) / to_array(), numpy <-> numpy",
        "from_array_s": round(best_c, 4), "to_array_s": round(best_d, 4),
        "from_array_Msamples_per_s": round(arr.size / best_c / 1e6, 1), "to_array_Msamples_per_s": round(arr.size / best_d / 1e6, 1),
        "compressed_bytes": int(f.nbytes),
        "reference_published": {"from_array_use_threads_s": 1.23, "from_array_s": 2.6, "to_array_s": 0.447,
                                "hardware": "not stated (notebook output, OMP_NUM_THREADS=4)", "source": "docs/docs/cookbook.ipynb cells 5-8"},
    }


def bench_other_geometries(torch, fa, dev, level=5):
    """Device-resident encode of the geometries the headline's kernel (K3F: full 4096-sample mono frames) does not take --
    the reference's own test shapes (src/flacarray/tests/bindings.py:165-230), 1152-sample blocks, int64 -- through the
    placing encoder K3G (csrc/encode_placed.hpp), each beside the slot sequence it retired (FLACARRAY_HIP_SLOTS=1, read
    per call).  Median wall time of a call incl. its one host wait."""
    from flacarray_amd.libflacarray import EncodeWorkspace

    ws = EncodeWorkspace()
    big = make_data(torch, 512, 1 << 20, level, dev)

    def timed(x, lvl, reps):
        ts = []
        for i in range(reps + 1):
            t0 = time.perf_counter()
            out = fa.encode_flac_device(x, level=lvl, workspace=ws)
            torch.cuda.synchronize()
            if i:
                ts.append(time.perf_counter() - t0)
            del out
        return float(np.median(ts))

    res = {}
    cases = [
        ("int32 (12, 1000)", big[:12, :1000].contiguous(), level, 30),
        ("int32 (1, 10000)", big[:1, :10000].contiguous(), level, 30),
        ("int64 (12, 1000)", (big[:12, :1000].to(torch.int64) << 13).contiguous(), level, 30),
        ("int32 (512, 2^20 - 3)", big[:, : (1 << 20) - 3].contiguous(), level, 3),
        ("int32 (512, 2^20) level 1", big, 1, 3),
        ("int64 (512, 2^20)", big.to(torch.int64) * 8192 + torch.randint(-4096, 4096, big.shape, device=dev, dtype=torch.int64), level, 3),
    ]
    for name, x, lvl, reps in cases:
        t_new = timed(x, lvl, reps)
        os.environ["FLACARRAY_HIP_SLOTS"] = "1"
        try:
            t_old = timed(x, lvl, reps)
        finally:
            del os.environ["FLACARRAY_HIP_SLOTS"]
        res[name] = {"single_pass_ms": round(t_new * 1e3, 3), "slot_sequence_ms": round(t_old * 1e3, 3),
                     "Msamples_per_s": round(x.numel() / t_new / 1e6, 1)}
        del x
    res["what"] = "encode_flac_device of geometries outside the headline kernel's (K3G), median ms per call; slot sequence = K3 + K4 + K5 forced"
    return res


def bench_cfg5(torch, fa, comp, st, nb, n_ch, n_samp, x, dev, n_req=10000, reps=5):
    """Configuration 5 on this rank's store: 10 000 scattered (channel, range) slices in one batched launch,
    from tensors resident in HBM."""
    ch, first, cnt = slice_requests(n_ch, n_samp, n_req)
    out, off = fa.decode_slices_device(comp, st, nb, n_samp, ch, first, cnt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, off = fa.decode_slices_device(comp, st, nb, n_samp, ch, first, cnt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    for i in (0, n_req // 2, n_req - 1):  # spot check (the parity tests cover the rest)
        assert torch.equal(out[off[i] : off[i] + cnt[i]], x[ch[i], first[i] : first[i] + cnt[i]])
    return {
        "workload": f"{n_req} scattered (channel, first, length<=8192) slices of the {n_ch}ch x {n_samp} store, one batched launch",
        "ms_per_batch": round(dt * 1e3, 3),
        "slices_per_s": round(n_req / dt, 0),
        "decoded_Msamples_per_s": round(float(cnt.sum()) / dt / 1e6, 1),
    }


def bench_small_reads(fa, comp, st, nb, n_ch, n_samp, x):
    """The reference's usage pattern (array.py:409-449: one decode call per key): single scattered reads and batches
    of 100 from the HBM-resident store through a decode index, samples returned to the host."""
    ch, first, cnt = slice_requests(n_ch, n_samp, 400, seed=24680)
    ix = fa.DeviceDecodeIndex(comp, st.reshape(-1), nb.reshape(-1), n_samp)
    try:
        for i in range(5):
            out, _ = ix.decode_slices(ch[i : i + 1], first[i : i + 1], cnt[i : i + 1], to_host=True)
            assert np.array_equal(out, x[ch[i], first[i] : first[i] + cnt[i]].cpu().numpy())
        for _ in range(2):  # second of two passes (the first one's results open the pinned result pool's size classes: ~10 ms each, once)
            t0 = time.perf_counter()
            for i in range(200):
                ix.decode_slices(ch[i : i + 1], first[i : i + 1], cnt[i : i + 1], to_host=True)
            single = (time.perf_counter() - t0) / 200
        ix.decode_slices(ch[:100], first[:100], cnt[:100], to_host=True)
        t0 = time.perf_counter()
        for r in range(20):
            ix.decode_slices(ch[100 * (r % 4) : 100 * (r % 4) + 100], first[100 * (r % 4) : 100 * (r % 4) + 100], cnt[100 * (r % 4) : 100 * (r % 4) + 100], to_host=True)
        b100 = (time.perf_counter() - t0) / 20
    finally:
        ix.close()
    return {"workload": "scattered (channel, first, length<=8192) reads from the resident store, samples to the host",
            "single_read_us": round(single * 1e6, 1), "batch_of_100_us": round(b100 * 1e6, 1), "batch_of_100_slices_per_s": round(100 / b100, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configuration 3 / 5 legs (extra keys of the JSON line)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the all-gather-v of the compressed blobs out of the step")
    args = ap.parse_args()

    # A multi-rank run that stops making progress (a rank that fell out of a collective) must end with every thread's
    # stack and a non-zero exit, not with the driver's kill: on by default when WORLD_SIZE > 1 (FA_BENCH_WATCHDOG_S
    # overrides the 900 s; 0 disables).  Never a re-exec.
    wd = os.environ.get("FA_BENCH_WATCHDOG_S", "900" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0")
    if int(wd) > 0:
        import faulthandler

        faulthandler.dump_traceback_later(int(wd), exit=True)

    if not args.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # the CPU checker is compiled (if stale) before anything touches the GPU: no compiler child process
        # may start once the HIP runtime -- or a profiler preload -- is live
        from oracle import oracle as O

        O.build()

    import torch

    import flacarray_amd as fa
    from flacarray_amd import _lib, dist as fdist
    from flacarray_amd.libflacarray import EncodeWorkspace

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # test hooks (one-GPU box rehearsal of the multi-rank path): FA_BENCH_FORCE_DEVICE pins every
    # rank to one device, FA_BENCH_BACKEND=gloo avoids RCCL's one-rank-per-GPU requirement
    dev_index = int(os.environ.get("FA_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("FA_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    L = _lib.lib()

    n_ch, n_samp = args.channels, args.samples
    x = make_data(torch, n_ch, n_samp, 123456789 + rank, dev)
    ws = EncodeWorkspace()
    n_global = n_ch * world
    gather_state = {"on": world > 1 and not args.no_gather, "error": None, "bytes": 0, "local": 0, "ms": []}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_agree_ok(ok):
        """The ranks decide TOGETHER whether the assembly leg stays in the step: a rank that failed on its own and
        switched collectives by itself would leave the others waiting in the old one."""
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag.item()) == 0

    def step(with_gather, agree=False):
        comp, st, nb = fa.encode_flac_device(x, level=args.level, workspace=ws)
        pending = None
        if world > 1:
            # Configuration 4: the global (compressed, stream_starts, stream_nbytes) triple on every GPU --
            # all-gather of the per-stream byte counts + exclusive scan (global_bytes, mpi.py:156-187), then the
            # all-gather-v of the shard blobs over xGMI (one batched round of point-to-point transfers) on a side
            # stream, UNDER the decode of the same step, which needs only the local blob.
            if with_gather:
                pending = fdist.assemble_global_async(comp, nb.reshape(-1), n_global, agree=agree)
            else:
                fdist.gather_stream_nbytes(nb.reshape(-1), n_global)
        y = fa.decode_flac_device(comp, st, nb, n_samp)
        if pending is not None:
            g_blob, _, _ = pending.wait()
            gather_state["bytes"] = int(g_blob.numel())
            gather_state["local"] = int(comp.numel())
            ms = pending.elapsed_ms()
            if ms is not None:
                gather_state["ms"].append(ms)
            del g_blob
        return comp, st, nb, y

    def timed_loop(with_gather):
        for _ in range(args.warmup):
            step(with_gather)
        sync()
        L.fa_profile_enable(1)
        prof = []
        t0 = time.perf_counter()
        comp = st = nb = y = None
        for _ in range(args.steps):
            comp = st = nb = y = None  # hand the previous outputs back to the caching allocator (no hipMalloc in the timed region)
            comp, st, nb, y = step(with_gather)
            prof.append(profile_read(L))
        sync()
        t1 = time.perf_counter()
        L.fa_profile_enable(0)
        elapsed = t1 - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, prof, comp, st, nb, y

    compute_only = None
    if gather_state["on"]:
        # one rehearsal step decides, for all ranks together, whether the never-before-run RCCL leg is usable
        # (agree=True: the ranks compare notes after their local preparations and before any transfer is queued, so a
        # rank that cannot take part stops all of them there; a failure inside the transfers themselves ends through
        # the watchdog)
        ok = True
        try:
            step(True, agree=True)
        except Exception as e:  # noqa: BLE001
            ok = False
            gather_state["error"] = f"{type(e).__name__}: {e}"[:300]
        if not all_agree_ok(ok):
            gather_state["on"] = False
            gather_state["error"] = gather_state["error"] or "another rank failed in the all-gather-v rehearsal step"
        gather_state["ms"].clear()
    if world > 1 and gather_state["on"]:
        # the same K steps without the blobs (32 KiB of byte counts per rank and step): the kernels' own scaling
        e_c, _, comp, st, nb, y = timed_loop(False)
        compute_only = {"ms_per_step": round(e_c / args.steps * 1e3, 3), "value": round(n_ch * n_samp * world / (e_c / args.steps) / 1e6, 1),
                        "unit": "Msamples/s", "what": "same loop, all-gather of the stream byte counts only"}
        comp = st = nb = y = None
    elapsed, prof, comp, st, nb, y = timed_loop(gather_state["on"])

    # correctness of what was timed (outside the timed region)
    assert torch.equal(y, x), "decode(encode(x)) != x"
    c_bytes = comp.numel() / x.numel()

    # the decode leg again with the frame CRC-16 check libFLAC performs (decompress.c:104-121): K9 beside K7 on a side
    # stream (event pair 4 closes after both), and the unverified decode timed the same way right beside it
    decode_verified = None
    if rank == 0 and world == 1:
        y = None
        L.fa_profile_enable(1)
        pair = {}
        for vf in (False, True):
            ms = []
            for _ in range(4):
                yv = fa.decode_flac_device(comp, st, nb, n_samp, verify=vf)
                ms.append(profile_read(L)[4])
            pair[vf] = float(np.mean(ms[1:]))
            assert torch.equal(yv, x)
            yv = None
        L.fa_profile_enable(0)
        decode_verified = {"ms": round(pair[True], 3), "Msamples_per_s": round(n_ch * n_samp / (pair[True] * 1e-3) / 1e6, 1),
                           "unverified_ms_same_loop": round(pair[False], 3), "overhead": round(pair[True] / pair[False] - 1.0, 4),
                           "what": "decode sequence with every frame's CRC-16 re-computed (K9 on a side stream beside K7), HIP events K6 .. both kernels done"}
        y = fa.decode_flac_device(comp, st, nb, n_samp)

    cfg5 = cfg3 = None
    if not args.no_extra:
        if world > 1:
            # configuration 5 on the sharded store: every rank keeps ITS shard resident (FlacArray + decode index) and
            # dist.route_slices sends every request to the GPU that owns its channel; 2 warm-ups, median of 5
            y = None
            store = fa.FlacArray.from_device_array(x, level=args.level)
            store_req = slice_requests(n_global, n_samp, 10000)
            times = []
            for r in range(7):
                sync()
                g0 = time.perf_counter()
                fdist.route_slices(store, store_req[0], store_req[1], store_req[2], n_global)
                sync()
                if r >= 2:
                    times.append(time.perf_counter() - g0)
            med = float(np.median(times))
            cfg5 = {"workload": f"10000 scattered slices of the {n_global}ch store: resident shard stores, dist.route_slices (owner routing), outputs to the host per GPU",
                    "ms_per_batch": round(med * 1e3, 3), "slices_per_s": round(10000 / med, 0), "timing": "median of 5 after 2 warm-ups, barrier to barrier"}
            store.release_device()
            del store
        elif rank == 0:
            cfg5 = bench_cfg5(torch, fa, comp, st, nb, n_ch, n_samp, x, dev)
            cfg5["small_reads"] = bench_small_reads(fa, comp, st, nb, n_ch, n_samp, x)

    if rank == 0:
        samples_per_step = n_ch * n_samp * world
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step / (elapsed / args.steps) / 1e6
        pm = np.mean(np.array(prof), axis=0)
        enc, cmpm, dec, enc_seq, dec_seq = (float(v) for v in pm[:5])
        n_local = n_ch * n_samp
        alg_gb = (4 + c_bytes) * n_local / 1e9  # algorithmic bytes of one direction: 4 + c per sample
        gbs = lambda ms: round(alg_gb / (ms * 1e-3), 1) if ms > 0 else None  # noqa: E731
        single_pass = cmpm <= 0  # the single-pass encoder ran: there is no compaction kernel
        enc_name = "encode_fused_kernel" if single_pass else "encode_frames_kernel"
        kernels = {
            enc_name: {"ms": round(enc, 3), "algorithmic_GBs": gbs(enc)},
            "decode_frames_kernel": {"ms": round(dec, 3), "algorithmic_GBs": gbs(dec)},
        }
        if not single_pass:
            kernels["compact_frames_kernel"] = {"ms": round(cmpm, 3), "algorithmic_GBs": round(2 * c_bytes * n_local / (cmpm * 1e-3) / 1e9, 1)}
        dom = max((enc_name, "decode_frames_kernel"), key=lambda k: kernels[k]["ms"])
        achieved = kernels[dom]["algorithmic_GBs"]
        issue_floor = None
        mix = measured_mix(dom)
        if mix and mix.get("waves"):
            # K3F: a wave per 4096-sample frame; K7: a wave per 64 frames -- either way waves scale with the samples
            waves = mix["waves"] * (n_local / float(4096 * (1 << 20)))
            issue_floor = round(mix["valu_per_wave"] * waves * 4.0 / 1024.0 / 2.4e9 * 1e3, 3)
        out = {
            "metric": "Msamples/s encode+decode, 4096ch x 1Msamp int32",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_ch}ch x {n_samp} int32 sinusoid+noise per GPU, FLAC level {args.level} (LPC order 8), encode then decode, HBM resident",
                "channels_per_gpu": n_ch,
                "samples_per_channel": n_samp,
                "level": args.level,
                "compressed_bytes_per_sample": round(c_bytes, 4),
                "parallelism": (f"channels sharded over {world} GPU(s), no data-path collective"
                                + ("" if world == 1 else ("; per step: all-gather of stream byte counts"
                                                          + (" + all-gather-v of the compressed blobs (global triple on every GPU)" if gather_state["on"] else "")))),
            },
            "roofline": {
                # achieved / peak / frac are HBM figures (the contract's roofline for this path); `bound` names what
                # actually limits the dominant kernel: the vector instruction stream ("issue": its own floor is
                # issue_floor_ms, i.e. frac cannot pass issue_ceiling_frac whatever the memory system does) or HBM
                "bound": "issue" if issue_floor is not None and issue_floor > alg_gb / HBM_PEAK_GBS * 1e3 else "hbm",
                "kernel": dom,
                "achieved": achieved,
                "algorithmic_GB_per_launch": round(alg_gb, 3),
                "ms_per_launch": kernels[dom]["ms"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(dom) if (n_ch, n_samp) == (4096, 1 << 20) else None,
                "traffic_unit": "GB per launch; constant from the newest committed PMC pass (profiles/r*_traffic.json), not measured in this run",
                "algorithmic_bytes_per_sample": round(4 + c_bytes, 4),
                # What limits the kernels is not HBM.  K3F: vector issue at three waves per SIMD (`issue`: share of a wave's
                # life issuing vector instructions x waves per SIMD, from the newest committed PMC pass), plus the wait for the
                # frame's byte offset.  K7: neither -- 12 % fewer vector instructions changed nothing; its per-lane read
                # shape (64 cache lines per load instruction) and the latency of the Rice chain do (profiles/r03_k7_experiments.md)
                "limiter": "vector_issue" if dom != "decode_frames_kernel" else "per_lane_read_shape_and_latency",
                "issue": measured_issue(dom),
                # VALU per wave x waves x 4 cycles / (256 CUs x 4 SIMDs) / 2.4 GHz: the time the vector instructions of
                # one launch take if every SIMD issues one of them every 4 cycles (instruction counts from the newest
                # committed PMC pass, scaled to this run's number of frames)
                "issue_floor_ms": issue_floor,
                "issue_ceiling_frac": round(alg_gb / (issue_floor * 1e-3) / HBM_PEAK_GBS, 4) if issue_floor else None,
                # the unit is the whole sequence -- HIP events around begin..finish (K3+K4+K5, host gaps
                # included) and around K6+K7, on the launch stream
                "encode_sequence": {"ms": round(enc_seq, 3), "achieved": gbs(enc_seq), "frac": round(gbs(enc_seq) / HBM_PEAK_GBS, 4) if enc_seq > 0 else None},
                "decode_sequence": {"ms": round(dec_seq, 3), "achieved": gbs(dec_seq), "frac": round(gbs(dec_seq) / HBM_PEAK_GBS, 4) if dec_seq > 0 else None},
            },
            "kernels": kernels,
            # each direction on its own (sequence time, HIP events on the launch stream)
            "encode_Msamples_per_s": round(n_local * world / (enc_seq * 1e-3) / 1e6, 1) if enc_seq > 0 else None,
            "decode_Msamples_per_s": round(n_local * world / (dec_seq * 1e-3) / 1e6, 1) if dec_seq > 0 else None,
        }
        if decode_verified is not None:
            out["decode_verified"] = decode_verified
        if world > 1:
            ag = {"in_step": bool(gather_state["on"]), "overlapped_with_decode": bool(gather_state["on"] and backend == "nccl"),
                  "global_blob_bytes": gather_state["bytes"], "error": gather_state["error"]}
            if gather_state["on"] and gather_state["ms"]:
                # stream time of the blob transfers on rank 0 (side stream), and what that is per GPU in received bytes
                # (xGMI: one link per peer, all transfers of a step at once; ~76 GB/s per link and direction)
                ag["ms"] = round(float(np.mean(gather_state["ms"][-args.steps:])), 3)
                recv = gather_state["bytes"] - gather_state["local"]
                ag["received_GBs_per_gpu"] = round(recv / (ag["ms"] * 1e-3) / 1e9, 1) if ag["ms"] > 0 else None
                ag["per_link_GBs"] = round(ag["received_GBs_per_gpu"] / (world - 1), 1) if ag["received_GBs_per_gpu"] else None
                ag["per_link_frac_of_76GBs"] = round(ag["per_link_GBs"] / 76.0, 3) if ag["per_link_GBs"] else None
            out["allgatherv"] = ag
            if compute_only is not None:
                out["compute_only"] = compute_only
        if cfg5 is not None:
            out["cfg5"] = cfg5
    if not args.no_extra and world == 1:
        # configuration 3 needs the HBM the headline's tensors hold
        x = y = comp = st = nb = ws = None
        torch.cuda.empty_cache()
        L.fa_release_scratch()
        cfg3 = bench_cfg3(torch, fa, L, n_ch, n_samp, args.level, dev)
        out["cfg3"] = cfg3
        torch.cuda.empty_cache()
        L.fa_release_scratch()
        out["end_to_end"] = bench_end_to_end(fa, level=args.level)
        try:
            out["cookbook"] = bench_cookbook(fa)
        except Exception as e:  # (never takes the line down)
            out["cookbook"] = f"{type(e).__name__}: {e}"[:200]
        try:
            torch.cuda.empty_cache()
            L.fa_release_scratch()
            out["other_geometries"] = bench_other_geometries(torch, fa, dev, args.level)
        except Exception as e:  # (never takes the line down)
            out["other_geometries"] = f"{type(e).__name__}: {e}"[:200]
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(n_samp)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
