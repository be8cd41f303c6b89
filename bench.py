#!/usr/bin/env python3
"""bench.py -- Msamples/s encode+decode of a 4096ch x 1Msamp int32 array per MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: encode the HBM-resident int32 matrix
(K3 analyse+pack, K4 scan, K5 headers+compaction+CRC) into the reference's
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
    """HBM bytes per launch of `kernel` from the most recent committed PMC measurement
    (profiles/r*_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, corrected as MI355X_MICROARCH.md prescribes); None if there is none."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
        for name, rec in data["kernels"].items():
            if kernel in name:
                return round(rec["hbm_bytes"] / 1e9, 3)
    except Exception:
        return None
    return None


def cpu_baseline(n_samp, seconds_budget=25.0):
    """Time the CPU oracle (a port, not libFLAC: libFLAC is absent from this image) on a bounded
    sample of the same workload with every host core."""
    from oracle import oracle as O

    # a one-GPU box gives this process a CPU share of 16 cores whatever the host's core count is
    threads = min(O.lib().oracle_num_threads(), int(os.environ.get("FA_BENCH_CPU_THREADS", "16")))
    O.lib().oracle_set_threads(threads)
    n_ch = 16 * threads  # ~15-20 s of CPU work at 16 threads
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
    return {
        "value": round(x.size / (t2 - t0) / 1e6, 2),
        "unit": "Msamples/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"{n_ch}ch x {n_samp} int32 sinusoid+noise, level 5, encode {x.size/(t1-t0)/1e6:.1f} + decode {x.size/(t2-t1)/1e6:.1f} Msamples/s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL all-gather-v of the blobs (reported separately)")
    args = ap.parse_args()

    if os.environ.get("FA_BENCH_WATCHDOG_S"):  # rehearsals: dump every thread's stack and exit instead of hanging
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["FA_BENCH_WATCHDOG_S"]), exit=True)

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
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("FA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    L = _lib.lib()

    n_ch, n_samp = args.channels, args.samples
    x = make_data(torch, n_ch, n_samp, 123456789 + rank, dev)
    ws = EncodeWorkspace()
    n_global = n_ch * world

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        comp, st, nb = fa.encode_flac_device(x, level=args.level, workspace=ws)
        if world > 1:
            # global stream_starts: all-gather of the per-stream byte counts + exclusive scan
            fdist.gather_stream_nbytes(nb.reshape(-1), n_global)
        y = fa.decode_flac_device(comp, st, nb, n_samp)
        return comp, st, nb, y

    for _ in range(args.warmup):
        step()
    sync()
    L.fa_profile_enable(1)
    enc_ms, cmp_ms, dec_ms = [], [], []
    t0 = time.perf_counter()
    comp = st = nb = y = None
    for _ in range(args.steps):
        comp = st = nb = y = None  # hand the previous outputs back to the caching allocator (no hipMalloc in the timed region)
        comp, st, nb, y = step()
        ms = (ctypes.c_float * 3)()
        L.fa_profile_last(ms)
        enc_ms.append(ms[0]); cmp_ms.append(ms[1]); dec_ms.append(ms[2])
    sync()
    t1 = time.perf_counter()
    L.fa_profile_enable(0)
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # correctness of what was timed (outside the timed region)
    assert torch.equal(y, x), "decode(encode(x)) != x"
    c_bytes = comp.numel() / x.numel()

    gather_s = None
    if args.gather and world > 1:
        sync()
        g0 = time.perf_counter()
        fdist.assemble_global(comp, nb.reshape(-1), n_global)
        sync()
        gather_s = time.perf_counter() - g0

    if rank == 0:
        samples_per_step = n_ch * n_samp * world
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step / (elapsed / args.steps) / 1e6
        enc = float(np.mean(enc_ms)); dec = float(np.mean(dec_ms)); cmpm = float(np.mean(cmp_ms))
        n_local = n_ch * n_samp
        kernels = {
            "encode_frames_kernel": {"ms": round(enc, 3), "algorithmic_GBs": round((4 + c_bytes) * n_local / (enc * 1e-3) / 1e9, 1)},
            "compact_frames_kernel": {"ms": round(cmpm, 3), "algorithmic_GBs": round(2 * c_bytes * n_local / (cmpm * 1e-3) / 1e9, 1)},
            "decode_frames_kernel": {"ms": round(dec, 3), "algorithmic_GBs": round((4 + c_bytes) * n_local / (dec * 1e-3) / 1e9, 1)},
        }
        dom = max(("encode_frames_kernel", "decode_frames_kernel"), key=lambda k: kernels[k]["ms"])
        achieved = kernels[dom]["algorithmic_GBs"]
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
                "parallelism": f"channels sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "algorithmic_GB_per_launch": round((4 + c_bytes) * n_local / 1e9, 3),
                "ms_per_launch": kernels[dom]["ms"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(dom) if (n_ch, n_samp) == (4096, 1 << 20) else None,
                "traffic_unit": "GB per launch (PMC, profiles/r*_traffic.json)",
                "algorithmic_bytes_per_sample": round(4 + c_bytes, 4),
            },
            "kernels": kernels,
            # each direction on its own (kernel time of K3+K5 / of K7, HIP events on the launch stream)
            "encode_Msamples_per_s": round(n_local * world / ((enc + cmpm) * 1e-3) / 1e6, 1),
            "decode_Msamples_per_s": round(n_local * world / (dec * 1e-3) / 1e6, 1),
        }
        if gather_s is not None:
            out["allgatherv_s"] = round(gather_s, 4)
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(n_samp)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
