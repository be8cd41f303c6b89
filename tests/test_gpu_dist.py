"""Multi-rank paths with the HIP encoder / decoder in the loop (run on the one-GPU box: both ranks share
cuda:0 and talk over gloo; RCCL needs one GPU per rank, so its branch is exercised with a world of one).

  * two ranks HIP-encode their shards, dist.assemble_global builds the global triple; it must equal a
    single-process HIP encode of the whole array and the oracle's bytes (mpi.py:84-90,156-187);
  * two ranks serve a scattered request table from their resident shard stores through
    dist.route_slices (cfg 5): same samples as the single-process batched decode;
  * backend "nccl" (RCCL), world size 1: init with device_id, all-gather of byte counts, barrier,
    all-reduce -- the calls bench.py and dist.py make on a multi-GPU node.

The ranks are fresh child processes (multiprocessing spawn); nothing re-executes the pytest process.
"""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, sinusoid_noise_i32

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _requests(n_ch, n_samp, n, seed=987654321):
    rng = np.random.default_rng(seed)
    ch = rng.integers(0, n_ch, n)
    cnt = rng.integers(1, 8193, n)
    first = np.array([rng.integers(0, n_samp - c + 1) for c in cnt])
    return ch, first, cnt


def _worker(rank, world, port, n_ch, n_samp, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)  # a hang ends with every thread's stack, not a silent kill
    import torch
    import torch.distributed as dist

    import flacarray_amd as fa
    from flacarray_amd import dist as fdist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = sinusoid_noise_i32(n_ch, n_samp, seed=42)
    lo, hi = fdist.shard_range(n_ch, world, rank)
    comp, st, nb = fa.encode_flac_device(torch.from_numpy(x[lo:hi]).cuda(), level=5)
    g_blob, g_starts, g_nbytes = fdist.assemble_global(comp, nb.reshape(-1), n_ch)
    one_blob, one_st, one_nb = fa.encode_flac_device(torch.from_numpy(x).cuda(), level=5)
    # the same assembly through the handle bench.py uses (blobs travel under the decode of the local shard; with
    # gloo the transfers are host transfers that finish inside the call, with RCCL they run on a side stream)
    pend = fdist.assemble_global_async(comp, nb.reshape(-1), n_ch)
    y_local = fa.decode_flac_device(comp, st, nb, n_samp)  # needs only the local blob
    a_blob, a_starts, a_nbytes = pend.wait()
    ok_async = torch.equal(a_blob, g_blob) and torch.equal(a_starts, g_starts) and torch.equal(a_nbytes, g_nbytes)
    ok_async = ok_async and np.array_equal(y_local.cpu().numpy(), x[lo:hi]) and pend.elapsed_ms() is None
    ok = (
        ok_async
        and g_blob.is_cuda
        and torch.equal(g_blob, one_blob)
        and torch.equal(g_starts, one_st.reshape(-1))
        and torch.equal(g_nbytes, one_nb.reshape(-1))
    )
    # any rank decodes any stream of the assembled store
    ok = ok and np.array_equal(fa.decode_flac_device(g_blob, g_starts, g_nbytes, n_samp).cpu().numpy(), x)
    # cfg 5: scattered requests routed to the owner of each stream
    store = fa.FlacArray.from_device_array(torch.from_numpy(x[lo:hi]).cuda())
    ch, first, cnt = _requests(n_ch, n_samp, 300)
    idx, outs = fdist.route_slices(store, ch, first, cnt, n_ch)
    ok = ok and all(lo <= ch[i] < hi for i in idx)
    ok = ok and all(np.array_equal(o, x[ch[i], first[i] : first[i] + cnt[i]]) for i, o in zip(idx, outs))
    allouts = fdist.route_slices(store, ch, first, cnt, n_ch, gather=True)
    ok = ok and len(allouts) == 300 and all(np.array_equal(o, x[c, f : f + k]) for o, c, f, k in zip(allouts, ch, first, cnt))
    ret[rank] = (bool(ok), g_blob.cpu().numpy().tobytes() if rank == 0 else None, int(len(idx)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_ch", [5, 8])
def test_two_ranks_hip_encode_assemble_and_route(n_ch, oracle):
    import torch.multiprocessing as mp

    world, n_samp = 2, 30000
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ch, n_samp, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert ret[0][2] + ret[1][2] == 300  # every request was served by exactly one rank
    # ... and the assembled bytes are the oracle's bytes for the whole array
    x = sinusoid_noise_i32(n_ch, n_samp, seed=42)
    blob_o, _, _ = oracle.encode_i32(x, 5)
    assert ret[0][1] == blob_o.tobytes()


def _nccl_worker(port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)
    import torch
    import torch.distributed as dist

    import flacarray_amd as fa
    from flacarray_amd import dist as fdist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = sinusoid_noise_i32(6, 20000, seed=3)
    comp, st, nb = fa.encode_flac_device(torch.from_numpy(x).to(dev), level=5)
    g_nb, g_st, rank_bytes = fdist.gather_stream_nbytes(nb.reshape(-1), 6)  # device tensors through RCCL
    blob, g_st2, g_nb2 = fdist.assemble_global(comp, nb.reshape(-1), 6)
    blob3, g_st3, g_nb3 = fdist.assemble_global_async(comp, nb.reshape(-1), 6).wait()
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    ok = (
        g_nb.is_cuda and torch.equal(g_nb, nb.reshape(-1)) and torch.equal(g_st, st.reshape(-1)) and rank_bytes == [comp.numel()]
        and torch.equal(blob, comp) and torch.equal(g_st2, st.reshape(-1)) and float(t.item()) == 1.5
        and torch.equal(blob3, comp) and torch.equal(g_st3, g_st2) and torch.equal(g_nb3, g_nb2)
    )
    ret[0] = bool(ok)
    dist.destroy_process_group()


def test_rccl_branch_world_of_one():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), ret))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    assert ret.get(0) is True
