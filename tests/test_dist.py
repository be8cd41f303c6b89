"""Multi-process assembly of the global (compressed, starts, nbytes) triple with torch.distributed
(gloo, world_size 2, CPU tensors).  Per-rank encoding uses the oracle here (test only); on a node
of MI355X the same dist.py code runs over RCCL with the HIP encoder."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, sinusoid_noise_i32


def _worker(rank, world, port, n_ch, n_samp, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from flacarray_amd import dist as fdist
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = sinusoid_noise_i32(n_ch, n_samp, seed=42)
    lo, hi = fdist.shard_range(n_ch, world, rank)
    blob, st, nb = O.encode_i32(x[lo:hi], 5)
    g_blob, g_starts, g_nbytes = fdist.assemble_global(torch.from_numpy(blob), torch.from_numpy(nb), n_ch)
    full_blob, full_st, full_nb = O.encode_i32(x, 5)
    ok = (
        np.array_equal(g_blob.numpy(), full_blob)
        and np.array_equal(g_starts.numpy(), full_st)
        and np.array_equal(g_nbytes.numpy(), full_nb)
    )
    # every rank can decode any stream of the assembled store
    y = O.decode_i32(g_blob.numpy(), g_starts.numpy(), g_nbytes.numpy(), n_samp)
    ok = ok and np.array_equal(y, x)
    # the handle form bench.py uses at N > 1 (host tensors: the transfers finish inside the call)
    pend = fdist.assemble_global_async(torch.from_numpy(blob), torch.from_numpy(nb), n_ch)
    a_blob, a_starts, a_nbytes = pend.wait()
    ok = ok and torch.equal(a_blob, g_blob) and torch.equal(a_starts, g_starts) and torch.equal(a_nbytes, g_nbytes) and pend.elapsed_ms() is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_ch", [(2, 5), (2, 8)])
def test_assemble_global_gloo(world, n_ch):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ch, 6000, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


class _StubStore:
    """Stands in for a rank's FlacArray shard in the routing test (no GPU here): serves slices of a plain array."""

    def __init__(self, rows):
        self.rows = rows
        self.nstreams = rows.shape[0]
        self.typestr = "int32"

    def read_slices(self, streams, first, count, as_tensor=False):
        import torch

        parts = [self.rows[s, f : f + c] for s, f, c in zip(streams, first, count)]
        if as_tensor:
            flat = np.concatenate(parts) if parts else np.zeros(0, np.int32)
            return torch.from_numpy(flat), None
        return parts


def _route_worker(rank, world, port, n_ch, n_samp, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from flacarray_amd import dist as fdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = sinusoid_noise_i32(n_ch, n_samp, seed=9)
    lo, hi = fdist.shard_range(n_ch, world, rank)
    store = _StubStore(x[lo:hi])
    rng = np.random.default_rng(987654321)  # SURVEY 8(d) S4 recipe, scaled down
    n = 200
    ch = rng.integers(0, n_ch, n)
    cnt = rng.integers(1, 2000, n)
    first = np.array([rng.integers(0, n_samp - c + 1) for c in cnt])
    idx, outs = fdist.route_slices(store, ch, first, cnt, n_ch)
    ok = all(lo <= ch[i] < hi for i in idx) and all(np.array_equal(o, x[ch[i], first[i] : first[i] + cnt[i]]) for i, o in zip(idx, outs))
    allouts = fdist.route_slices(store, ch, first, cnt, n_ch, gather=True)
    ok = ok and all(np.array_equal(o, x[c, f : f + k]) for o, c, f, k in zip(allouts, ch, first, cnt))
    try:
        fdist.route_slices(store, np.array([n_ch]), np.array([0]), np.array([1]), n_ch)
        ok = False
    except RuntimeError:
        pass
    ret[rank] = (bool(ok), int(len(idx)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_ch", [(2, 5), (3, 7)])
def test_route_slices_gloo(world, n_ch):
    """cfg 5 routing: every request goes to the rank that owns its stream, exactly once, and the
    gathered outputs come back in request order on every rank."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_route_worker, args=(r, world, port, n_ch, 5000, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert sum(ret[r][1] for r in range(world)) == 200


def test_owner_of_inverts_shard_range():
    from flacarray_amd.dist import owner_of, shard_range

    for n, w in [(5, 2), (8, 2), (10, 3), (2, 4), (32768, 8), (7, 7), (1, 3), (4099, 8)]:
        o = owner_of(np.arange(n), n, w)
        for r in range(w):
            lo, hi = shard_range(n, w, r)
            assert np.all(o[lo:hi] == r)


def _world8_worker(rank, world, port, n_ch, n_samp, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from flacarray_amd import dist as fdist
    from oracle import oracle as O

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = sinusoid_noise_i32(n_ch, n_samp, seed=4)
    lo, hi = fdist.shard_range(n_ch, world, rank)
    blob, st, nb = O.encode_i32(x[lo:hi], 5)
    g_blob, g_starts, g_nbytes = fdist.assemble_global(torch.from_numpy(blob), torch.from_numpy(nb), n_ch)
    full_blob, full_st, full_nb = O.encode_i32(x, 5)
    ok = np.array_equal(g_blob.numpy(), full_blob) and np.array_equal(g_starts.numpy(), full_st) and np.array_equal(g_nbytes.numpy(), full_nb)
    # cfg 5 on the sharded store: owner routing of a replicated request table, outputs gathered back in request order
    store = _StubStore(x[lo:hi])
    rng = np.random.default_rng(987654321)
    n = 400
    ch = rng.integers(0, n_ch, n)
    cnt = rng.integers(1, n_samp + 1, n)
    first = np.array([rng.integers(0, n_samp - c + 1) for c in cnt])
    idx, outs = fdist.route_slices(store, ch, first, cnt, n_ch)
    ok = ok and all(lo <= ch[i] < hi for i in idx) and all(np.array_equal(o, x[ch[i], first[i] : first[i] + cnt[i]]) for i, o in zip(idx, outs))
    allouts = fdist.route_slices(store, ch, first, cnt, n_ch, gather=True)
    ok = ok and all(np.array_equal(o, x[c, f : f + k]) for o, c, f, k in zip(allouts, ch, first, cnt))
    ret[rank] = (bool(ok), int(len(idx)), int(hi - lo))
    dist.barrier()
    dist.destroy_process_group()


def test_world_of_eight_with_an_uneven_stream_count():
    """The shape of cfg 4 / cfg 5 at N = 8 (mpi.py:84-90,156-187): 32 771 streams do not divide by 8 -- np.array_split
    gives the first three ranks 4097 streams --, every rank's shard blob has its own size, and the assembled triple must
    still be the single-process encode byte for byte on every rank; requests are routed to the owners and come back in
    request order.  (gloo on CPU, short streams; RCCL runs the same dist.py code on the node.)"""
    import torch.multiprocessing as mp

    world, n_ch = 8, 32771
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, n_ch, 64, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert all(ret[r][0] for r in range(world))
    assert sum(ret[r][1] for r in range(world)) == 400
    assert [ret[r][2] for r in range(world)] == [4097, 4097, 4097, 4096, 4096, 4096, 4096, 4096]
