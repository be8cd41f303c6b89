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
