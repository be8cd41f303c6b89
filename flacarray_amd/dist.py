"""Multi-GPU sharding of the encode/decode path: one process per GPU, torch.distributed.

Streams are independent, so the leading axis is split into contiguous ranges exactly as the
reference's MPI layer does (np.array_split semantics, src/flacarray/mpi.py:84-90); every rank
encodes / decodes its own shard with no data-path collective.  The only exchange step is the
optional assembly of the global (compressed, stream_starts, stream_nbytes) triple:

  1. all-gather of the per-stream byte counts (fixed size per rank after padding) -- the
     analogue of `allgather(local_nbytes)` in mpi.py:177 -- followed by an exclusive scan, which
     reproduces `global_bytes` (mpi.py:156-187);
  2. an all-gather-v of the shard blobs.  RCCL has no AllGatherv; xGMI is a fully connected
     point-to-point mesh, so each rank posts one send and one receive per peer in a single
     batch (grouped ncclSend/ncclRecv), every transfer on its own link, written straight to
     the rank's offset in the global blob.

Works with backend "nccl" (= RCCL, device tensors) and "gloo" (CPU tensors; used by the tests).
"""
import numpy as np


def shard_range(n_stream, world_size, rank):
    """[lo, hi) of the leading-axis rows owned by `rank` (np.array_split semantics, mpi.py:84-90)."""
    base, rem = divmod(int(n_stream), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_counts(n_stream, world_size):
    return [shard_range(n_stream, world_size, r)[1] - shard_range(n_stream, world_size, r)[0] for r in range(world_size)]


def _settle(t):
    """Rule for every exchange in this module: a payload is handed to the backend only after the stream
    that produced it has drained, and collectives are issued against torch's CURRENT stream -- the stream
    the encode / decode kernels were launched on (libflacarray.py:_stream_ptr).  (assemble_global_async is the one
    deliberate exception: RCCL only, blobs on a side stream ordered behind the current one by an event.)
    The C entry points already end with a hipStreamSynchronize of that stream (the byte total goes to
    the host), so this costs nothing; it makes the ordering explicit instead of implied.  See DESIGN.md
    section 6 for what this rule has to do with the two-ranks-on-one-GPU rehearsal hang of round 1."""
    if t.is_cuda:
        import torch

        torch.cuda.current_stream(t.device).synchronize()
    return t


def gather_stream_nbytes(local_nbytes, n_stream_global, group=None):
    """All-gather the per-stream byte counts; returns (global nbytes int64[n_stream_global],
    global starts int64[n_stream_global], per-rank byte totals list)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = shard_counts(n_stream_global, world)
    maxc = max(counts)
    dev = local_nbytes.device
    _settle(local_nbytes)
    # RCCL moves device tensors; gloo gets the (small) table through the host, also when the data
    # lives on a GPU (its device-tensor all-gather does not complete when ranks share a device)
    cdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    padded = torch.zeros(maxc, dtype=torch.int64, device=cdev)
    padded[: local_nbytes.numel()] = local_nbytes.reshape(-1).to(cdev)
    gathered = torch.empty(world * maxc, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    gathered = gathered.reshape(world, maxc).to(dev)
    parts = [gathered[r, : counts[r]] for r in range(world)]
    g_nbytes = torch.cat(parts)
    g_starts = torch.cumsum(g_nbytes, 0) - g_nbytes  # exclusive scan == global_bytes(), mpi.py:181-186
    rank_bytes = [int(p.sum().item()) for p in parts]
    return g_nbytes, g_starts, rank_bytes


def all_gather_blobs(local_blob, rank_bytes, group=None):
    """All-gather-v of uint8 blobs: returns the concatenation over ranks on every rank.

    One batched round of point-to-point transfers (each rank -> every peer), receives landing
    directly at the rank's byte offset of the output.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    offs = np.concatenate([[0], np.cumsum(rank_bytes)]).astype(np.int64)
    dev = local_blob.device
    _settle(local_blob)
    if world > 1 and dist.get_backend(group) != "nccl" and dev.type != "cpu":
        # gloo: point-to-point transfers of host tensors
        return all_gather_blobs(local_blob.cpu(), rank_bytes, group).to(dev)
    out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
    out[int(offs[rank]) : int(offs[rank + 1])] = local_blob
    if world == 1:
        return out
    ops = []
    for step in range(1, world):
        dst = (rank + step) % world
        src = (rank - step) % world
        gdst = dist.get_global_rank(group, dst) if group is not None else dst
        gsrc = dist.get_global_rank(group, src) if group is not None else src
        if rank_bytes[rank] > 0:
            ops.append(dist.P2POp(dist.isend, local_blob, gdst, group))
        if rank_bytes[src] > 0:
            ops.append(dist.P2POp(dist.irecv, out[int(offs[src]) : int(offs[src + 1])], gsrc, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


class _PendingAssembly:
    """Handle of assemble_global_async: the byte counts are in, the blobs may still be travelling."""

    def __init__(self, blob, g_starts, g_nbytes, side, reqs, events):
        self._blob, self._starts, self._nbytes = blob, g_starts, g_nbytes
        self._side, self._reqs, self._events = side, reqs, events

    def wait(self):
        """(global blob, global starts, global nbytes); the CURRENT stream is ordered behind the transfers."""
        import torch

        for r in self._reqs:
            r.wait()
        self._reqs = []
        if self._side is not None:
            torch.cuda.current_stream(self._blob.device).wait_stream(self._side)
            self._side = None
        return self._blob, self._starts, self._nbytes

    def elapsed_ms(self):
        """Stream time of the blob transfers (None when they ran on the host: gloo)."""
        if not self._events:
            return None
        self._events[1].synchronize()
        return float(self._events[0].elapsed_time(self._events[1]))


def assemble_global_async(local_blob, local_nbytes, n_stream_global, group=None, agree=False):
    """assemble_global with the all-gather-v of the blobs issued on a side stream (RCCL): the caller can keep
    launching work that needs only its OWN shard -- the decode of the same step -- on the current stream and call
    `.wait()` where it needs the global triple.  The byte counts (32 KiB per rank) are gathered synchronously
    first: every rank needs them to size the receive buffer.  With gloo (CPU rehearsals) the transfers are host
    transfers and complete inside this call; the handle then just returns the result.

    agree=True: after the local preparations (the receive buffer -- the one step that can fail on one rank alone) the
    ranks all-reduce a failure flag and EVERY rank raises if any failed, BEFORE a transfer is queued: a rank that
    dropped out after the others had queued their batch would leave their next collective behind a transfer that never
    completes.  Costs one host synchronisation; meant for a rehearsal step (bench.py), not for the steady state."""
    import torch
    import torch.distributed as dist

    g_nbytes, g_starts, rank_bytes = gather_stream_nbytes(local_nbytes, n_stream_global, group)
    world = dist.get_world_size(group)
    if world == 1 or dist.get_backend(group) != "nccl" or not local_blob.is_cuda:
        return _PendingAssembly(all_gather_blobs(local_blob, rank_bytes, group), g_starts, g_nbytes, None, [], None)
    rank = dist.get_rank(group)
    dev = local_blob.device
    offs = np.concatenate([[0], np.cumsum(rank_bytes)]).astype(np.int64)
    cur = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    out, failure = None, None
    try:
        out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)  # (allocated on the current stream: freed there, too)
    except Exception as e:  # noqa: BLE001
        if not agree:
            raise
        failure = e
    if agree:
        flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()):
            raise RuntimeError("all-gather-v not started: a rank could not prepare its receive buffer"
                               + (f" (this rank: {type(failure).__name__}: {failure})" if failure is not None else ""))
    side.wait_stream(cur)  # the payload is complete and `out` exists before the side stream touches either
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    with torch.cuda.stream(side):
        ev[0].record()
        out[int(offs[rank]) : int(offs[rank + 1])] = local_blob
        ops = []
        for step in range(1, world):
            dst, src = (rank + step) % world, (rank - step) % world
            gdst = dist.get_global_rank(group, dst) if group is not None else dst
            gsrc = dist.get_global_rank(group, src) if group is not None else src
            if rank_bytes[rank] > 0:
                ops.append(dist.P2POp(dist.isend, local_blob, gdst, group))
            if rank_bytes[src] > 0:
                ops.append(dist.P2POp(dist.irecv, out[int(offs[src]) : int(offs[src + 1])], gsrc, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for r in reqs:
            r.wait()  # (RCCL: orders the SIDE stream behind the transfers, does not block the host)
        ev[1].record()
    local_blob.record_stream(side)
    out.record_stream(side)
    return _PendingAssembly(out, g_starts, g_nbytes, side, [], ev)


_SIDE = {}


def _side_stream(dev):
    import torch

    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def assemble_global(local_blob, local_nbytes, n_stream_global, group=None):
    """Global (compressed, stream_starts, stream_nbytes) on every rank, bit-identical to a
    single-process encode of the whole array."""
    g_nbytes, g_starts, rank_bytes = gather_stream_nbytes(local_nbytes, n_stream_global, group)
    blob = all_gather_blobs(local_blob, rank_bytes, group)
    return blob, g_starts, g_nbytes


def owner_of(streams, n_stream_global, world_size):
    """Rank that owns each GLOBAL flat stream index under shard_range (vectorised inverse of it)."""
    streams = np.asarray(streams, dtype=np.int64)
    base, rem = divmod(int(n_stream_global), int(world_size))
    cut = rem * (base + 1)  # the first `rem` ranks hold base+1 streams
    if base == 0:
        return streams.copy()  # fewer streams than ranks: stream i lives on rank i
    return np.where(streams < cut, streams // (base + 1), rem + (streams - cut) // base)


def route_slices(local_store, streams, first, count, n_stream_global, group=None, gather=False):
    """Owner-routed scattered decode over a sharded store (cfg 5).

    Every rank holds `local_store`, the FlacArray of ITS contiguous range of streams (shard_range; the
    reference distributes the leading axis the same way, mpi.py:84-90), ideally resident in HBM
    (FlacArray.to_device).  All ranks pass the same request table: `streams` are GLOBAL flat stream
    indices, `first` / `count` sample ranges -- the reference would serve it with one decode call per
    request (array.py:409-449).  Each request is decoded by the rank that owns its stream, in one
    batched launch per rank; no compressed byte moves between GPUs.

    gather=False: returns (idx, outs) -- the positions of this rank's requests in the table and their
    decoded arrays ("outputs returned to host per GPU").
    gather=True: additionally all-gathers the decoded samples (all-gather-v, same transport as
    all_gather_blobs) and returns the list of all outputs in request order on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    streams = np.asarray(streams, dtype=np.int64)
    first = np.asarray(first, dtype=np.int64)
    count = np.asarray(count, dtype=np.int64)
    if streams.size and (streams.min() < 0 or streams.max() >= n_stream_global):
        raise RuntimeError("route_slices: stream index outside the global array")
    lo, hi = shard_range(n_stream_global, world, rank)
    if local_store.nstreams != hi - lo:
        raise RuntimeError(f"route_slices: rank {rank} holds {local_store.nstreams} streams, its shard has {hi - lo}")
    owner = owner_of(streams, n_stream_global, world)
    idx = np.flatnonzero(owner == rank)
    if not gather:
        outs = local_store.read_slices(streams[idx] - lo, first[idx], count[idx]) if idx.size else []
        return idx, outs
    # decoded samples of every rank, concatenated in rank order, then cut back into request order
    if idx.size:
        flat, _ = local_store.read_slices(streams[idx] - lo, first[idx], count[idx], as_tensor=True)
    else:
        res = getattr(local_store, "_resident", None)
        dev = res["device"] if res else torch.device("cpu")
        flat = torch.zeros(0, dtype=getattr(torch, local_store.typestr), device=dev)
    esz = flat.element_size()
    per_rank = [int(count[owner == r].sum()) * esz for r in range(world)]
    blob = flat.view(torch.uint8) if world == 1 else all_gather_blobs(flat.view(torch.uint8), per_rank, group)
    allv = blob.view(flat.dtype).cpu().numpy()
    outs = [None] * streams.size
    base = 0
    for r in range(world):
        ridx = np.flatnonzero(owner == r)
        offs = base + np.concatenate([[0], np.cumsum(count[ridx])[:-1]]) if ridx.size else []
        for i, o in zip(ridx, offs):
            outs[i] = allv[int(o) : int(o) + int(count[i])]
        base += int(count[ridx].sum())
    return outs
