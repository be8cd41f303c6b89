"""HDF5 "format version 1" layout of a compressed array (reference: src/flacarray/hdf5.py:30-245
write side, src/flacarray/hdf5_load_v1.py:93-375 read side, names hdf5_utils.py `hdf5_names`).

A group holds
    attrs   flacarray_format_version = "1", flacarray_software_version, flac_channels = "1" | "2"
    stream_starts  int64[leading]   attr stream_size
    stream_bytes   int64[leading]
    stream_offsets, stream_gains    float32[leading] (float data only)
    compressed     uint8[total]
so stores written here are readable by stock flacarray and vice versa.  The functions are duck
typed on the h5py.Group API (attrs mapping, create_dataset, item access): h5py itself is only
needed to open files.  Single-process I/O; distributed arrays are assembled first with
flacarray_amd.dist (the reference funnels every rank through rank 0, hdf5.py:262-291).
"""
import numpy as np

from . import __version__ as _version
from .compress import array_compress
from .decompress import array_decompress
from .utils import function_timer, keep_select, select_keep_indices

hdf5_names = {
    "compressed": "compressed",
    "stream_starts": "stream_starts",
    "stream_bytes": "stream_bytes",
    "stream_size": "stream_size",
    "stream_offsets": "stream_offsets",
    "stream_gains": "stream_gains",
    "flac_channels": "flac_channels",
}


def _no_mpi(mpi_comm):
    if mpi_comm is not None and getattr(mpi_comm, "size", 1) > 1:
        raise NotImplementedError(
            "distributed HDF5 I/O is not part of this build: gather the triple with flacarray_amd.dist.assemble_global "
            "and write it from one process"
        )


def _create(grp, name, shape, dtype):
    """h5py.Group / zarr-2 Group: create_dataset; zarr-3 Group: create_array (zarr.py:236-241)."""
    shape = tuple(int(x) for x in shape)
    if hasattr(grp, "create_array"):
        return grp.create_array(name, shape=shape, dtype=dtype)
    return grp.create_dataset(name, shape=shape, dtype=dtype)


def _one_element(arr, dtype):
    return np.asarray(arr, dtype=dtype).reshape(-1) if np.ndim(arr) == 0 else np.asarray(arr, dtype=dtype)


@function_timer
def write_compressed(
    hgrp,
    leading_shape,
    global_leading_shape,
    stream_size,
    stream_starts,
    global_stream_starts,
    stream_nbytes,
    stream_offsets,
    stream_gains,
    compressed,
    n_channels,
    local_nbytes=None,
    global_nbytes=None,
    global_process_nbytes=None,
    mpi_comm=None,
    mpi_dist=None,
):
    """Write an already compressed array to a group (hdf5.py:30-291; same positional arguments)."""
    _no_mpi(mpi_comm)
    shape = tuple(leading_shape) if len(leading_shape) > 0 else (1,)  # a single stream is stored as (1,)
    if shape != (tuple(global_leading_shape) if len(global_leading_shape) > 0 else (1,)):
        raise RuntimeError("single-process write: local and global leading shapes must agree")
    starts = _one_element(global_stream_starts if global_stream_starts is not None else stream_starts, np.int64).reshape(shape)
    nbytes = _one_element(stream_nbytes, np.int64).reshape(shape)
    comp = np.asarray(compressed, dtype=np.uint8).reshape(-1)
    fdt = np.float64 if int(n_channels) == 2 else np.float32

    hgrp.attrs["flacarray_format_version"] = "1"
    hgrp.attrs["flacarray_software_version"] = _version
    hgrp.attrs[hdf5_names["flac_channels"]] = f"{int(n_channels)}"

    dstarts = _create(hgrp, hdf5_names["stream_starts"], shape, np.int64)
    dstarts.attrs[hdf5_names["stream_size"]] = int(stream_size)
    dstarts[...] = starts
    dbytes = _create(hgrp, hdf5_names["stream_bytes"], shape, np.int64)
    dbytes[...] = nbytes
    if stream_offsets is not None:
        off = _one_element(stream_offsets, fdt).reshape(shape)
        d = _create(hgrp, hdf5_names["stream_offsets"], shape, off.dtype)
        d[...] = off
    if stream_gains is not None:
        gn = _one_element(stream_gains, fdt).reshape(shape)
        d = _create(hgrp, hdf5_names["stream_gains"], shape, gn.dtype)
        d[...] = gn
    dcomp = _create(hgrp, hdf5_names["compressed"], (comp.shape[0],), np.uint8)
    dcomp[...] = comp


@function_timer
def write_array(arr, hgrp, level=5, quanta=None, precision=None, mpi_comm=None, use_threads=False):
    """Compress a numpy array and write it to a group (hdf5.py:294-378)."""
    _no_mpi(mpi_comm)
    n_channels = 2 if arr.dtype in (np.dtype(np.int64), np.dtype(np.float64)) else 1
    compressed, starts, nbytes, offsets, gains = array_compress(
        arr, level=level, quanta=quanta, precision=precision, use_threads=use_threads
    )
    leading_shape = (1,) if arr.ndim == 1 else arr.shape[:-1]
    write_compressed(
        hgrp, leading_shape, leading_shape, arr.shape[-1], starts, starts, nbytes, offsets, gains, compressed, n_channels
    )


def _read_all(dset):
    return np.asarray(dset[...])


@function_timer
def read_compressed(hgrp, keep=None, mpi_comm=None, mpi_dist=None):
    """Load the compressed representation from a group (hdf5.py:381-440 dispatch + hdf5_load_v1.py:93-279).

    Returns the reference's 10-tuple (local_shape, global_shape, compressed, n_channel, stream_starts,
    stream_nbytes, stream_offsets, stream_gains, mpi_dist, keep_indices).  With `keep` only the
    selected streams' bytes are read and the auxiliary arrays become 1-D over the kept streams.
    """
    _no_mpi(mpi_comm)
    ver = int(hgrp.attrs["flacarray_format_version"]) if "flacarray_format_version" in hgrp.attrs else 0
    if ver != 1:
        raise RuntimeError(f"Unsupported flacarray HDF5 format version {ver} (this build reads version 1)")
    n_channel = int(hgrp.attrs[hdf5_names["flac_channels"]])
    dstarts = hgrp[hdf5_names["stream_starts"]]
    stream_size = int(dstarts.attrs[hdf5_names["stream_size"]])
    global_shape = tuple(dstarts.shape) + (stream_size,)
    raw_starts = _read_all(dstarts).astype(np.int64)
    raw_nbytes = _read_all(hgrp[hdf5_names["stream_bytes"]]).astype(np.int64)
    raw_offsets = _read_all(hgrp[hdf5_names["stream_offsets"]]) if hdf5_names["stream_offsets"] in hgrp else None
    raw_gains = _read_all(hgrp[hdf5_names["stream_gains"]]) if hdf5_names["stream_gains"] in hgrp else None
    dcomp = hgrp[hdf5_names["compressed"]]
    if mpi_dist is None:
        mpi_dist = [(0, global_shape[0])]

    if keep is None:
        # one contiguous read of every stream (io_common.py:35-50)
        total = int(np.sum(raw_nbytes))
        if total == 0:
            return (None, global_shape, None, n_channel, None, None, None, None, mpi_dist, None)
        first = int(raw_starts.reshape(-1)[0])
        compressed = np.asarray(dcomp[first : first + total], dtype=np.uint8)
        local_starts = raw_starts - first
        indices = None
        stream_nbytes, stream_offsets, stream_gains = raw_nbytes, raw_offsets, raw_gains
    else:
        # one read per kept stream into a packed buffer (io_common.py:51-74)
        starts, nbytes, indices = keep_select(np.asarray(keep, dtype=bool), raw_starts, raw_nbytes)
        if len(starts) == 0:
            return (None, global_shape, None, n_channel, None, None, None, None, mpi_dist, None)
        local_starts = np.zeros_like(starts)
        local_starts[1:] = np.cumsum(nbytes)[:-1]
        compressed = np.empty(int(np.sum(nbytes)), dtype=np.uint8)
        for i in range(len(starts)):
            compressed[local_starts[i] : local_starts[i] + nbytes[i]] = dcomp[int(starts[i]) : int(starts[i] + nbytes[i])]
        stream_nbytes = nbytes
        stream_offsets = select_keep_indices(raw_offsets, indices)
        stream_gains = select_keep_indices(raw_gains, indices)
    local_shape = tuple(local_starts.shape) + (stream_size,)
    return (
        local_shape, global_shape, compressed, n_channel, local_starts, stream_nbytes, stream_offsets, stream_gains,
        mpi_dist, indices,
    )


@function_timer
def read_array(
    hgrp, keep=None, stream_slice=None, keep_indices=False, mpi_comm=None, mpi_dist=None, use_threads=False, no_flatten=False
):
    """Read a group and decompress it (hdf5_load_v1.py:281-375)."""
    (local_shape, global_shape, compressed, n_channel, stream_starts, stream_nbytes, stream_offsets, stream_gains, mpi_dist,
     indices) = read_compressed(hgrp, keep=keep, mpi_comm=mpi_comm, mpi_dist=mpi_dist)
    first_samp = None
    last_samp = None
    if stream_slice is not None:
        if stream_slice.step is not None and stream_slice.step != 1:
            raise RuntimeError("Only stream slices with a step size of 1 are supported")
        first_samp = stream_slice.start
        last_samp = stream_slice.stop
    arr = array_decompress(
        compressed,
        local_shape[-1],
        stream_starts,
        stream_nbytes,
        stream_offsets=stream_offsets,
        stream_gains=stream_gains,
        first_stream_sample=first_samp,
        last_stream_sample=last_samp,
        is_int64=(n_channel == 2),
        use_threads=use_threads,
        no_flatten=no_flatten,
    )
    if keep_indices:
        return arr, indices
    return arr
