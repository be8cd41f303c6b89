"""FlacArray container (reference: src/flacarray/array.py:19-884).

Holds the concatenated per-stream FLAC bytes plus the int64 `stream_starts` / `stream_nbytes`
index (and float32 offsets/gains for quantised float data) and decompresses numpy-style
selections on the fly through the MI355X decode path.  Same constructor arguments,
properties, `from_array` / `to_array` / `__getitem__` / `__eq__` semantics as the reference;
HDF5/Zarr I/O and the mpi4py distribution are outside this hot path (multi-GPU sharding lives
in flacarray_amd.dist).  `read_slices` is an addition: one batched launch for many scattered
(stream, sample-range) requests, which the reference can only serve one call at a time.
"""
import copy
from dataclasses import dataclass, field
from typing import Any, Optional, Tuple

import numpy as np

from .compress import array_compress
from .decompress import array_decompress_slice
from .utils import log

_KINDS = {"int32": (False, False), "int64": (True, False), "float32": (False, True), "float64": (True, True)}


@dataclass(frozen=True)
class _Store:
    """Everything a FlacArray knows about its compressed store, validated once and never mutated.

    The five arrays are the reference's triple (+ the two quantisation vectors); all other fields are derived
    from `shape` / `dtype` by `_Store.build`.  A single process holds the whole array here, so each `global_*`
    quantity equals its local twin (the reference fills them through mpi.py:93-187 when a communicator is given).
    """

    shape: Tuple[int, ...]            # the user's shape (a 1-D array stays 1-D)
    global_shape: Tuple[int, ...]
    dtype: np.dtype
    blob: Any                         # uint8, all streams back to back
    starts: Any                       # int64 over the leading shape
    nbytes_per_stream: Any            # int64 over the leading shape
    offsets: Optional[Any]            # float data only
    gains: Optional[Any]
    dist: Any = None
    # derived
    single: bool = field(default=False)          # the original was 1-D: one stream, results are flattened
    grid: Tuple[int, ...] = field(default=())     # shape with the stream axis made explicit
    kind: str = field(default="int32")
    wide: bool = field(default=False)             # 64-bit samples: two FLAC channels per stream

    @classmethod
    def build(cls, shape, global_shape, dtype, blob, starts, nbytes, offsets, gains, dist=None):
        dt = np.dtype(dtype)
        kind = next((k for k in _KINDS if dt == np.dtype(k)), None)
        if kind is None:
            raise RuntimeError(f"Unsupported dtype '{dt}'")
        shp = tuple(int(n) for n in shape)
        single = len(shp) == 1
        grid = (1,) + shp if single else shp
        gshape = tuple(int(n) for n in global_shape) if global_shape is not None else grid
        return cls(shp, gshape, dt, blob, starts, nbytes, offsets, gains, dist, single, grid, kind, _KINDS[kind][0])

    def clone(self):
        dup = copy.deepcopy
        return _Store.build(self.shape, self.global_shape, self.dtype, dup(self.blob), dup(self.starts), dup(self.nbytes_per_stream),
                            dup(self.offsets), dup(self.gains), dup(self.dist))

    # what the accessors below hand out
    lead = property(lambda s: s.grid[:-1])
    glead = property(lambda s: s.global_shape[:-1])
    samples = property(lambda s: s.grid[-1])
    blob_bytes = property(lambda s: int(s.blob.nbytes))
    count = property(lambda s: int(np.prod(s.grid[:-1], dtype=np.int64)))
    gcount = property(lambda s: int(np.prod(s.global_shape[:-1], dtype=np.int64)))


# public read-only attribute -> (field or derived property of _Store, one-line description)
_ACCESSORS = {
    "shape": ("shape", "Shape of the array this object decompresses to."),
    "global_shape": ("global_shape", "Shape across all processes (equal to the local one without a communicator)."),
    "leading_shape": ("lead", "Local shape without the compressed (last) axis."),
    "global_leading_shape": ("glead", "Global shape without the compressed axis."),
    "stream_size": ("samples", "Number of samples in every stream."),
    "nbytes": ("blob_bytes", "Size of the local compressed bytes."),
    "global_nbytes": ("blob_bytes", "Size of the compressed bytes of all processes."),
    "nstreams": ("count", "Number of local streams."),
    "global_nstreams": ("gcount", "Number of streams of all processes."),
    "compressed": ("blob", "uint8 array: every stream's FLAC bytes, back to back."),
    "stream_starts": ("starts", "int64 byte offset of each stream inside `compressed`."),
    "stream_nbytes": ("nbytes_per_stream", "int64 byte count of each stream."),
    "global_stream_starts": ("starts", "Stream offsets inside the global byte range."),
    "global_stream_nbytes": ("nbytes_per_stream", "Stream byte counts of all processes."),
    "stream_offsets": ("offsets", "Per-stream offset removed before quantisation (float data; else None)."),
    "stream_gains": ("gains", "Per-stream scale applied at quantisation (float data; else None)."),
    "mpi_dist": ("dist", "Ranges of the leading axis held by each process (None here)."),
    "dtype": ("dtype", "numpy dtype of the decompressed samples."),
    "typestr": ("kind", "dtype as one of 'int32', 'int64', 'float32', 'float64'."),
}


def _store_reader(attr, doc):
    return property(lambda self: getattr(self._st, attr), doc=doc)


class FlacArray:
    """FLAC compressed array representation; the last axis is the compressed one.

    Constructed directly only to copy (`FlacArray(other)`); use `from_array` otherwise.
    Keyword arguments are those of the reference constructor (array.py:78-90).
    """

    def __init__(self, other, shape=None, global_shape=None, compressed=None, dtype=None, stream_starts=None,
                 stream_nbytes=None, stream_offsets=None, stream_gains=None, mpi_comm=None, mpi_dist=None):
        if (other._comm if other is not None else mpi_comm) is not None:
            raise NotImplementedError("mpi4py communicators are not supported; see flacarray_amd.dist for multi-GPU sharding")
        self._comm = None
        self._st = other._st.clone() if other is not None else _Store.build(
            shape, global_shape, dtype, compressed, stream_starts, stream_nbytes, stream_offsets, stream_gains, mpi_dist)
        self._resident = None  # device copies of (compressed, starts, nbytes, offsets, gains): see to_device(); never copied

    mpi_comm = property(lambda self: self._comm, doc="Always None: distribution goes through flacarray_amd.dist.")
    global_process_nbytes = property(lambda self: [self._st.blob_bytes], doc="Compressed bytes held by each process.")

    # the names the rest of this class uses for the store's fields
    _shape = property(lambda self: self._st.shape)
    _global_shape = property(lambda self: self._st.global_shape)
    _local_shape = property(lambda self: self._st.grid)
    _leading_shape = property(lambda self: self._st.lead)
    _global_leading_shape = property(lambda self: self._st.glead)
    _stream_size = property(lambda self: self._st.samples)
    _flatten_single = property(lambda self: self._st.single)
    _compressed = property(lambda self: self._st.blob)
    _stream_starts = property(lambda self: self._st.starts)
    _global_stream_starts = property(lambda self: self._st.starts)
    _stream_nbytes = property(lambda self: self._st.nbytes_per_stream)
    _stream_offsets = property(lambda self: self._st.offsets)
    _stream_gains = property(lambda self: self._st.gains)
    _dtype = property(lambda self: self._st.dtype)
    _typestr = property(lambda self: self._st.kind)
    _is_int64 = property(lambda self: self._st.wide)
    _local_nbytes = property(lambda self: self._st.blob_bytes)

    # ---- numpy-style selection -> decode ----
    def _plan_selection(self, raw_key):
        """Turn a numpy-style key into (result shape, keep mask over the streams, first, last sample).

        Same results as the reference's key handling (array.py:297-407): integers drop their axis,
        slices keep it, an out-of-range integer on a leading axis gives an empty result instead of an
        IndexError, the stream axis takes an integer or a step-1 slice, and the streams are picked
        through a boolean mask (so a negative leading step does not reverse their order)."""
        key = raw_key if isinstance(raw_key, tuple) else (raw_key,)
        if self._flatten_single:  # a 1-D array: the user's key addresses the samples only
            if len(key) != 1:
                raise ValueError(f"Slice key {raw_key} is not valid for single, flattened stream.")
            key = (0,) + key
        ndim = len(self._local_shape)
        if len(key) > ndim:
            raise ValueError(f"Invalid slice key {raw_key}, too many dimensions")
        key = key + (slice(None),) * (ndim - len(key))
        *lead_key, samp_key = key

        # stream axis
        n = self._stream_size
        if samp_key is None:
            first, last, samp_shape = 0, n, (n,)
        elif isinstance(samp_key, slice):
            first, last, step = samp_key.indices(n)
            if step != 1:
                raise ValueError("Only stride==1 supported on stream slices")
            if last <= first:
                first, last = 0, 0
            samp_shape = (last - first,)
        elif isinstance(samp_key, (int, np.integer)):
            first, last, samp_shape = samp_key, samp_key + 1, ()
        else:
            raise ValueError("Stream dimension supports contiguous slices or single indices.")

        # leading axes
        lead_shape = []
        nothing = False
        for k, dim in zip(lead_key, self._leading_shape):
            if isinstance(k, (int, np.integer)):
                if not 0 <= k < dim:
                    lead_shape.append(0)
                    nothing = True
            else:
                lo, hi, st = k.indices(dim)
                lead_shape.append(len(range(lo, hi, st)))
        keep = None
        if len(lead_key) > 0:
            keep = np.zeros(self._leading_shape, dtype=bool)
            if not nothing:
                keep[tuple(lead_key)] = True
        return tuple(lead_shape) + samp_shape, keep, first, last

    # ---- HBM residency (addition to the reference API) ----
    def to_device(self, device=None):
        """Keep the compressed store resident in HBM: bytes, starts, nbytes (and offsets / gains) are uploaded
        once; `__getitem__`, `to_array` and `read_slices` then decode straight from those tensors and only the
        decoded samples cross PCIe.  The reference's usage pattern is many small reads from one store
        (array.py:409-449: one decode call per key); without residency every read re-uploads its byte span.
        Returns self."""
        import torch

        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        flat = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt).reshape(-1)).to(dev)  # noqa: E731
        res = {
            "device": dev,
            "compressed": torch.from_numpy(np.ascontiguousarray(self._compressed)).to(dev),
            "starts": flat(self._stream_starts, np.int64),
            "nbytes": flat(self._stream_nbytes, np.int64),
            "offsets": None,
            "gains": None,
        }
        if self._stream_offsets is not None:
            ft = np.float64 if self._is_int64 else np.float32
            res["offsets"] = flat(self._stream_offsets, ft)
            res["gains"] = flat(self._stream_gains, ft)
        self._resident = res
        return self

    def _index(self):
        """The store's decode index (stream headers parsed, frame offsets tabulated once), built on first use."""
        res = self._resident
        if res.get("index") is None:
            from .libflacarray import DeviceDecodeIndex

            res["index"] = DeviceDecodeIndex(res["compressed"], res["starts"], res["nbytes"], self._stream_size, is_int64=self._is_int64)
        return res["index"]

    def release_device(self):
        """Drop the HBM copy made by to_device() (and its decode index)."""
        if self._resident is not None and self._resident.get("index") is not None:
            self._resident["index"].close()
        self._resident = None
        return self

    @property
    def is_resident(self):
        return self._resident is not None

    def _decode_resident(self, keep, first, last, as_tensor=False):
        """Decode [first, last) (negative: everything) of the kept streams from the resident store.
        Returns (2-D result: kept streams x samples, list of kept multi-indices or None)."""
        import torch

        res = self._resident
        indices = None
        off, gain = res["offsets"], res["gains"]
        if keep is None:
            out = self._index().decode(first, last, offsets=off, gains=gain)
        else:
            if keep.shape != tuple(self._leading_shape):
                raise RuntimeError("The keep array should have the same shape as stream_starts")
            sel = np.flatnonzero(np.asarray(keep).reshape(-1))
            indices = list(zip(*(ax.tolist() for ax in np.unravel_index(sel, self._leading_shape))))
            f0, n = (0, self._stream_size) if (first < 0 or last < 0) else (first, last - first)
            if sel.size == 0:
                out = torch.zeros((0, n), dtype=getattr(torch, self._typestr), device=res["device"])
            else:
                # the kept streams as one batch of slices against the index (one launch, nothing re-parsed)
                flat, _ = self._index().decode_slices(sel, np.full(sel.size, f0, np.int64), np.full(sel.size, n, np.int64), offsets=off, gains=gain,
                                                      to_host=not as_tensor)  # (host result: copied inside the decode call)
                out = flat.reshape(sel.size, n)
                if not as_tensor:
                    return out, indices
        return (out if as_tensor else out.cpu().numpy()), indices

    def __getitem__(self, raw_key):
        """Decompress a selection on the fly; the result has numpy's shape for the same key."""
        shape, keep, first, last = self._plan_selection(raw_key)
        if 0 in shape:
            return np.zeros(shape, dtype=self._dtype)
        if self._resident is not None:
            arr, _ = self._decode_resident(keep, first, last)
            return arr.reshape(shape)
        arr, _ = self._decode_host(keep, first, last)
        return arr.reshape(shape)

    def _decode_host(self, keep, first, last, **extra):
        """array_decompress_slice (decompress.py:18) over this store: bytes go up, samples come back."""
        st = self._st
        return array_decompress_slice(st.blob, st.samples, st.starts, st.nbytes_per_stream, stream_offsets=st.offsets,
                                      stream_gains=st.gains, keep=keep, first_stream_sample=first, last_stream_sample=last,
                                      is_int64=st.wide, **extra)

    def __delitem__(self, key):
        raise RuntimeError("Cannot delete individual streams")

    def __setitem__(self, key, value):
        raise RuntimeError("Cannot modify individual byte streams")

    def __repr__(self):
        return f"<FlacArray {self._typestr} shape={self._shape} bytes={self._local_nbytes}>"

    def __eq__(self, other):
        if self._shape != other._shape or self._dtype != other._dtype or self._global_shape != other._global_shape:
            log.debug("FlacArray shape/dtype mismatch")
            return False
        if not np.array_equal(self._stream_starts, other._stream_starts):
            return False
        if not np.array_equal(self._compressed, other._compressed):
            return False
        for mine, theirs in ((self._stream_offsets, other._stream_offsets), (self._stream_gains, other._stream_gains)):
            if (mine is None) != (theirs is None):
                return False
            if mine is not None and not np.allclose(mine, theirs):
                return False
        return True

    def to_array(self, keep=None, stream_slice=None, keep_indices=False, use_threads=False):
        """Decompress into a numpy array (array.py:518-584).

        `stream_slice`: step-1 slice of samples taken from every stream (normalised with
        slice.indices(); the reference forwards raw start/stop, so negative values there decode
        the whole stream).  `keep`: bool mask over the leading shape; the result is then the
        2-D array of kept streams (and their indices if `keep_indices`).
        """
        span = None
        if stream_slice is not None:
            if stream_slice.step not in (None, 1):
                raise RuntimeError("Only stream slices with a step size of 1 are supported")
            span = stream_slice.indices(self._stream_size)[:2]
        if self._resident is not None:
            lo, hi = span if span is not None else (-1, -1)
            if lo >= 0 and hi <= lo:
                raise RuntimeError("first_sample is larger than last_sample")
            arr, indices = self._decode_resident(keep, lo, hi)
            if keep is None:
                arr = arr.reshape(-1) if self._flatten_single else arr.reshape(self._shape[:-1] + (arr.shape[-1],))
        else:
            lo, hi = span if span is not None else (None, None)
            arr, indices = self._decode_host(keep, lo, hi, use_threads=use_threads, no_flatten=not self._flatten_single)
        return (arr, indices) if (keep is not None and keep_indices) else arr

    def read_slices(self, streams, first, count, as_tensor=False):
        """Batched random access (addition to the reference API).

        streams: flat (C-order) stream indices; first/count: sample ranges.  Returns a list of
        1-D arrays, one per request, decoded with ONE kernel launch on the GPU (as_tensor: the flat
        device tensor and the int64 array of its per-request offsets instead).  On a store made
        resident with to_device() nothing but the request table is uploaded.
        """
        import torch

        from .libflacarray import decode_slices_device

        res = self._resident
        if res is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            comp = torch.from_numpy(np.ascontiguousarray(self._compressed)).to(dev)
            st = torch.from_numpy(np.ascontiguousarray(self._stream_starts).reshape(-1)).to(dev)
            nb = torch.from_numpy(np.ascontiguousarray(self._stream_nbytes).reshape(-1)).to(dev)
            off = gain = None
            if self._stream_offsets is not None:
                off = torch.from_numpy(np.ascontiguousarray(self._stream_offsets).reshape(-1))
                gain = torch.from_numpy(np.ascontiguousarray(self._stream_gains).reshape(-1))
        if res is not None:
            out, out_off = self._index().decode_slices(streams, first, count, offsets=res["offsets"], gains=res["gains"], to_host=not as_tensor)
            if not as_tensor:
                count = np.asarray(count, dtype=np.int64)
                return [out[o : o + c] for o, c in zip(out_off, count)]
        else:
            out, out_off = decode_slices_device(
                comp, st, nb, self._stream_size, streams, first, count, offsets=off, gains=gain, is_int64=self._is_int64
            )
        if as_tensor:
            return out, out_off
        flat = out.cpu().numpy()
        count = np.asarray(count, dtype=np.int64)
        return [flat[o : o + c] for o, c in zip(out_off, count)]

    @classmethod
    def from_device_array(cls, data, level=5, quanta=None):
        """Construct a RESIDENT FlacArray from a torch tensor that already lives in HBM (int32 / int64, or
        float32 with per-stream `quanta`): quantise + encode on the device, keep the store there, and mirror
        it to host arrays so that every property of the reference API still answers with numpy."""
        import torch

        from .libflacarray import encode_flac_device, encode_flac_device_f32

        offsets = gains = None
        if data.dtype == torch.float32:
            if quanta is None:
                raise RuntimeError("Compressing floating point data ('float32') requires specifying either quanta or precision.")
            lead = tuple(data.shape[:-1]) if data.dim() > 1 else (1,)
            q = torch.as_tensor(quanta, dtype=torch.float32, device=data.device)
            q = q.expand(lead).contiguous() if q.dim() == 0 else q
            comp, st, nb, offsets, gains = encode_flac_device_f32(data.contiguous(), q, level=level, compact=True)
        elif data.dtype in (torch.int32, torch.int64):
            comp, st, nb = encode_flac_device(data.contiguous(), level=level, compact=True)
        else:
            raise ValueError(f"Unsupported data type '{data.dtype}'")
        host = lambda t: None if t is None else t.cpu().numpy()  # noqa: E731
        out = cls._assemble(tuple(data.shape), None, np.dtype(str(data.dtype).replace("torch.", "")), host(comp), host(st), host(nb),
                            host(offsets), host(gains))
        out._resident = {
            "device": data.device, "compressed": comp, "starts": st.reshape(-1), "nbytes": nb.reshape(-1),
            "offsets": None if offsets is None else offsets.reshape(-1), "gains": None if gains is None else gains.reshape(-1),
        }
        return out

    @classmethod
    def from_array(cls, arr, level=5, quanta=None, precision=None, mpi_comm=None, use_threads=False):
        """Construct a FlacArray from a numpy ndarray (array.py:587-637)."""
        if mpi_comm is not None:
            raise NotImplementedError("mpi4py communicators are not supported; see flacarray_amd.dist")
        pieces = array_compress(arr, level=level, quanta=quanta, precision=precision, use_threads=use_threads)
        return cls._assemble(arr.shape, None, arr.dtype, *pieces)

    @classmethod
    def _assemble(cls, shape, global_shape, dtype, blob, starts, nbytes, offsets, gains):
        """A FlacArray around an existing (bytes, starts, nbytes, offsets, gains) store; without a global shape the
        array is the whole array (global_array_properties, mpi.py:109-117: a 1-D array counts as one stream)."""
        shape = tuple(shape)
        if global_shape is None:
            global_shape = (1,) + shape if len(shape) == 1 else shape
        return cls(None, shape=shape, global_shape=global_shape, compressed=blob, dtype=dtype, stream_starts=starts,
                   stream_nbytes=nbytes, stream_offsets=offsets, stream_gains=gains)

    def write_hdf5(self, hgrp):
        """Write the compressed representation to an open HDF5 group (array.py:639-682), format
        version 1 (flacarray_amd/hdf5.py)."""
        from .hdf5 import write_compressed

        st = self._st
        write_compressed(hgrp, st.lead, st.glead, st.samples, st.starts, st.starts, st.nbytes_per_stream, st.offsets, st.gains,
                         st.blob, 2 if st.wide else 1)

    @classmethod
    def read_hdf5(cls, hgrp, keep=None, mpi_comm=None, mpi_dist=None, no_flatten=False):
        """Construct a FlacArray from an HDF5 group (array.py:684-764).  With `keep` the array
        holds only the selected streams, as a 2-D (n_kept, stream_size) array."""
        from .hdf5 import read_compressed
        from .utils import compressed_dtype

        got = read_compressed(hgrp, keep=keep, mpi_comm=mpi_comm, mpi_dist=mpi_dist)
        local, whole, blob, n_channels, starts, nbytes, offsets, gains = got[:8]
        if len(local) == 2 and local[0] == 1 and not no_flatten:
            local = (local[1],)  # a single stream reads back as the 1-D array it was written from
        return cls._assemble(local, whole, compressed_dtype(n_channels, offsets, gains), blob, starts, nbytes, offsets, gains)

    def write_zarr(self, zgrp):
        """Write the compressed representation to an open Zarr group (array.py:766-804); same schema as HDF5."""
        self.write_hdf5(zgrp)

    @classmethod
    def read_zarr(cls, zgrp, keep=None, mpi_comm=None, mpi_dist=None, no_flatten=False):
        """Construct a FlacArray from a Zarr group (array.py:806-884)."""
        return cls.read_hdf5(zgrp, keep=keep, mpi_comm=mpi_comm, mpi_dist=mpi_dist, no_flatten=no_flatten)


for _name, (_attr, _doc) in _ACCESSORS.items():
    setattr(FlacArray, _name, _store_reader(_attr, _doc))
del _name, _attr, _doc
