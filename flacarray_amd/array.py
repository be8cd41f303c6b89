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

import numpy as np

from .compress import array_compress
from .decompress import array_decompress_slice
from .utils import log


class FlacArray:
    """FLAC compressed array representation; the last axis is the compressed one.

    Constructed directly only to copy (`FlacArray(other)`); use `from_array` otherwise.
    """

    def __init__(
        self,
        other,
        shape=None,
        global_shape=None,
        compressed=None,
        dtype=None,
        stream_starts=None,
        stream_nbytes=None,
        stream_offsets=None,
        stream_gains=None,
        mpi_comm=None,
        mpi_dist=None,
    ):
        if other is not None:
            self._shape = copy.deepcopy(other._shape)
            self._global_shape = copy.deepcopy(other._global_shape)
            self._compressed = copy.deepcopy(other._compressed)
            self._dtype = np.dtype(other._dtype)
            self._stream_starts = copy.deepcopy(other._stream_starts)
            self._stream_nbytes = copy.deepcopy(other._stream_nbytes)
            self._stream_offsets = copy.deepcopy(other._stream_offsets)
            self._stream_gains = copy.deepcopy(other._stream_gains)
            self._mpi_dist = copy.deepcopy(other._mpi_dist)
            self._mpi_comm = other._mpi_comm  # (a copy starts without the HBM mirror of `other`)
        else:
            self._shape = tuple(shape)
            if global_shape is not None:
                self._global_shape = tuple(global_shape)
            else:  # single process: mpi.py:109-117 (a 1-D array is one stream: global shape (1, n))
                self._global_shape = (1, self._shape[0]) if len(self._shape) == 1 else self._shape
            self._compressed = compressed
            self._dtype = np.dtype(dtype)
            self._stream_starts = stream_starts
            self._stream_nbytes = stream_nbytes
            self._stream_offsets = stream_offsets
            self._stream_gains = stream_gains
            self._mpi_comm = mpi_comm
            self._mpi_dist = mpi_dist
        if self._mpi_comm is not None:
            raise NotImplementedError("mpi4py communicators are not supported; see flacarray_amd.dist for multi-GPU sharding")
        self._resident = None  # device copies of (compressed, starts, nbytes, offsets, gains): see to_device()
        self._init_params()

    def _init_params(self):
        # a 1-D original keeps its flattened shape; internally there is always a stream axis
        if len(self._shape) == 1:
            self._flatten_single = True
            self._local_shape = (1, self._shape[0])
        else:
            self._flatten_single = False
            self._local_shape = self._shape
        self._local_nbytes = self._compressed.nbytes
        self._global_nbytes = self._local_nbytes
        self._global_proc_nbytes = [self._local_nbytes]
        self._global_stream_starts = self._stream_starts
        self._global_stream_nbytes = self._stream_nbytes
        self._leading_shape = self._local_shape[:-1]
        self._global_leading_shape = self._global_shape[:-1]
        self._stream_size = self._local_shape[-1]
        self._local_nstreams = int(np.prod(self._leading_shape))
        self._global_nstreams = int(np.prod(self._global_leading_shape)) if len(self._global_leading_shape) else 1
        self._typestr = self._dtype_str(self._dtype)
        self._is_int64 = self._dtype == np.dtype(np.int64) or self._dtype == np.dtype(np.float64)

    @staticmethod
    def _dtype_str(dt):
        for name in ("float64", "float32", "int64", "int32"):
            if dt == np.dtype(name):
                return name
        raise RuntimeError(f"Unsupported dtype '{dt}'")

    # ---- shapes of the decompressed array ----
    @property
    def shape(self):
        """The shape of the local, uncompressed array."""
        return self._shape

    @property
    def global_shape(self):
        return self._global_shape

    @property
    def leading_shape(self):
        """The local shape of leading uncompressed dimensions."""
        return self._leading_shape

    @property
    def global_leading_shape(self):
        return self._global_leading_shape

    @property
    def stream_size(self):
        """The uncompressed length of each stream."""
        return self._stream_size

    # ---- properties of the compressed data ----
    @property
    def nbytes(self):
        """Bytes used by the compressed data."""
        return self._local_nbytes

    @property
    def global_nbytes(self):
        return self._global_nbytes

    @property
    def global_process_nbytes(self):
        return self._global_proc_nbytes

    @property
    def nstreams(self):
        return self._local_nstreams

    @property
    def global_nstreams(self):
        return self._global_nstreams

    @property
    def compressed(self):
        """The concatenated raw bytes of all streams."""
        return self._compressed

    @property
    def stream_starts(self):
        return self._stream_starts

    @property
    def stream_nbytes(self):
        return self._stream_nbytes

    @property
    def global_stream_starts(self):
        return self._global_stream_starts

    @property
    def global_stream_nbytes(self):
        return self._global_stream_nbytes

    @property
    def stream_offsets(self):
        """The value subtracted from each stream during conversion to int32."""
        return self._stream_offsets

    @property
    def stream_gains(self):
        """The gain factor for each stream during conversion to int32."""
        return self._stream_gains

    @property
    def mpi_comm(self):
        return self._mpi_comm

    @property
    def mpi_dist(self):
        return self._mpi_dist

    @property
    def dtype(self):
        return self._dtype

    @property
    def typestr(self):
        return self._typestr

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
                flat, _ = self._index().decode_slices(sel, np.full(sel.size, f0, np.int64), np.full(sel.size, n, np.int64), offsets=off, gains=gain)
                out = flat.reshape(sel.size, n)
        return (out if as_tensor else out.cpu().numpy()), indices

    def __getitem__(self, raw_key):
        """Decompress a selection on the fly; the result has numpy's shape for the same key."""
        shape, keep, first, last = self._plan_selection(raw_key)
        if 0 in shape:
            return np.zeros(shape, dtype=self._dtype)
        if self._resident is not None:
            arr, _ = self._decode_resident(keep, first, last)
            return arr.reshape(shape)
        arr, _ = array_decompress_slice(
            self._compressed, self._stream_size, self._stream_starts, self._stream_nbytes, stream_offsets=self._stream_offsets,
            stream_gains=self._stream_gains, keep=keep, first_stream_sample=first, last_stream_sample=last,
            is_int64=self._is_int64,
        )
        return arr.reshape(shape)

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
        first_samp = None
        last_samp = None
        if stream_slice is not None:
            if stream_slice.step is not None and stream_slice.step != 1:
                raise RuntimeError("Only stream slices with a step size of 1 are supported")
            first_samp, last_samp, _ = stream_slice.indices(self._stream_size)
        if self._resident is not None:
            f, l = (-1, -1) if first_samp is None else (first_samp, last_samp)
            if f >= 0 and l <= f:
                raise RuntimeError("first_sample is larger than last_sample")
            arr, indices = self._decode_resident(keep, f, l)
            if keep is None:
                arr = arr.reshape(self._shape[:-1] + (arr.shape[-1],)) if not self._flatten_single else arr.reshape(-1)
            if keep is not None and keep_indices:
                return (arr, indices)
            return arr
        arr, indices = array_decompress_slice(
            self._compressed,
            self._stream_size,
            self._stream_starts,
            self._stream_nbytes,
            stream_offsets=self._stream_offsets,
            stream_gains=self._stream_gains,
            keep=keep,
            first_stream_sample=first_samp,
            last_stream_sample=last_samp,
            is_int64=self._is_int64,
            use_threads=use_threads,
            no_flatten=(not self._flatten_single),
        )
        if keep is not None and keep_indices:
            return (arr, indices)
        return arr

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
            out, out_off = self._index().decode_slices(streams, first, count, offsets=res["offsets"], gains=res["gains"])
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
        shape = tuple(data.shape)
        out = FlacArray(
            None,
            shape=shape,
            global_shape=(1, shape[0]) if len(shape) == 1 else shape,
            compressed=comp.cpu().numpy(),
            dtype=np.dtype(str(data.dtype).replace("torch.", "")),
            stream_starts=st.cpu().numpy(),
            stream_nbytes=nb.cpu().numpy(),
            stream_offsets=None if offsets is None else offsets.cpu().numpy(),
            stream_gains=None if gains is None else gains.cpu().numpy(),
        )
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
        compressed, starts, nbytes, offsets, gains = array_compress(
            arr, level=level, quanta=quanta, precision=precision, use_threads=use_threads
        )
        return FlacArray(
            None,
            shape=arr.shape,
            global_shape=(1, arr.shape[0]) if arr.ndim == 1 else arr.shape,  # global_array_properties, mpi.py:109-117
            compressed=compressed,
            dtype=arr.dtype,
            stream_starts=starts,
            stream_nbytes=nbytes,
            stream_offsets=offsets,
            stream_gains=gains,
            mpi_comm=None,
            mpi_dist=None,
        )

    def write_hdf5(self, hgrp):
        """Write the compressed representation to an open HDF5 group (array.py:639-682), format
        version 1 (flacarray_amd/hdf5.py)."""
        from .hdf5 import write_compressed

        write_compressed(
            hgrp,
            self._leading_shape,
            self._global_leading_shape,
            self._stream_size,
            self._stream_starts,
            self._global_stream_starts,
            self._stream_nbytes,
            self._stream_offsets,
            self._stream_gains,
            self._compressed,
            2 if self._is_int64 else 1,
        )

    @classmethod
    def read_hdf5(cls, hgrp, keep=None, mpi_comm=None, mpi_dist=None, no_flatten=False):
        """Construct a FlacArray from an HDF5 group (array.py:684-764).  With `keep` the array
        holds only the selected streams, as a 2-D (n_kept, stream_size) array."""
        from .hdf5 import read_compressed
        from .utils import compressed_dtype

        (local_shape, global_shape, compressed, n_channels, stream_starts, stream_nbytes, stream_offsets, stream_gains,
         mpi_dist, keep_indices) = read_compressed(hgrp, keep=keep, mpi_comm=mpi_comm, mpi_dist=mpi_dist)
        dt = compressed_dtype(n_channels, stream_offsets, stream_gains)
        if (len(local_shape) == 2 and local_shape[0] == 1) and not no_flatten:
            shape = (local_shape[1],)
        else:
            shape = local_shape
        return FlacArray(
            None,
            shape=shape,
            global_shape=global_shape,
            compressed=compressed,
            dtype=dt,
            stream_starts=stream_starts,
            stream_nbytes=stream_nbytes,
            stream_offsets=stream_offsets,
            stream_gains=stream_gains,
            mpi_comm=None,
            mpi_dist=None,
        )

    def write_zarr(self, zgrp):
        """Write the compressed representation to an open Zarr group (array.py:766-804); same schema as HDF5."""
        self.write_hdf5(zgrp)

    @classmethod
    def read_zarr(cls, zgrp, keep=None, mpi_comm=None, mpi_dist=None, no_flatten=False):
        """Construct a FlacArray from a Zarr group (array.py:806-884)."""
        return cls.read_hdf5(zgrp, keep=keep, mpi_comm=mpi_comm, mpi_dist=mpi_dist, no_flatten=no_flatten)
