"""Binding layer: the names the reference's Cython module exports, on top of the HIP C ABI.

Mirrors src/flacarray/libflacarray/libflacarray.pyx of the reference: `encode_flac` (:529),
`decode_flac` (:713), `wrap_encode_i32[_threaded]` (:285/:346), `wrap_decode_i32` (:597),
`wrap_float32_to_int32` (:113), `wrap_int32_to_float32` (:215) -- same arguments, return
shapes/dtypes and error text.  numpy in / numpy out goes through the host-pointer C entry
points; the `*_device` functions at the bottom take torch tensors already resident in HBM.

The int64 / float64 twins (`wrap_encode_i64[_threaded]` :407/:468, `wrap_decode_i64` :656,
`wrap_float64_to_int64` :164, `wrap_int64_to_float64` :250) run the two-channel variants of the
same kernels.
"""
import ctypes
import os
import sys
import threading
import weakref

import numpy as np

from . import _lib

flac_i32_dtype = np.dtype(np.int32)
flac_i64_dtype = np.dtype(np.int64)
compressed_dtype = np.dtype(np.uint8)
offset_dtype = np.dtype(np.int64)



def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _adopt_malloc(addr, n):
    """numpy view of a malloc()'d buffer that free()s it when the last view dies."""
    if n == 0:
        _lib.libc_free(addr)
        return np.zeros(0, dtype=np.uint8)
    buf = (ctypes.c_uint8 * n).from_address(addr)
    arr = np.frombuffer(buf, dtype=np.uint8)
    weakref.finalize(buf, _lib.libc_free, addr)
    return arr


class _PinnedPool:
    """Pinned, device-visible host blocks behind the numpy arrays the small-read paths return.

    A read whose samples are wanted on the host is fastest when the decoder writes them where they are going: into
    pinned memory (fa_decode_indexed_host, include/flacarray_hip.h).  A fresh numpy array is neither pinned nor present
    (its pages appear on first touch, ~19 GB/s on the benchmark host: 75 of the 250 us of a 100-slice read), so results
    from 64 KB to 64 MB come from this pool instead: power-of-two blocks, handed back when the last view of the
    array dies and kept for the next read (at most FLACARRAY_HIP_PINNED_POOL_MB, default 256, sit idle; at most
    FLACARRAY_HIP_PINNED_MAX_MB, default 1024, are out at any time -- beyond that, and whenever pinned memory is not to be
    had, the caller gets an ordinary array).  FLACARRAY_HIP_PINNED_POOL_MB=0 switches the pool off."""

    _MIN, _MAX = 65536, 64 << 20

    def __init__(self):
        self._free = {}
        self._idle = 0
        self._out = 0
        self._lock = threading.Lock()
        self._idle_limit = int(os.environ.get("FLACARRAY_HIP_PINNED_POOL_MB", "256")) << 20
        self._out_limit = int(os.environ.get("FLACARRAY_HIP_PINNED_MAX_MB", "1024")) << 20

    def _take(self, nbytes):
        cap = max(self._MIN, 1 << max(nbytes - 1, 0).bit_length())
        if self._idle_limit <= 0 or cap > self._MAX:
            return None
        with self._lock:
            if self._out + cap > self._out_limit:
                return None
            self._out += cap
            blocks = self._free.get(cap)
            if blocks:
                self._idle -= cap
                return blocks.pop(), cap
        addr = _lib.lib().fa_pinned_alloc(cap)
        if not addr:
            with self._lock:
                self._out -= cap
            return None
        return addr, cap

    def _give(self, addr, cap):
        if sys is None or sys.is_finalizing():
            return
        with self._lock:
            self._out -= cap
            if self._idle + cap <= self._idle_limit:
                self._free.setdefault(cap, []).append(addr)
                self._idle += cap
                return
        _lib.lib().fa_pinned_free(addr)

    def empty(self, n, dtype):
        """1-D array of n elements on a pinned block, or None."""
        dtype = np.dtype(dtype)
        nbytes = int(n) * dtype.itemsize
        if nbytes <= 0:
            return None
        got = self._take(nbytes)
        if got is None:
            return None
        addr, cap = got
        buf = (ctypes.c_uint8 * nbytes).from_address(addr)
        fin = weakref.finalize(buf, self._give, addr, cap)
        fin.atexit = False  # (at interpreter exit the block goes with the process)
        return np.frombuffer(buf, dtype=dtype)

    def drain(self):
        """Free the idle blocks (tests; fa_release_scratch callers)."""
        with self._lock:
            blocks = [a for lst in self._free.values() for a in lst]
            self._free.clear()
            self._idle = 0
        for a in blocks:
            _lib.lib().fa_pinned_free(a)


_pinned_pool = _PinnedPool()


def _advise_huge_pages(arr):
    """madvise(MADV_HUGEPAGE) over the 2 MB-aligned inside of a large array this module has just allocated (never on
    a caller's memory).  What bounds decode_i32 into a fresh array is the appearance of its pages (profiles/
    r03_host_abi.md: 19 GB/s in 4 KB pages, 26 GB/s in transparent huge pages on the benchmark host)."""
    if arr.nbytes < (64 << 20):
        return
    try:
        huge = 2 << 20
        lo = (arr.ctypes.data + huge - 1) & ~(huge - 1)
        hi = (arr.ctypes.data + arr.nbytes) & ~(huge - 1)
        if hi > lo:
            _lib.libc_madvise(lo, hi - lo, 14)  # MADV_HUGEPAGE; failure (no THP) changes nothing
    except Exception:  # noqa: BLE001
        pass


def wrap_float32_to_int32(flatdata, n_stream, stream_size, quanta, _f64=False):
    """libflacarray.pyx:113-161.  `quanta` is used only if len(quanta) == n_stream."""
    _lib.require_device()
    ft, it = (np.float64, np.int64) if _f64 else (np.float32, np.int32)
    flatdata = np.ascontiguousarray(flatdata, dtype=ft)
    size = n_stream * stream_size
    output = np.empty(size, dtype=it)
    offsets = np.empty(n_stream, dtype=ft)
    gains = np.empty(n_stream, dtype=ft)
    q = None
    if len(quanta) == n_stream:
        q = np.ascontiguousarray(quanta, dtype=ft)
    errcode = (_lib.lib().float64_to_int64 if _f64 else _lib.lib().float32_to_int32)(
        _ptr(flatdata), n_stream, stream_size, _ptr(q) if q is not None else None, _ptr(output), _ptr(offsets), _ptr(gains)
    )
    if errcode & _lib.ERROR_NAN_INPUT:
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    return (output, offsets, gains)


def wrap_int32_to_float32(idata, n_stream, stream_size, offsets, gains, _f64=False):
    """libflacarray.pyx:215-247"""
    _lib.require_device()
    ft, it = (np.float64, np.int64) if _f64 else (np.float32, np.int32)
    idata = np.ascontiguousarray(idata, dtype=it)
    offsets = np.ascontiguousarray(offsets, dtype=ft)
    gains = np.ascontiguousarray(gains, dtype=ft)
    output = np.empty(n_stream * stream_size, dtype=ft)
    (_lib.lib().int64_to_float64 if _f64 else _lib.lib().int32_to_float32)(
        _ptr(idata), n_stream, stream_size, _ptr(offsets), _ptr(gains), _ptr(output)
    )
    return output


def wrap_float64_to_int64(flatdata, n_stream, stream_size, quanta):
    """libflacarray.pyx:164-212"""
    return wrap_float32_to_int32(flatdata, n_stream, stream_size, quanta, _f64=True)


def wrap_int64_to_float64(idata, n_stream, stream_size, offsets, gains):
    """libflacarray.pyx:250-282"""
    return wrap_int32_to_float32(idata, n_stream, stream_size, offsets, gains, _f64=True)


def _wrap_encode(fn, flatdata, n_stream, stream_size, level, dtype=np.int32):
    _lib.require_device()
    flatdata = np.ascontiguousarray(flatdata, dtype=dtype)
    flat_starts = np.empty(n_stream, dtype=np.int64)
    flat_nbytes = np.empty(n_stream, dtype=np.int64)
    n_bytes = ctypes.c_int64(0)
    raw = ctypes.c_void_p(None)
    errcode = fn(_ptr(flatdata), n_stream, stream_size, level, ctypes.byref(n_bytes), _ptr(flat_starts), ctypes.byref(raw))
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    flat_nbytes[:-1] = np.diff(flat_starts)
    flat_nbytes[-1] = n_bytes.value - flat_starts[-1]
    return (_adopt_malloc(raw.value, n_bytes.value), flat_starts, flat_nbytes)


def encode_flac_f32(data, quanta, level):
    """float32 array -> (compressed, starts, nbytes, offsets, gains) in one trip over PCIe (fa_encode_f32_host): the
    samples go up once and are quantised where the encoder loads them -- the same integers, offsets, gains and bytes as
    `float_to_int` (utils.py:246-342) followed by `encode_flac` (libflacarray.pyx:529-594), which moves the array up,
    the integers down, and the integers up again.  `quanta`: None (from each stream's range) or one value per stream."""
    _lib.require_device()
    if data.dtype != np.dtype(np.float32):
        raise ValueError("Only float32 data is supported")
    if level < 0 or level > 8:
        raise RuntimeError("FLAC only supports compression levels 0-8")
    data = np.ascontiguousarray(data)
    stream_size = data.shape[-1]
    lead = data.shape[:-1] if data.ndim > 1 else (1,)
    n_stream = int(np.prod(lead))
    q = None
    if quanta is not None:
        q = np.ascontiguousarray(quanta, dtype=np.float32).reshape(-1)
        if q.size != n_stream:
            raise ValueError("quanta must have one value per stream")
    flat_starts = np.empty(n_stream, dtype=np.int64)
    flat_nbytes = np.empty(n_stream, dtype=np.int64)
    offsets = np.empty(n_stream, dtype=np.float32)
    gains = np.empty(n_stream, dtype=np.float32)
    n_bytes = ctypes.c_int64(0)
    raw = ctypes.c_void_p(None)
    errcode = _lib.lib().fa_encode_f32_host(_ptr(data), n_stream, stream_size, level, _ptr(q) if q is not None else None,
                                            ctypes.byref(n_bytes), _ptr(flat_starts), ctypes.byref(raw), _ptr(offsets), _ptr(gains))
    if errcode & _lib.ERROR_NAN_INPUT:
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    flat_nbytes[:-1] = np.diff(flat_starts)
    flat_nbytes[-1] = n_bytes.value - flat_starts[-1]
    return (_adopt_malloc(raw.value, n_bytes.value), flat_starts.reshape(lead), flat_nbytes.reshape(lead), offsets.reshape(lead),
            gains.reshape(lead))


def encode_flac_f64(data, quanta, level):
    """float64 array -> (compressed, starts, nbytes, offsets, gains) in one trip over PCIe (fa_encode_f64_host): the
    float64 samples go up once, are quantised on the device (float64_to_int64, utils.c:245-327) and encoded from there as
    two-channel streams -- the same integers, offsets, gains and bytes as `float_to_int` followed by `encode_flac`
    (compress.py:50-84), which moves the array up, the int64 image down, and the int64 image up again."""
    _lib.require_device()
    if data.dtype != np.dtype(np.float64):
        raise ValueError("Only float64 data is supported")
    if level < 0 or level > 8:
        raise RuntimeError("FLAC only supports compression levels 0-8")
    data = np.ascontiguousarray(data)
    stream_size = data.shape[-1]
    lead = data.shape[:-1] if data.ndim > 1 else (1,)
    n_stream = int(np.prod(lead))
    q = None
    if quanta is not None:
        q = np.ascontiguousarray(quanta, dtype=np.float64).reshape(-1)
        if q.size != n_stream:
            raise ValueError("quanta must have one value per stream")
    flat_starts = np.empty(n_stream, dtype=np.int64)
    flat_nbytes = np.empty(n_stream, dtype=np.int64)
    offsets = np.empty(n_stream, dtype=np.float64)
    gains = np.empty(n_stream, dtype=np.float64)
    n_bytes = ctypes.c_int64(0)
    raw = ctypes.c_void_p(None)
    errcode = _lib.lib().fa_encode_f64_host(_ptr(data), n_stream, stream_size, level, _ptr(q) if q is not None else None,
                                            ctypes.byref(n_bytes), _ptr(flat_starts), ctypes.byref(raw), _ptr(offsets), _ptr(gains))
    if errcode & _lib.ERROR_NAN_INPUT:
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    flat_nbytes[:-1] = np.diff(flat_starts)
    flat_nbytes[-1] = n_bytes.value - flat_starts[-1]
    return (_adopt_malloc(raw.value, n_bytes.value), flat_starts.reshape(lead), flat_nbytes.reshape(lead), offsets.reshape(lead),
            gains.reshape(lead))


def wrap_encode_i32(flatdata, n_stream, stream_size, level):
    """libflacarray.pyx:285-343"""
    return _wrap_encode(_lib.lib().encode_i32, flatdata, n_stream, stream_size, level)


def wrap_encode_i32_threaded(flatdata, n_stream, stream_size, level):
    """libflacarray.pyx:346-404"""
    return _wrap_encode(_lib.lib().encode_i32_threaded, flatdata, n_stream, stream_size, level)


def _blocksize_classes(header_bytes):
    """Streams of one decode call normally share their block size (one array, one level).  A store assembled from
    several encodes may not -- libFLAC decodes every stream on its own terms (decompress.c:256-305), the GPU decoder
    takes one block size per launch.  header_bytes: uint8 [n_stream, 12], the first bytes of every stream
    ("fLaC", STREAMINFO header, min / max block size).  Returns a list of index arrays, one per block size, or None
    when the headers are not what they should be or all streams agree."""
    h = np.asarray(header_bytes, dtype=np.uint8)
    if h.ndim != 2 or h.shape[1] < 12 or h.shape[0] < 2:
        return None
    if not (np.all(h[:, 0:4] == np.frombuffer(b"fLaC", np.uint8)) and np.all((h[:, 4] & 0x7F) == 0)):
        return None
    bmin = (h[:, 8].astype(np.int64) << 8) | h[:, 9]
    bmax = (h[:, 10].astype(np.int64) << 8) | h[:, 11]
    key = bmin * 65536 + bmax
    kinds = np.unique(key)
    if kinds.size < 2:
        return None
    return [np.flatnonzero(key == k) for k in kinds]


def wrap_decode_i32(compressed, starts, nbytes, n_stream, stream_size, first_sample, last_sample, use_threads, _i64=False):
    """libflacarray.pyx:597-653"""
    _lib.require_device()
    n_decode = stream_size
    if first_sample >= 0 and last_sample >= 0:
        n_decode = last_sample - first_sample
    compressed = np.ascontiguousarray(compressed, dtype=np.uint8)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    nbytes = np.ascontiguousarray(nbytes, dtype=np.int64)
    output = np.empty(max(n_stream * n_decode, 0), dtype=flac_i64_dtype if _i64 else flac_i32_dtype)
    _advise_huge_pages(output)
    errcode = (_lib.lib().decode_i64 if _i64 else _lib.lib().decode_i32)(
        _ptr(compressed), _ptr(starts), _ptr(nbytes), n_stream, stream_size, first_sample, last_sample, _ptr(output),
        bool(use_threads),
    )
    if errcode != 0:
        # streams of different block sizes in one call: one launch per block size, rows scattered back
        groups = None
        if n_stream > 1 and np.all(starts >= 0) and np.all(nbytes >= 12) and np.all(starts + nbytes <= compressed.size):
            groups = _blocksize_classes(compressed[starts[:, None] + np.arange(12)[None, :]])
        if groups is None:
            raise RuntimeError(f"Decoding failed, return code = {errcode}")
        out2 = output.reshape(n_stream, max(n_decode, 0))
        for idx in groups:
            out2[idx] = wrap_decode_i32(compressed, starts[idx], nbytes[idx], int(idx.size), stream_size, first_sample, last_sample,
                                        use_threads, _i64=_i64).reshape(idx.size, -1)
    return output


def wrap_decode_i64(compressed, starts, nbytes, n_stream, stream_size, first_sample, last_sample, use_threads):
    """libflacarray.pyx:656-710"""
    return wrap_decode_i32(compressed, starts, nbytes, n_stream, stream_size, first_sample, last_sample, use_threads, _i64=True)


def wrap_encode_i64(flatdata, n_stream, stream_size, level):
    """libflacarray.pyx:407-465"""
    return _wrap_encode(_lib.lib().encode_i64, flatdata, n_stream, stream_size, level, dtype=np.int64)


def wrap_encode_i64_threaded(flatdata, n_stream, stream_size, level):
    """libflacarray.pyx:468-526"""
    return _wrap_encode(_lib.lib().encode_i64_threaded, flatdata, n_stream, stream_size, level, dtype=np.int64)





def encode_flac(data, level, use_threads=False):
    """Compress an integer array to FLAC streams (libflacarray.pyx:529-594).

    Returns (compressed bytestream, stream starting bytes, stream nbytes); starts and nbytes
    have the leading shape of `data` and are at least 1-D.
    """
    if data.dtype != flac_i32_dtype and data.dtype != flac_i64_dtype:
        raise RuntimeError("Only 32bit or 64bit integer data is supported")
    if not data.flags.c_contiguous:
        raise RuntimeError("Only C-contiguous arrays are supported")
    if level < 0 or level > 8:
        raise RuntimeError("FLAC only supports compression levels 0-8")
    is_i64 = data.dtype == flac_i64_dtype
    stream_size = data.shape[-1]
    if len(data.shape[:-1]) == 0:
        n_stream = 1
        starts_shape = (1,)
    else:
        n_stream = int(np.prod(data.shape[:-1]))
        starts_shape = data.shape[:-1]
    flatdata = data.reshape((-1,))
    if is_i64:
        enc = wrap_encode_i64_threaded if use_threads else wrap_encode_i64
    else:
        enc = wrap_encode_i32_threaded if use_threads else wrap_encode_i32
    compressed, flatstarts, flatnbytes = enc(flatdata, n_stream, stream_size, level)
    return (compressed, flatstarts.reshape(starts_shape), flatnbytes.reshape(starts_shape))


def decode_flac(compressed, starts, nbytes, stream_size, first_sample=-1, last_sample=-1, use_threads=False, is_int64=False):
    """Decompress FLAC streams (libflacarray.pyx:713-823); output shape starts.shape + (n_decode,)."""
    if compressed.dtype != compressed_dtype:
        raise RuntimeError("Compressed data should be of type uint8")
    if not compressed.flags.c_contiguous:
        raise RuntimeError("Only C-contiguous arrays are supported")
    if starts.dtype != offset_dtype:
        raise RuntimeError("starts data should be of type int64")
    if not starts.flags.c_contiguous:
        raise RuntimeError("Only C-contiguous arrays are supported")
    if nbytes.dtype != offset_dtype:
        raise RuntimeError("nbytes data should be of type int64")
    if not nbytes.flags.c_contiguous:
        raise RuntimeError("Only C-contiguous arrays are supported")
    if stream_size <= 0:
        raise RuntimeError("You must specify the non-zero output stream size")
    if len(compressed.shape) != 1:
        raise RuntimeError("Compressed byte array should be one dimensional")
    n_decode = stream_size
    if first_sample >= 0 and last_sample >= 0:
        if last_sample > stream_size:
            raise RuntimeError("last_sample is beyond end of stream")
        if first_sample > stream_size - 1:
            raise RuntimeError("first_sample is beyond last element of stream")
        if first_sample >= last_sample:
            raise RuntimeError("first_sample is larger than last_sample")
        n_decode = last_sample - first_sample
    output_shape = starts.shape + (n_decode,)
    n_stream = int(np.prod(starts.shape))
    flat_output = (wrap_decode_i64 if is_int64 else wrap_decode_i32)(
        compressed, starts.reshape((-1,)), nbytes.reshape((-1,)), n_stream, int(stream_size), int(first_sample),
        int(last_sample), use_threads,
    )
    return flat_output.reshape(output_shape)


def decode_flac_restore(compressed, starts, nbytes, stream_size, offsets, gains, first_sample=-1, last_sample=-1, is_int64=False):
    """`decode_flac` followed by `int_to_float` (decompress.py:107-136) in one trip over PCIe: compressed bytes up, the
    restore of utils.c:329-368 fused into the decoder's store, float32 / float64 down (fa_decode_f32_host /
    fa_decode_f64_host) -- the same values, bit for bit, as the two calls.  Shapes as decode_flac; offsets / gains have
    the shape of `starts`.  Returns None when the library refuses the call (e.g. streams of different block sizes): the
    caller then takes the two-call path, which knows how to split such a store."""
    _lib.require_device()
    n_decode = stream_size if (first_sample < 0 or last_sample < 0) else last_sample - first_sample
    ftype = np.float64 if is_int64 else np.float32
    st = np.ascontiguousarray(starts, dtype=np.int64)
    nb = np.ascontiguousarray(nbytes, dtype=np.int64).reshape(-1)
    off = np.ascontiguousarray(offsets, dtype=ftype).reshape(-1)
    gain = np.ascontiguousarray(gains, dtype=ftype).reshape(-1)
    n_stream = int(st.size)
    if off.size != n_stream or gain.size != n_stream or n_decode <= 0:
        return None
    compressed = np.ascontiguousarray(compressed, dtype=np.uint8)
    out = np.empty(n_stream * n_decode, dtype=ftype)
    _advise_huge_pages(out)
    fn = _lib.lib().fa_decode_f64_host if is_int64 else _lib.lib().fa_decode_f32_host
    errcode = fn(_ptr(compressed), _ptr(st.reshape(-1)), _ptr(nb), n_stream, int(stream_size), int(first_sample), int(last_sample),
                 _ptr(off), _ptr(gain), _ptr(out))
    if errcode != 0:
        return None
    return out.reshape(st.shape + (n_decode,))


def decode_flac_into(compressed, starts, nbytes, stream_size, out, first_sample=-1, last_sample=-1):
    """`decode_flac` into an array the caller owns: the C boundary takes a caller-allocated output (decode_i32 /
    decode_i64, flacarray.h:249-271; libflacarray.pyx:634 allocates a fresh one per call), and a caller that decodes
    store after store into the same array pays for the transfer, not for the population of fresh pages.  `out`:
    C-contiguous int32 / int64 of starts.size * n_decode elements."""
    _lib.require_device()
    n_decode = stream_size if (first_sample < 0 or last_sample < 0) else last_sample - first_sample
    starts = np.ascontiguousarray(starts, dtype=np.int64).reshape(-1)
    nbytes = np.ascontiguousarray(nbytes, dtype=np.int64).reshape(-1)
    compressed = np.ascontiguousarray(compressed, dtype=np.uint8)
    is64 = out.dtype == np.dtype(np.int64)
    if out.dtype not in (np.dtype(np.int32), np.dtype(np.int64)) or not out.flags.c_contiguous or out.size != starts.size * n_decode:
        raise RuntimeError("out must be a C-contiguous int32 / int64 array of starts.size x n_decode elements")
    errcode = (_lib.lib().decode_i64 if is64 else _lib.lib().decode_i32)(
        _ptr(compressed), _ptr(starts), _ptr(nbytes), int(starts.size), int(stream_size), int(first_sample), int(last_sample), _ptr(out), False)
    if errcode != 0:
        raise RuntimeError(f"Decoding failed, return code = {errcode}")
    return out


# ---------------------------------------------------------------------------------------------
# Device-resident variants (torch tensors on the GPU; torch supplies memory and streams only)
# ---------------------------------------------------------------------------------------------
def _torch():
    import torch

    return torch


def _stream_ptr():
    """The current HIP stream of the current device as a void* (torch's raw-stream query where it exists: the public
    route through torch.cuda.current_stream() costs ~9 us per call, a tenth of a small read)."""
    torch = _torch()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        try:
            return ctypes.c_void_p(raw(torch.cuda.current_device()))
        except Exception:  # noqa: BLE001
            pass
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(device):
    """`torch.cuda.device(device)` only when that is a switch: entering and leaving the context manager costs ~8 us, a
    tenth of a small call, and nearly every call runs on the current device already."""
    torch = _torch()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _NO_SWITCH if idx == torch.cuda.current_device() else torch.cuda.device(device)


class EncodeWorkspace:
    """Reusable HBM scratch for encode_flac_device (per-frame slots + scan arrays)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        torch = _torch()
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


def encode_flac_device(data, level=5, workspace=None, return_info=False, compact=False, capacity_bytes=None):
    """Encode a C-contiguous int32 (or int64: two-channel streams) CUDA tensor [..., stream_size] held in HBM.

    Returns (compressed uint8 tensor, starts int64 tensor, nbytes int64 tensor), all on the
    device, starts/nbytes with the leading shape of `data` (at least 1-D) -- the device-resident
    analogue of encode_flac (libflacarray.pyx:529-594).

    Every geometry is encoded in a single pass (K3F for full 4096-sample mono frames of levels 3-8, K3G for the rest:
    short streams, levels 0-2, int64; include/flacarray_hip.h): the frames end up at their final offsets inside a buffer
    sized for the worst case, and `compressed` is the view [0, total) of that buffer (it keeps the whole buffer alive;
    `compact=True` returns an exact-size copy instead).  `capacity_bytes` sizes that buffer instead of the worst case
    (1.016 x the input: every frame VERBATIM): if the blob does not fit, nothing outside the buffer is written and the
    call raises "Encoding failed, return code = 1" (ERROR_ALLOC) -- retry without it.  Under FLACARRAY_HIP_SLOTS (the
    diagnostic cross-check) everything runs the slot sequence (K3, K4, K5) and returns an exact-size tensor.
    """
    torch = _torch()
    if data.dtype != torch.int32 and data.dtype != torch.int64:
        raise RuntimeError("Only 32bit or 64bit integer data is supported")
    if not data.is_contiguous():
        raise RuntimeError("Only C-contiguous arrays are supported")
    if level < 0 or level > 8:
        raise RuntimeError("FLAC only supports compression levels 0-8")
    if not data.is_cuda:
        raise RuntimeError("encode_flac_device needs a tensor on the GPU")
    i64 = data.dtype == torch.int64
    stream_size = data.shape[-1]
    if data.dim() == 1:
        n_stream, starts_shape = 1, (1,)
    else:
        n_stream = int(np.prod(data.shape[:-1]))
        starts_shape = tuple(data.shape[:-1])
    L = _lib.lib()
    if workspace is None:
        workspace = EncodeWorkspace()
    index = torch.empty(2 * n_stream, dtype=torch.int64, device=data.device)  # (one allocation: starts | nbytes)
    starts, nbytes = index[:n_stream], index[n_stream:]
    info = None
    if return_info:
        bs = 1152 if level <= 2 else 4096
        nf = (stream_size + bs - 1) // bs
        info = torch.zeros((n_stream * nf * (2 if i64 else 1), 8), dtype=torch.int32, device=data.device)
    total = ctypes.c_int64(0)
    if L.fa_encode_single_pass_supported(n_stream, stream_size, level) and hasattr(L, "fa_encode_i64_device"):
        cap = (L.fa_encode_capacity_bytes_i64 if i64 else L.fa_encode_capacity_bytes)(n_stream, stream_size, level)
        if capacity_bytes is not None:
            cap = min(cap, int(capacity_bytes))
        ws = workspace.get((L.fa_encode_single_pass_workspace_bytes_i64 if i64 else L.fa_encode_single_pass_workspace_bytes)(n_stream, stream_size, level), data.device)
        with _on_device(data.device):
            buf = torch.empty(cap, dtype=torch.uint8, device=data.device)
            errcode = (L.fa_encode_i64_device if i64 else L.fa_encode_i32_device)(
                _dp(data), n_stream, stream_size, level, _dp(ws), ws.numel(), _dp(buf), cap, _dp(starts), _dp(nbytes),
                ctypes.byref(total), _dp(info), _stream_ptr(),
            )
        if errcode != 0:
            raise RuntimeError(f"Encoding failed, return code = {errcode}")
        compressed = buf[: total.value]
        if compact:
            compressed = compressed.clone()
        out = (compressed, starts.reshape(starts_shape), nbytes.reshape(starts_shape))
        return out + (info,) if return_info else out
    ws_bytes = (L.fa_encode_workspace_bytes_i64 if i64 else L.fa_encode_workspace_bytes)(n_stream, stream_size, level)
    if ws_bytes < 0:
        raise RuntimeError("Encoding failed, return code = 512")
    ws = workspace.get(ws_bytes, data.device)
    with _on_device(data.device):
        errcode = (L.fa_encode_i64_device_begin if i64 else L.fa_encode_i32_device_begin)(
            _dp(data), n_stream, stream_size, level, _dp(ws), ws.numel(), _dp(starts), _dp(nbytes), ctypes.byref(total),
            _dp(info), _stream_ptr(),
        )
        if errcode != 0:
            raise RuntimeError(f"Encoding failed, return code = {errcode}")
        compressed = torch.empty(total.value, dtype=torch.uint8, device=data.device)
        errcode = (L.fa_encode_i64_device_finish if i64 else L.fa_encode_i32_device_finish)(n_stream, stream_size, level, _dp(ws), _dp(starts), _dp(compressed), _stream_ptr())
        if errcode != 0:
            raise RuntimeError(f"Encoding failed, return code = {errcode}")
    out = (compressed, starts.reshape(starts_shape), nbytes.reshape(starts_shape))
    if return_info:
        return out + (info,)
    return out


def set_decode_verify(on):
    """Process-wide DEFAULT of the decoder's integrity pass, used by device decode calls whose `verify` argument is
    None: when on, the call re-computes the CRC-16 of each frame it read and raises ("Decoding failed, return code =
    16384", ERROR_DECODE_PROCESS) on a mismatch -- the condition libFLAC reports through the error callback the
    reference prints (decompress.c:104-121).  Returns the previous setting.  Initially off for device-resident stores
    (the pass re-reads the compressed bytes); the host-pointer path behind `decode_flac` always checks."""
    return bool(_lib.lib().fa_set_decode_verify(1 if on else 0))


def _verify_arg(verify):
    return -1 if verify is None else (1 if verify else 0)


def encode_flac_device_f32(data, quanta=None, level=5, workspace=None, compact=False):
    """Quantise and encode a C-contiguous float32 CUDA tensor [..., stream_size] held in HBM: the device-resident
    analogue of array_compress on float32 input (compress.py:50-84 -> float_to_int + encode_flac).

    `quanta`: None (per-stream quanta from the data range), or a tensor with one value per stream.  Returns
    (compressed, starts, nbytes, offsets, gains), offsets / gains float32 with the leading shape of `data`.  Where
    the single-pass kernel applies (levels 3-8, stream length a multiple of 4096) the quantisation happens in the
    encoder's staging load after a range pre-pass -- the int32 array never exists in HBM; otherwise the two steps
    run one after the other.  Same bytes, offsets and gains either way."""
    torch = _torch()
    if data.dtype != torch.float32 or not data.is_contiguous():
        raise ValueError("Only float32 and float64 data are supported")
    if not data.is_cuda:
        raise RuntimeError("encode_flac_device_f32 needs a tensor on the GPU")
    if level < 0 or level > 8:
        raise RuntimeError("FLAC only supports compression levels 0-8")
    stream_size = data.shape[-1]
    lead = tuple(data.shape[:-1]) if data.dim() > 1 else (1,)
    n_stream = int(np.prod(lead))
    q = None
    if quanta is not None:
        q = quanta.to(device=data.device, dtype=torch.float32).reshape(-1).contiguous()
        if q.numel() != n_stream:
            raise RuntimeError("quanta must have one entry per stream")
    L = _lib.lib()
    if not (data.data_ptr() % 16 == 0 and stream_size % 4096 == 0 and L.fa_encode_single_pass_supported(n_stream, stream_size, level)):
        ints, offsets, gains = float32_to_int32_device(data, q)
        comp, st, nb = encode_flac_device(ints, level=level, workspace=workspace, compact=compact)
        return comp, st, nb, offsets, gains
    if workspace is None:
        workspace = EncodeWorkspace()
    starts = torch.empty(n_stream, dtype=torch.int64, device=data.device)
    nbytes = torch.empty(n_stream, dtype=torch.int64, device=data.device)
    offsets = torch.empty(n_stream, dtype=torch.float32, device=data.device)
    gains = torch.empty(n_stream, dtype=torch.float32, device=data.device)
    cap = L.fa_encode_capacity_bytes(n_stream, stream_size, level)
    ws = workspace.get(L.fa_encode_single_pass_workspace_bytes(n_stream, stream_size, level), data.device)
    total = ctypes.c_int64(0)
    with _on_device(data.device):
        buf = torch.empty(cap, dtype=torch.uint8, device=data.device)
        errcode = L.fa_encode_f32_device(
            _dp(data), n_stream, stream_size, level, _dp(q), _dp(ws), ws.numel(), _dp(buf), cap, _dp(starts), _dp(nbytes),
            _dp(offsets), _dp(gains), ctypes.byref(total), None, _stream_ptr(),
        )
    if errcode & _lib.ERROR_NAN_INPUT:
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    compressed = buf[: total.value]
    if compact:
        compressed = compressed.clone()
    return compressed, starts.reshape(lead), nbytes.reshape(lead), offsets.reshape(lead), gains.reshape(lead)


def _device_regroup(errcode, out, compressed, starts, nbytes, stream_size, first_sample, last_sample, offsets, gains, is_int64):
    """A failed decode call whose streams differ in block size: one launch per block size (see _blocksize_classes),
    rows scattered back into `out`.  Anything else raises the reference's error."""
    torch = _torch()
    st, nb = starts.reshape(-1), nbytes.reshape(-1)
    n_stream = st.numel()
    groups = None
    if n_stream > 1 and bool(((st >= 0) & (nb >= 12) & (st + nb <= compressed.numel())).all()):
        groups = _blocksize_classes(compressed[st[:, None] + torch.arange(12, device=st.device)[None, :]].cpu().numpy())
    if groups is None:
        raise RuntimeError(f"Decoding failed, return code = {errcode}")
    out2 = out.reshape(n_stream, -1)
    for g in groups:
        gi = torch.from_numpy(g).to(st.device)
        out2[gi] = decode_flac_device(
            compressed, st[gi].contiguous(), nb[gi].contiguous(), stream_size, first_sample, last_sample,
            offsets=None if offsets is None else offsets.reshape(-1)[gi.to(offsets.device)],
            gains=None if gains is None else gains.reshape(-1)[gi.to(gains.device)], is_int64=is_int64,
        )
    return out


def decode_flac_device(compressed, starts, nbytes, stream_size, first_sample=-1, last_sample=-1, offsets=None, gains=None,
                       is_int64=False, verify=None):
    """Decode device-resident streams into an int32 tensor (or float32 when offsets/gains are
    given: the int->float restore of utils.c:350-368 is fused into the store).  is_int64:
    two-channel streams -> int64 (or float64 with float64 offsets/gains, utils.c:329-348).
    verify: True / False = check every frame's CRC-16 or not; None = the default of set_decode_verify."""
    torch = _torch()
    vfy = _verify_arg(verify)
    if compressed.dtype != torch.uint8:
        raise RuntimeError("Compressed data should be of type uint8")
    if starts.dtype != torch.int64:
        raise RuntimeError("starts data should be of type int64")
    if nbytes.dtype != torch.int64:
        raise RuntimeError("nbytes data should be of type int64")
    if not (compressed.is_contiguous() and starts.is_contiguous() and nbytes.is_contiguous()):
        raise RuntimeError("Only C-contiguous arrays are supported")
    if stream_size <= 0:
        raise RuntimeError("You must specify the non-zero output stream size")
    n_decode = stream_size
    if first_sample >= 0 and last_sample >= 0:
        if last_sample > stream_size:
            raise RuntimeError("last_sample is beyond end of stream")
        if first_sample > stream_size - 1:
            raise RuntimeError("first_sample is beyond last element of stream")
        if first_sample >= last_sample:
            raise RuntimeError("first_sample is larger than last_sample")
        n_decode = last_sample - first_sample
    if (offsets is None) != (gains is None):
        raise RuntimeError("When specifying offsets, you must also provide the gains")
    n_stream = int(np.prod(starts.shape))
    shape = tuple(starts.shape) + (n_decode,)
    dev = compressed.device
    L = _lib.lib()
    if is_int64:
        with _on_device(dev):
            if offsets is None:
                out = torch.empty(shape, dtype=torch.int64, device=dev)
                errcode = L.fa_decode_i64_device(
                    _dp(compressed), compressed.numel(), _dp(starts), _dp(nbytes), n_stream, stream_size, first_sample,
                    last_sample, _dp(out), None, None, None, _stream_ptr(), vfy,
                )
            else:
                out = torch.empty(shape, dtype=torch.float64, device=dev)
                offsets = offsets.to(device=dev, dtype=torch.float64).contiguous()
                gains = gains.to(device=dev, dtype=torch.float64).contiguous()
                errcode = L.fa_decode_i64_device(
                    _dp(compressed), compressed.numel(), _dp(starts), _dp(nbytes), n_stream, stream_size, first_sample,
                    last_sample, None, _dp(out), _dp(offsets), _dp(gains), _stream_ptr(), vfy,
                )
        if errcode != 0:
            return _device_regroup(errcode, out, compressed, starts, nbytes, stream_size, first_sample, last_sample, offsets, gains, True)
        return out
    with _on_device(dev):
        if offsets is None:
            out = torch.empty(shape, dtype=torch.int32, device=dev)
            errcode = L.fa_decode_i32_device(
                _dp(compressed), compressed.numel(), _dp(starts), _dp(nbytes), n_stream, stream_size, first_sample, last_sample,
                _dp(out), None, None, None, _stream_ptr(), vfy,
            )
        else:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
            offsets = offsets.to(device=dev, dtype=torch.float32).contiguous()
            gains = gains.to(device=dev, dtype=torch.float32).contiguous()
            errcode = L.fa_decode_i32_device(
                _dp(compressed), compressed.numel(), _dp(starts), _dp(nbytes), n_stream, stream_size, first_sample, last_sample,
                None, _dp(out), _dp(offsets), _dp(gains), _stream_ptr(), vfy,
            )
    if errcode != 0:
        return _device_regroup(errcode, out, compressed, starts, nbytes, stream_size, first_sample, last_sample, offsets, gains, False)
    return out


def decode_slices_device(compressed, starts, nbytes, stream_size, slice_stream, slice_first, slice_count, offsets=None, gains=None,
                         is_int64=False, verify=None):
    """Batched random access: slice i = samples [first[i], first[i]+count[i]) of (flat) stream
    slice_stream[i].  Returns (flat output tensor, int64 numpy array of output offsets).  The
    reference needs one decode call per slice (decompress.py:42-48)."""
    torch = _torch()
    slice_stream = np.ascontiguousarray(slice_stream, dtype=np.int64)
    slice_first = np.ascontiguousarray(slice_first, dtype=np.int64)
    slice_count = np.ascontiguousarray(slice_count, dtype=np.int64)
    n = slice_stream.shape[0]
    out_off = np.zeros(n, dtype=np.int64)
    if n > 1:
        np.cumsum(slice_count[:-1], out=out_off[1:])
    total = int(slice_count.sum())
    dev = compressed.device
    n_stream = int(np.prod(starts.shape))
    L = _lib.lib()
    f32 = offsets is not None
    ft, it = (torch.float64, torch.int64) if is_int64 else (torch.float32, torch.int32)
    out = torch.empty(total, dtype=ft if f32 else it, device=dev)
    if f32:
        # per-stream offsets / gains; the kernel looks them up by the slice's stream
        soff = offsets.reshape(-1).to(device=dev, dtype=ft).contiguous()
        sgain = gains.reshape(-1).to(device=dev, dtype=ft).contiguous()
    with _on_device(dev):
        errcode = (L.fa_decode_slices_i64_device if is_int64 else L.fa_decode_slices_i32_device)(
            _dp(compressed), compressed.numel(), _dp(starts), _dp(nbytes), n_stream, stream_size, n,
            ctypes.c_void_p(slice_stream.ctypes.data), ctypes.c_void_p(slice_first.ctypes.data),
            ctypes.c_void_p(slice_count.ctypes.data), ctypes.c_void_p(out_off.ctypes.data),
            None if f32 else _dp(out), _dp(out) if f32 else None, _dp(soff) if f32 else None, _dp(sgain) if f32 else None,
            _stream_ptr(), _verify_arg(verify),
        )
    if errcode != 0:
        raise RuntimeError(f"Decoding failed, return code = {errcode}")
    return out, out_off


class DeviceDecodeIndex:
    """Decode index of one HBM-resident store (fa_decode_index_create): the stream headers are parsed and the byte
    offset of every frame is tabulated ONCE; `decode` / `decode_slices` then cost one kernel launch each.  Holds
    references to the store's tensors (the C side refers to their memory)."""

    def __init__(self, compressed, starts, nbytes, stream_size, is_int64=False):
        torch = _torch()
        if compressed.dtype != torch.uint8 or starts.dtype != torch.int64 or nbytes.dtype != torch.int64:
            raise RuntimeError("Compressed data should be of type uint8")
        self.compressed = compressed.contiguous()
        self.starts = starts.reshape(-1).contiguous()
        self.nbytes = nbytes.reshape(-1).contiguous()
        self.stream_size = int(stream_size)
        self.n_stream = int(self.starts.numel())
        self.is_int64 = bool(is_int64)
        self.device = compressed.device
        self._L = _lib.lib()
        h = ctypes.c_void_p(None)
        with _on_device(self.device):
            errcode = self._L.fa_decode_index_create(
                _dp(self.compressed), self.compressed.numel(), _dp(self.starts), _dp(self.nbytes), self.n_stream, self.stream_size,
                2 if is_int64 else 1, ctypes.byref(h), _stream_ptr(),
            )
        if errcode != 0:
            raise RuntimeError(f"Decoding failed, return code = {errcode}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.fa_decode_index_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        if sys is None or sys.is_finalizing():  # the HIP runtime may be gone already: the process ends anyway
            return
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _types(self, to_float):
        torch = _torch()
        ft, it = (torch.float64, torch.int64) if self.is_int64 else (torch.float32, torch.int32)
        return ft if to_float else it, ft

    def decode(self, first_sample=-1, last_sample=-1, offsets=None, gains=None, verify=None):
        """[first_sample, last_sample) (or everything) of ALL streams -> tensor [n_stream, n_decode]."""
        torch = _torch()
        n_decode = self.stream_size
        if first_sample >= 0 and last_sample >= 0:
            if last_sample > self.stream_size:
                raise RuntimeError("last_sample is beyond end of stream")
            if first_sample > self.stream_size - 1:
                raise RuntimeError("first_sample is beyond last element of stream")
            if first_sample >= last_sample:
                raise RuntimeError("first_sample is larger than last_sample")
            n_decode = last_sample - first_sample
        dt, ft = self._types(offsets is not None)
        out = torch.empty((self.n_stream, n_decode), dtype=dt, device=self.device)
        if offsets is not None:
            offsets = offsets.reshape(-1).to(device=self.device, dtype=ft).contiguous()
            gains = gains.reshape(-1).to(device=self.device, dtype=ft).contiguous()
        with _on_device(self.device):
            errcode = self._L.fa_decode_indexed(
                self._h, first_sample, last_sample, -1, None, None, None, None, None if offsets is not None else _dp(out),
                _dp(out) if offsets is not None else None, _dp(offsets), _dp(gains), _stream_ptr(), _verify_arg(verify),
            )
        if errcode != 0:
            raise RuntimeError(f"Decoding failed, return code = {errcode}")
        return out

    def decode_slices(self, slice_stream, slice_first, slice_count, offsets=None, gains=None, verify=None, to_host=False):
        """Batched random access (see decode_slices_device): returns (flat tensor, int64 numpy array of offsets);
        `to_host`: the flat result as a numpy array instead, copied inside the decode call (one synchronisation)."""
        torch = _torch()
        slice_stream = np.ascontiguousarray(slice_stream, dtype=np.int64)
        slice_first = np.ascontiguousarray(slice_first, dtype=np.int64)
        slice_count = np.ascontiguousarray(slice_count, dtype=np.int64)
        n = slice_stream.shape[0]
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            np.cumsum(slice_count[:-1], out=out_off[1:])
        dt, ft = self._types(offsets is not None)
        out = torch.empty(int(slice_count.sum()), dtype=dt, device=self.device)
        if n == 0:
            return out, out_off
        if offsets is not None:
            offsets = offsets.reshape(-1).to(device=self.device, dtype=ft).contiguous()
            gains = gains.reshape(-1).to(device=self.device, dtype=ft).contiguous()
        if to_host:
            ndt = np.dtype(str(dt).replace("torch.", ""))
            # (64 KB and more: a pinned block the decoder writes straight into; below that the library's own landing
            # buffer and a memcpy are cheaper than the pool's bookkeeping: 122 against 129 us per single read)
            host = _pinned_pool.empty(out.numel(), ndt) if out.numel() * ndt.itemsize >= 65536 else None
            if host is None:
                host = np.empty(out.numel(), dtype=ndt)
            with _on_device(self.device):
                errcode = self._L.fa_decode_indexed_host(
                    self._h, n, ctypes.c_void_p(slice_stream.ctypes.data), ctypes.c_void_p(slice_first.ctypes.data),
                    ctypes.c_void_p(slice_count.ctypes.data), ctypes.c_void_p(out_off.ctypes.data),
                    None if offsets is not None else _dp(out), _dp(out) if offsets is not None else None, _dp(offsets), _dp(gains),
                    ctypes.c_void_p(host.ctypes.data), host.nbytes, _stream_ptr(), _verify_arg(verify),
                )
            if errcode != 0:
                raise RuntimeError(f"Decoding failed, return code = {errcode}")
            return host, out_off
        with _on_device(self.device):
            errcode = self._L.fa_decode_indexed(
                self._h, -1, -1, n, ctypes.c_void_p(slice_stream.ctypes.data), ctypes.c_void_p(slice_first.ctypes.data),
                ctypes.c_void_p(slice_count.ctypes.data), ctypes.c_void_p(out_off.ctypes.data),
                None if offsets is not None else _dp(out), _dp(out) if offsets is not None else None, _dp(offsets), _dp(gains),
                _stream_ptr(), _verify_arg(verify),
            )
        if errcode != 0:
            raise RuntimeError(f"Decoding failed, return code = {errcode}")
        return out, out_off


def float32_to_int32_device(data, quanta=None):
    """Device float32 -> int32 quantisation (utils.c:160-243); returns (int32 tensor, offsets, gains)."""
    torch = _torch()
    if data.dtype != torch.float32 or not data.is_contiguous():
        raise ValueError("Only float32 and float64 data are supported")
    stream_size = data.shape[-1]
    lead = tuple(data.shape[:-1]) if data.dim() > 1 else (1,)
    n_stream = int(np.prod(lead))
    out = torch.empty(data.shape, dtype=torch.int32, device=data.device)
    offsets = torch.empty(n_stream, dtype=torch.float32, device=data.device)
    gains = torch.empty(n_stream, dtype=torch.float32, device=data.device)
    q = None
    if quanta is not None:
        q = quanta.to(device=data.device, dtype=torch.float32).reshape(-1).contiguous()
        if q.numel() != n_stream:
            raise RuntimeError("quanta must have one entry per stream")
    with _on_device(data.device):
        errcode = _lib.lib().fa_float32_to_int32_device(
            _dp(data), n_stream, stream_size, _dp(q), _dp(out), _dp(offsets), _dp(gains), _stream_ptr()
        )
    if errcode & _lib.ERROR_NAN_INPUT:
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if errcode != 0:
        raise RuntimeError(f"Encoding failed, return code = {errcode}")
    return out, offsets.reshape(lead), gains.reshape(lead)
