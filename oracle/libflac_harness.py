"""TEST INFRASTRUCTURE ONLY -- a ctypes harness around a SYSTEM libFLAC (>= 1.4), if the machine has one.

SURVEY 8(c) asks for it: libFLAC is the third-party module that holds the reference's arithmetic
(src/flacarray/libflacarray/compress.c:337-390, decompress.c:256-305) and it is absent from the build image and
from the GPU image this round ("parity unpinned").  Where a libFLAC exists, this file drives it with the
reference's own settings -- compression level, blocksize 0 (auto), one channel, 32 bits per sample, memory
callbacks, no seek / tell / metadata callbacks -- so that

  * streams written by this repository can be decoded by the real library (ours -> libFLAC),
  * streams written by the real library can be decoded here (libFLAC -> oracle / HIP),
  * the real library can serve as the CPU baseline of bench.py.

Nothing here is copied from the reference: it is the public libFLAC C API (FLAC/stream_encoder.h,
FLAC/stream_decoder.h) bound with ctypes.  Only tests/ and bench.py's cpu_baseline leg may import it.
"""
import ctypes
import ctypes.util

import numpy as np

_lib = None
_tried = False

_ENC_WRITE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t, ctypes.c_uint32,
                              ctypes.c_uint32, ctypes.c_void_p)
_DEC_READ = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)
_DEC_WRITE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                              ctypes.c_void_p)
_DEC_ERROR = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)


def lib():
    """The loaded library, or None."""
    global _lib, _tried
    if not _tried:
        _tried = True
        name = ctypes.util.find_library("FLAC")
        if name:
            try:
                L = ctypes.CDLL(name)
                L.FLAC__stream_encoder_new.restype = ctypes.c_void_p
                L.FLAC__stream_decoder_new.restype = ctypes.c_void_p
                for f in ("FLAC__stream_encoder_set_compression_level", "FLAC__stream_encoder_set_blocksize", "FLAC__stream_encoder_set_channels",
                          "FLAC__stream_encoder_set_bits_per_sample"):
                    getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_uint32]
                    getattr(L, f).restype = ctypes.c_int
                L.FLAC__stream_encoder_init_stream.argtypes = [ctypes.c_void_p, _ENC_WRITE, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
                L.FLAC__stream_encoder_init_stream.restype = ctypes.c_int
                L.FLAC__stream_encoder_process_interleaved.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
                L.FLAC__stream_encoder_process_interleaved.restype = ctypes.c_int
                for f in ("FLAC__stream_encoder_finish", "FLAC__stream_decoder_finish", "FLAC__stream_decoder_process_until_end_of_stream"):
                    getattr(L, f).argtypes = [ctypes.c_void_p]
                    getattr(L, f).restype = ctypes.c_int
                for f in ("FLAC__stream_encoder_delete", "FLAC__stream_decoder_delete"):
                    getattr(L, f).argtypes = [ctypes.c_void_p]
                    getattr(L, f).restype = None
                L.FLAC__stream_decoder_init_stream.argtypes = [ctypes.c_void_p, _DEC_READ, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                               _DEC_WRITE, ctypes.c_void_p, _DEC_ERROR, ctypes.c_void_p]
                L.FLAC__stream_decoder_init_stream.restype = ctypes.c_int
                _lib = L
            except (OSError, AttributeError):
                _lib = None
    return _lib


def available():
    return lib() is not None


def version():
    L = lib()
    if L is None:
        return None
    try:
        return ctypes.c_char_p.in_dll(L, "FLAC__VERSION_STRING").value.decode()
    except ValueError:
        return "unknown"


def encode_i32(x, level=5):
    """One native FLAC stream per row with the reference's settings (compress.c:337-378).  Returns (bytes, starts,
    nbytes) laid out like the reference's encode_i32 (concatenated streams, exclusive prefix sums)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.int32)
    if x.ndim == 1:
        x = x[None, :]
    parts = []
    for row in x:
        chunks = []

        def on_write(_enc, buf, nbytes, _samples, _frame, _client):
            chunks.append(ctypes.string_at(buf, nbytes))
            return 0  # FLAC__STREAM_ENCODER_WRITE_STATUS_OK

        cb = _ENC_WRITE(on_write)
        enc = L.FLAC__stream_encoder_new()
        try:
            ok = L.FLAC__stream_encoder_set_compression_level(enc, level)
            ok = ok and L.FLAC__stream_encoder_set_blocksize(enc, 0)
            ok = ok and L.FLAC__stream_encoder_set_channels(enc, 1)
            ok = ok and L.FLAC__stream_encoder_set_bits_per_sample(enc, 32)
            if not ok or L.FLAC__stream_encoder_init_stream(enc, cb, None, None, None, None) != 0:
                raise RuntimeError("libFLAC encoder initialisation failed (32 bits per sample needs libFLAC >= 1.4)")
            row = np.ascontiguousarray(row)
            if not L.FLAC__stream_encoder_process_interleaved(enc, row.ctypes.data, row.shape[0]):
                raise RuntimeError("libFLAC process_interleaved failed")
            if not L.FLAC__stream_encoder_finish(enc):
                raise RuntimeError("libFLAC encoder finish failed")
        finally:
            L.FLAC__stream_encoder_delete(enc)
        parts.append(b"".join(chunks))
    nbytes = np.array([len(p) for p in parts], dtype=np.int64)
    starts = np.zeros_like(nbytes)
    np.cumsum(nbytes[:-1], out=starts[1:])
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), starts, nbytes


def decode_i32(blob, starts, nbytes, stream_size):
    """Every stream decoded in full with the real library (decompress.c:256-276).  Raises on a decoder error."""
    L = lib()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    out = np.empty((len(starts), stream_size), dtype=np.int32)
    for i, (s0, nb) in enumerate(zip(starts, nbytes)):
        data = blob[int(s0) : int(s0) + int(nb)].tobytes()
        state = {"pos": 0, "n": 0, "err": []}

        def on_read(_dec, buf, pbytes, _client):
            want = pbytes[0]
            take = min(want, len(data) - state["pos"])
            if take <= 0:
                pbytes[0] = 0
                return 1  # END_OF_STREAM
            ctypes.memmove(buf, data[state["pos"] : state["pos"] + take], take)
            state["pos"] += take
            pbytes[0] = take
            return 0

        def on_write(_dec, frame, buffers, _client):
            bs = frame[0]  # FLAC__FrameHeader.blocksize is the first member of FLAC__Frame
            n = min(bs, stream_size - state["n"])
            if n > 0:
                out[i, state["n"] : state["n"] + n] = np.ctypeslib.as_array(buffers[0], shape=(bs,))[:n]
            state["n"] += n
            return 0

        def on_error(_dec, status, _client):
            state["err"].append(int(status))

        cbs = (_DEC_READ(on_read), _DEC_WRITE(on_write), _DEC_ERROR(on_error))
        dec = L.FLAC__stream_decoder_new()
        try:
            if L.FLAC__stream_decoder_init_stream(dec, cbs[0], None, None, None, None, cbs[1], None, cbs[2], None) != 0:
                raise RuntimeError("libFLAC decoder initialisation failed")
            ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
            L.FLAC__stream_decoder_finish(dec)
        finally:
            L.FLAC__stream_decoder_delete(dec)
        if not ok or state["err"] or state["n"] != stream_size:
            raise RuntimeError(f"libFLAC could not decode stream {i}: ok={ok} errors={state['err']} samples={state['n']}/{stream_size}")
    return out
