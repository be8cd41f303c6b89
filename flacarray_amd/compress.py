"""array_compress: dtype dispatch in front of the encoder (reference: src/flacarray/compress.py:12-84).

Same call signature, return tuple and error behaviour as the reference; the work happens in
`encode_flac` (K3F / K3-K5) on the GPU; float data is quantised on the device in the same trip (K1: fused into the
encoder's load for float32, in front of the two-channel encoder for float64) -- `float_to_int`'s integers never cross PCIe.
"""
import numpy as np

from .libflacarray import encode_flac, encode_flac_f32, encode_flac_f64
from .utils import _quanta_for, _streams_of, function_timer

_INT_KINDS = (np.dtype(np.int32), np.dtype(np.int64))
_FLOAT_KINDS = (np.dtype(np.float32), np.dtype(np.float64))


def _per_stream_quanta(quanta, leading_shape, dtype):
    """Broadcast a scalar quanta over the streams, or check the shape of an array of them.

    (The reference trips over array-valued quanta here, compress.py:61-63 reads `.shape` of an int;
    per-stream quanta otherwise only work through float_to_int.  They are accepted.)"""
    if not hasattr(quanta, "__len__"):
        return np.full(leading_shape, quanta, dtype=dtype)
    per_stream = np.asarray(quanta)
    if per_stream.shape != leading_shape:
        msg = "If not a scalar, quanta must have the same shape as the "
        msg += "leading dimensions of the array"
        raise ValueError(msg)
    return per_stream.astype(dtype)


@function_timer
def array_compress(arr, level=5, quanta=None, precision=None, use_threads=False):
    """Compress a numpy array with optional floating point conversion.

    Integer input (int32, int64) is compressed as is and the last two elements of the result are None.
    Float input (float32, float64) needs exactly one of `quanta` (scalar or one value per stream)
    and `precision`; it is converted to integers with a per-stream offset and gain first.

    Returns (compressed bytes, stream starts, stream nbytes, stream offsets, stream gains); the
    auxiliary arrays have the leading shape of `arr` (one element for a single stream).
    """
    if arr.size == 0:
        raise ValueError("Cannot compress a zero-sized array!")
    kind = arr.dtype

    if kind in _INT_KINDS:
        compressed, starts, nbytes = encode_flac(np.ascontiguousarray(arr), level, use_threads=use_threads)
        return (compressed, starts, nbytes, None, None)

    if kind not in _FLOAT_KINDS:
        raise ValueError(f"Unsupported data type '{arr.dtype}'")

    if quanta is None and precision is None:
        msg = f"Compressing floating point data ('{arr.dtype}') "
        msg += "requires specifying either quanta or precision."
        raise RuntimeError(msg)
    if quanta is not None and precision is not None:
        raise RuntimeError("Cannot set both quanta and precision")

    stream_quanta = None if quanta is None else _per_stream_quanta(quanta, arr.shape[:-1], kind)
    if kind == np.dtype(np.float32):
        # one trip over PCIe: the float32 samples go up, are quantised where the encoder loads them (same integers,
        # offsets and gains as float_to_int: utils.c:160-243 on the device), and only the compressed bytes come back;
        # a NaN is found on the device and raises float_to_int's error
        lead, _, _ = _streams_of(arr)
        q = _quanta_for(arr, lead, stream_quanta, precision)
        return encode_flac_f32(arr, q if q.size else None, level)
    # float64 the same way: one trip up, quantised and encoded (two channels) on the device, bytes down
    lead, _, _ = _streams_of(arr)
    q = _quanta_for(arr, lead, stream_quanta, precision)
    return encode_flac_f64(arr, q if q.size else None, level)
