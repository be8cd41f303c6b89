"""array_compress (reference: src/flacarray/compress.py:12-84)."""
import numpy as np

from .libflacarray import encode_flac
from .utils import float_to_int, function_timer


@function_timer
def array_compress(arr, level=5, quanta=None, precision=None, use_threads=False):
    """Compress a numpy array with optional floating point conversion.

    int32 input: offsets and gains are None.  float32 input: exactly one of `quanta`
    (scalar or one value per stream) or `precision` is required.  Returns
    (compressed bytes, stream starts, stream nbytes, stream offsets, stream gains) with the
    auxiliary arrays shaped like the leading dimensions of `arr` (1 element for one stream).

    Unlike compress.py:61-63 an array-valued `quanta` is accepted (the reference raises
    AttributeError there; per-stream quanta otherwise only work through float_to_int).
    """
    if arr.size == 0:
        raise ValueError("Cannot compress a zero-sized array!")
    leading_shape = arr.shape[:-1]

    if arr.dtype == np.dtype(np.float32) or arr.dtype == np.dtype(np.float64):
        if quanta is None and precision is None:
            msg = f"Compressing floating point data ('{arr.dtype}') "
            msg += "requires specifying either quanta or precision."
            raise RuntimeError(msg)
        if quanta is not None:
            if precision is not None:
                raise RuntimeError("Cannot set both quanta and precision")
            if hasattr(quanta, "__len__"):
                dquanta = np.asarray(quanta)
                if dquanta.shape != leading_shape:
                    msg = "If not a scalar, quanta must have the same shape as the "
                    msg += "leading dimensions of the array"
                    raise ValueError(msg)
                dquanta = dquanta.astype(arr.dtype)
            else:
                dquanta = quanta * np.ones(leading_shape, dtype=arr.dtype)
        else:
            dquanta = None
        idata, foff, gains = float_to_int(arr, quanta=dquanta, precision=precision)
        (compressed, starts, nbytes) = encode_flac(idata, level, use_threads=use_threads)
        return (compressed, starts, nbytes, foff, gains)
    elif arr.dtype == np.dtype(np.int32) or arr.dtype == np.dtype(np.int64):
        (compressed, starts, nbytes) = encode_flac(np.ascontiguousarray(arr), level, use_threads=use_threads)
        return (compressed, starts, nbytes, None, None)
    else:
        raise ValueError(f"Unsupported data type '{arr.dtype}'")
