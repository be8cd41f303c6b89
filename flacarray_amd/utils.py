"""Conversion wrappers and stream-selection helpers (reference: src/flacarray/utils.py).

`float_to_int` (utils.py:246-342), `int_to_float` (:346-408), `keep_select` (:411-449),
`select_keep_indices` (:452-459) with the reference's signatures, shapes and exceptions.
`keep_select` is vectorised (the reference walks every stream with np.nditer per call).
"""
import logging
import os

import numpy as np

from .libflacarray import wrap_float32_to_int32, wrap_float64_to_int64, wrap_int32_to_float32, wrap_int64_to_float64

log = logging.getLogger("flacarray")
_lvl = os.environ.get("FLACARRAY_LOGLEVEL", os.environ.get("FLACARRAY_LOG_LEVEL"))
if _lvl is not None and hasattr(logging, _lvl):
    log.setLevel(getattr(logging, _lvl))



def function_timer(f):
    """Placeholder for the reference's env-driven timers (utils.py:95-151): no-op decorator."""
    return f


def ensure_one_element(value, dtype=None):
    """1-element array of `dtype` from a scalar, or the array itself after checking that it has
    shape (1,) and that dtype (utils.py:177-206; same ValueError texts)."""
    if isinstance(value, np.ndarray):
        if value.shape != (1,):
            raise ValueError("Input array does not have a single element.")
        if dtype is not None and value.dtype != np.dtype(dtype):
            raise ValueError(f"Input has dtype {value.dtype}, not {dtype}")
        return value
    if dtype is None:
        raise ValueError("Input is a scalar, dtype must be specified")
    return np.array([value], dtype=dtype)


def compressed_dtype(n_channel, offsets, gains):
    """dtype of the uncompressed data implied by stored metadata (utils.py:209-243)."""
    is_int = offsets is None or gains is None
    if n_channel == 2:
        return np.dtype(np.int64) if is_int else np.dtype(np.float64)
    return np.dtype(np.int32) if is_int else np.dtype(np.float32)


def float_to_int(data, quanta=None, precision=None):
    """Convert float32 data to int32 with a per-stream offset and gain.

    Returns (integer data, offset array, gain array); single-stream input gives 1-element
    offset/gain arrays (utils.py:325-342).
    """
    if np.any(np.isnan(data)):
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if quanta is not None and precision is not None:
        raise RuntimeError("Cannot specify both quanta and precision")
    if data.dtype != np.dtype(np.float32) and data.dtype != np.dtype(np.float64):
        raise ValueError("Only float32 and float64 data are supported")
    is_f64 = data.dtype == np.dtype(np.float64)

    leading_shape = data.shape[:-1]
    n_stream = 1 if len(leading_shape) == 0 else int(np.prod(leading_shape))
    stream_size = data.shape[-1]

    if precision is not None:
        rms = np.std(data, axis=-1, keepdims=True)
        if hasattr(precision, "__len__"):
            precision = np.asarray(precision)
            if precision.shape != leading_shape:
                msg = f"precision array ({precision}) has shape that does not "
                msg += f"match leading shape of data ({precision.shape} != {leading_shape})"
                raise RuntimeError(msg)
            quanta = rms.reshape(leading_shape) / 10 ** precision.reshape(leading_shape)
        else:
            quanta = rms.reshape(leading_shape) / 10**precision

    if quanta is None:
        quanta = np.zeros(0, dtype=data.dtype)  # "compute it from the data range"
    elif hasattr(quanta, "__len__"):
        quanta = np.asarray(quanta)
        if quanta.shape != leading_shape:
            msg = f"quanta array ({quanta}) has shape that does not "
            msg += f"match leading shape of data ({quanta.shape} != {leading_shape})"
            raise RuntimeError(msg)
    else:
        quanta = quanta * np.ones(leading_shape, dtype=data.dtype)

    output, offsets, gains = (wrap_float64_to_int64 if is_f64 else wrap_float32_to_int32)(
        np.ascontiguousarray(data).reshape((-1,)), n_stream, stream_size, np.asarray(quanta).reshape((-1,)).astype(data.dtype)
    )
    if len(leading_shape) == 0:
        return (output.reshape(data.shape), offsets.reshape((-1,)), gains.reshape((-1,)))
    return (output.reshape(data.shape), offsets.reshape(leading_shape), gains.reshape(leading_shape))


def int_to_float(idata, offset, gain):
    """Restore float32 data from int32 (utils.py:346-408)."""
    if idata.dtype != np.dtype(np.int32) and idata.dtype != np.dtype(np.int64):
        raise ValueError("Input data should be int32 or int64")
    is_i64 = idata.dtype == np.dtype(np.int64)
    ftype = np.float64 if is_i64 else np.float32
    leading_shape = idata.shape[:-1]
    if len(leading_shape) == 0 or (len(leading_shape) == 1 and leading_shape[0] == 1):
        n_stream = 1
        offset = ensure_one_element(offset, ftype)
        gain = ensure_one_element(gain, ftype)
    else:
        n_stream = int(np.prod(leading_shape))
        if offset.shape != leading_shape:
            raise ValueError(f"Offset array has shape {offset.shape}, expected shape {leading_shape}")
        if gain.shape != leading_shape:
            raise ValueError(f"Gain array has shape {gain.shape}, expected shape {leading_shape}")
    stream_size = idata.shape[-1]
    result = (wrap_int64_to_float64 if is_i64 else wrap_int32_to_float32)(
        np.ascontiguousarray(idata).reshape((-1,)), n_stream, stream_size, offset.reshape((-1,)), gain.reshape((-1,))
    )
    return result.reshape(idata.shape)


def keep_select(keep, stream_starts, stream_nbytes):
    """Select the streams flagged in the bool mask `keep` (utils.py:411-449).

    Returns (starts, nbytes, indices): 1-D int64 arrays of the kept streams in C order and the
    list of their multi-indices; (stream_starts, stream_nbytes, None) when keep is None.
    """
    if keep is None:
        return (stream_starts, stream_nbytes, None)
    if keep.shape != stream_starts.shape:
        raise RuntimeError("The keep array should have the same shape as stream_starts")
    if keep.shape != stream_nbytes.shape:
        raise RuntimeError("The keep array should have the same shape as stream_starts")
    sel = np.nonzero(keep)
    indices = list(zip(*(ax.tolist() for ax in sel)))
    return (
        np.ascontiguousarray(stream_starts[sel], dtype=np.int64),
        np.ascontiguousarray(stream_nbytes[sel], dtype=np.int64),
        indices,
    )


def select_keep_indices(arr, indices):
    """Extract array elements with a list of multi-indices (utils.py:452-459)."""
    if arr is None:
        return None
    if indices is None:
        return arr
    if len(indices) == 0:
        return np.zeros(0, dtype=arr.dtype)
    return np.array(arr[tuple(np.array(indices).T)], dtype=arr.dtype)
