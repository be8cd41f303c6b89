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


# binding-layer entry point per dtype, by NAME (looked up when called: the CPU tests swap these module attributes)
_FLOAT_TO_INT = {np.dtype(np.float32): "wrap_float32_to_int32", np.dtype(np.float64): "wrap_float64_to_int64"}
_INT_TO_FLOAT = {np.dtype(np.int32): ("wrap_int32_to_float32", np.float32), np.dtype(np.int64): ("wrap_int64_to_float64", np.float64)}


def _streams_of(arr):
    """(leading shape, number of streams, stream length) of an array whose last axis is the stream axis."""
    lead = arr.shape[:-1]
    return lead, (int(np.prod(lead)) if lead else 1), arr.shape[-1]


def _per_stream(values, lead, what):
    """`values` as an array shaped like the leading axes: a scalar is broadcast, an array must already match
    (the reference's message for a mismatch, utils.py:287-290 / :305-308)."""
    if not hasattr(values, "__len__"):
        return None, values
    arr = np.asarray(values)
    if arr.shape != lead:
        msg = f"{what} array ({arr}) has shape that does not "
        msg += f"match leading shape of data ({arr.shape} != {lead})"
        raise RuntimeError(msg)
    return arr, None


def _quanta_for(data, lead, quanta, precision):
    """The quanta handed to the C layer: a zero-length array means "derive them from the data range"
    (libflacarray.pyx:142-144); `precision` p stands for quanta = rms / 10^p per stream (utils.py:284-296)."""
    if precision is not None:
        rms = np.std(data, axis=-1, keepdims=True).reshape(lead)
        p_arr, p_scalar = _per_stream(precision, lead, "precision")
        quanta = rms / 10 ** (p_arr.reshape(lead) if p_arr is not None else p_scalar)
    if quanta is None:
        return np.zeros(0, dtype=data.dtype)
    q_arr, q_scalar = _per_stream(quanta, lead, "quanta")
    if q_arr is None:
        q_arr = np.full(lead, q_scalar, dtype=data.dtype)
    return q_arr.reshape(-1).astype(data.dtype)


def float_to_int(data, quanta=None, precision=None):
    """Quantise float32 / float64 data to int32 / int64 with one offset and gain per stream (utils.py:246-342).

    Exactly one of `quanta` (scalar or one value per stream) and `precision` (decimal digits kept relative to each
    stream's rms) may be given; neither means quanta from each stream's range.  Returns (integers shaped like
    `data`, offsets, gains); offsets / gains have the leading shape, one element for a single stream."""
    if np.any(np.isnan(data)):
        raise RuntimeError("Cannot convert data with NaNs to integers")
    if quanta is not None and precision is not None:
        raise RuntimeError("Cannot specify both quanta and precision")
    convert = _FLOAT_TO_INT.get(data.dtype)
    if convert is None:
        raise ValueError("Only float32 and float64 data are supported")
    lead, n_stream, stream_size = _streams_of(data)
    ints, offsets, gains = globals()[convert](np.ascontiguousarray(data).reshape(-1), n_stream, stream_size, _quanta_for(data, lead, quanta, precision))
    aux_shape = lead if lead else (-1,)
    return (ints.reshape(data.shape), offsets.reshape(aux_shape), gains.reshape(aux_shape))


def int_to_float(idata, offset, gain):
    """Restore float32 / float64 data from int32 / int64 with the per-stream offset and gain (utils.py:346-408)."""
    pair = _INT_TO_FLOAT.get(idata.dtype)
    if pair is None:
        raise ValueError("Input data should be int32 or int64")
    restore, ftype = pair
    lead, n_stream, stream_size = _streams_of(idata)
    if n_stream == 1 and len(lead) <= 1:
        offset, gain = ensure_one_element(offset, ftype), ensure_one_element(gain, ftype)
    else:
        for name, arr in (("Offset", offset), ("Gain", gain)):
            if arr.shape != lead:
                raise ValueError(f"{name} array has shape {arr.shape}, expected shape {lead}")
    flat = globals()[restore](np.ascontiguousarray(idata).reshape(-1), n_stream, stream_size, offset.reshape(-1), gain.reshape(-1))
    return flat.reshape(idata.shape)


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
