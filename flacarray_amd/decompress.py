"""array_decompress_slice / array_decompress: stream selection and dtype restore in front of the
decoder (reference: src/flacarray/decompress.py:18-205).

Same signatures, return values and error behaviour as the reference; K6/K7 do the decoding on the
GPU with the float restore (`int_to_float`, K2) fused into the decoder's store for float data.
"""
import numpy as np

from .libflacarray import decode_flac, decode_flac_restore
from .utils import ensure_one_element, function_timer, int_to_float, keep_select, select_keep_indices


def _plausible_float_call(compressed, starts, nbytes, stream_size, first, last, offsets, gains):
    """True when the arguments are what the fused decode + restore takes as they are; anything else goes through
    decode_flac / int_to_float, whose checks raise the reference's errors (libflacarray.pyx:751-789, utils.py:367-386)."""
    try:
        return (
            isinstance(compressed, np.ndarray) and compressed.dtype == np.uint8 and compressed.ndim == 1 and compressed.flags.c_contiguous
            and isinstance(starts, np.ndarray) and starts.dtype == np.int64 and starts.flags.c_contiguous
            and isinstance(nbytes, np.ndarray) and nbytes.dtype == np.int64 and nbytes.shape == starts.shape and nbytes.flags.c_contiguous
            and stream_size > 0 and isinstance(offsets, np.ndarray) and isinstance(gains, np.ndarray)
            and offsets.size == starts.size and gains.size == starts.size
            and (first < 0 or last < 0 or (first < last <= stream_size))
        )
    except Exception:  # noqa: BLE001  (odd argument types: the plain path will say what is wrong with them)
        return False


def _single_stream(stream_starts):
    """True for scalars and 1-element arrays: the result is then flattened unless no_flatten."""
    return not isinstance(stream_starts, np.ndarray) or stream_starts.shape == (1,)


@function_timer
def array_decompress_slice(
    compressed, stream_size, stream_starts, stream_nbytes, stream_offsets=None, stream_gains=None, keep=None,
    first_stream_sample=None, last_stream_sample=None, is_int64=False, use_threads=False, no_flatten=False,
):
    """Decompress a slice of a FLAC encoded array and restore the original data type.

    Both offsets and gains given -> float output; neither -> integer output; one of them ->
    RuntimeError.  `keep` (bool mask shaped like the starts) selects streams: the result is then
    the 2-D array of kept streams plus the list of their indices.  None / negative
    first/last_stream_sample decode whole streams (decompress.py:76-79).

    Returns (output array, list of stream indices or None).
    """
    first = -1 if first_stream_sample is None else first_stream_sample
    last = -1 if last_stream_sample is None else last_stream_sample
    to_float = stream_offsets is not None
    if to_float and stream_gains is None:
        raise RuntimeError("When specifying offsets, you must also provide the gains")
    if not to_float and stream_gains is not None:
        raise RuntimeError("When specifying gains, you must also provide the offsets")

    single = _single_stream(stream_starts)
    if single:
        stream_starts = ensure_one_element(stream_starts, np.int64)
        stream_nbytes = ensure_one_element(stream_nbytes, np.int64)
        if to_float:
            ftype = np.float64 if is_int64 else np.float32
            stream_offsets = ensure_one_element(stream_offsets, ftype)
            stream_gains = ensure_one_element(stream_gains, ftype)

    starts, nbytes, indices = keep_select(keep, stream_starts, stream_nbytes)
    arr = None
    if to_float:
        # one trip over PCIe: the restore is fused into the decoder's store, the integers stay on the device
        sel_off, sel_gain = select_keep_indices(stream_offsets, indices), select_keep_indices(stream_gains, indices)
        if _plausible_float_call(compressed, starts, nbytes, stream_size, first, last, sel_off, sel_gain):
            arr = decode_flac_restore(compressed, starts, nbytes, stream_size, sel_off, sel_gain, first_sample=first, last_sample=last,
                                      is_int64=is_int64)
    if arr is None:
        arr = decode_flac(
            compressed, starts, nbytes, stream_size, first_sample=first, last_sample=last, use_threads=use_threads,
            is_int64=is_int64,
        )
        if to_float:
            arr = int_to_float(arr, select_keep_indices(stream_offsets, indices), select_keep_indices(stream_gains, indices))
    if single and not no_flatten:
        arr = arr.reshape(-1)
    return (arr, indices)


@function_timer
def array_decompress(
    compressed, stream_size, stream_starts, stream_nbytes, stream_offsets=None, stream_gains=None,
    first_stream_sample=None, last_stream_sample=None, is_int64=False, use_threads=False, no_flatten=False,
):
    """Decompress a FLAC encoded array and restore the original data type (decompress.py:145-205)."""
    return array_decompress_slice(
        compressed, stream_size, stream_starts, stream_nbytes, stream_offsets=stream_offsets, stream_gains=stream_gains,
        keep=None, first_stream_sample=first_stream_sample, last_stream_sample=last_stream_sample, is_int64=is_int64,
        use_threads=use_threads, no_flatten=no_flatten,
    )[0]
