"""array_decompress_slice / array_decompress (reference: src/flacarray/decompress.py:18-205)."""
import numpy as np

from .libflacarray import decode_flac
from .utils import ensure_one_element, function_timer, int_to_float, keep_select, select_keep_indices


@function_timer
def array_decompress_slice(
    compressed,
    stream_size,
    stream_starts,
    stream_nbytes,
    stream_offsets=None,
    stream_gains=None,
    keep=None,
    first_stream_sample=None,
    last_stream_sample=None,
    is_int64=False,
    use_threads=False,
    no_flatten=False,
):
    """Decompress a slice of a FLAC encoded array and restore the original data type.

    Both offsets and gains given -> float output; neither -> integer output; one of them ->
    RuntimeError.  `keep` (bool mask shaped like the starts) selects streams: the result is then
    the 2-D array of kept streams plus the list of their indices.  None / negative
    first/last_stream_sample decode whole streams (decompress.py:76-79).

    Returns (output array, list of stream indices or None).
    """
    if first_stream_sample is None:
        first_stream_sample = -1
    if last_stream_sample is None:
        last_stream_sample = -1

    is_scalar = False
    if not isinstance(stream_starts, np.ndarray) or (len(stream_starts.shape) == 1 and stream_starts.shape[0] == 1):
        is_scalar = True
        stream_starts = ensure_one_element(stream_starts, np.int64)
        stream_nbytes = ensure_one_element(stream_nbytes, np.int64)
        if stream_offsets is not None:
            ftype = np.float64 if is_int64 else np.float32
            stream_offsets = ensure_one_element(stream_offsets, ftype)
            stream_gains = ensure_one_element(stream_gains, ftype)

    starts, nbytes, indices = keep_select(keep, stream_starts, stream_nbytes)
    offsets = select_keep_indices(stream_offsets, indices)
    gains = select_keep_indices(stream_gains, indices)

    if stream_offsets is not None:
        if stream_gains is None:
            raise RuntimeError("When specifying offsets, you must also provide the gains")
        idata = decode_flac(
            compressed, starts, nbytes, stream_size, first_sample=first_stream_sample, last_sample=last_stream_sample,
            use_threads=use_threads, is_int64=is_int64,
        )
        arr = int_to_float(idata, offsets, gains)
    else:
        if stream_gains is not None:
            raise RuntimeError("When specifying gains, you must also provide the offsets")
        arr = decode_flac(
            compressed, starts, nbytes, stream_size, first_sample=first_stream_sample, last_sample=last_stream_sample,
            use_threads=use_threads, is_int64=is_int64,
        )
    if is_scalar and not no_flatten:
        return (arr.reshape((-1)), indices)
    return (arr, indices)


@function_timer
def array_decompress(
    compressed,
    stream_size,
    stream_starts,
    stream_nbytes,
    stream_offsets=None,
    stream_gains=None,
    first_stream_sample=None,
    last_stream_sample=None,
    is_int64=False,
    use_threads=False,
    no_flatten=False,
):
    """Decompress a FLAC encoded array and restore the original data type (decompress.py:145-205)."""
    arr, _ = array_decompress_slice(
        compressed,
        stream_size,
        stream_starts,
        stream_nbytes,
        stream_offsets=stream_offsets,
        stream_gains=stream_gains,
        keep=None,
        first_stream_sample=first_stream_sample,
        last_stream_sample=last_stream_sample,
        is_int64=is_int64,
        use_threads=use_threads,
        no_flatten=no_flatten,
    )
    return arr
