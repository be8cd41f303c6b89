"""The HIP path against the only outputs of the reference's libFLAC path that exist in its tree: the compressed sizes
and restored values its executed notebooks print for seeded `create_fake_data` arrays
(tests/golden/reference_published.py holds the numbers, their cell citations and the header arithmetic)."""
import numpy as np
import pytest

from tests.golden import reference_published as P
from tests.test_oracle import published_values_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(3))
def test_flacarray_nbytes_is_the_published_size(case):
    """`FlacArray.from_array(arr, quanta= / precision=).nbytes`, the call of cookbook cells 18 / 28 and tutorial cell 7,
    through the HIP encoder: total frame bytes == the published size minus libFLAC's 86 B of stream header per stream."""
    import flacarray_amd as fa

    shape, dtype, kw, published, where = P.SIZES[case]
    arr = P.fake_data(shape, dtype)
    f = fa.FlacArray.from_array(arr, **kw)
    n_stream = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    assert f.nbytes - n_stream * P.own_stream_header(shape[-1], 5) == P.frame_bytes_published(P.SIZES[case]), where
    assert f.nbytes == published + 14 * n_stream
    # and the reference's own bound on the round trip (tests/array.py:251-260: half a quantum), plus the rounding of the
    # quantisation's subtract / multiply and the restore's multiply / add at the magnitude of the data (4 eps |x|: with
    # quanta 1e-8 on float32 values near 2 the float resolution, 1.2e-7, is what is left)
    back = f.to_array()
    quanta = kw["quanta"] if "quanta" in kw else (np.std(arr, axis=-1, keepdims=True) / 10 ** kw["precision"])
    assert np.all(np.abs(back - arr) <= 0.5 * quanta + 4 * np.finfo(dtype).eps * np.abs(arr).max())


def test_restored_values_of_cookbook_cell_12():
    """(1000, 100000) float32, quanta 1e-7, stream 500, the call of cookbook cell 12: `to_array(keep=, stream_slice=)`."""
    import flacarray_amd as fa

    rows = np.stack([P.fake_data_stream(P.VALUES_SHAPE, np.float32, P.VALUES_STREAM + d) for d in (-1, 0, 1)])
    f = fa.FlacArray.from_array(rows, quanta=P.VALUES_QUANTA)
    keep = np.zeros(3, dtype=bool)
    keep[1] = True
    sub = f.to_array(keep=keep)  # what the cell obtained: its negative stream_slice reached the C layer as "no slice"
    assert sub.shape == (1, P.VALUES_SHAPE[1])
    ints, _, gains = fa.float_to_int(rows, quanta=P.VALUES_QUANTA)
    published_values_check(sub[0], ints[1], gains[1])
    tail = f.to_array(keep=keep, stream_slice=slice(-10000, None, 1))  # what the cell meant
    assert np.array_equal(tail[0], sub[0, -10000:])
