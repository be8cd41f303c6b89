"""Interoperability with a REAL libFLAC, where the machine has one (ctypes.util.find_library("FLAC")).
Neither the build image nor the GPU image of this round carries libFLAC, so these tests skip there; they are the
recipe that pins the format against the reference's engine wherever the library exists:
  ours -> libFLAC : streams of the oracle / the HIP encoder decode to the input with the real decoder,
  libFLAC -> ours : streams of the real encoder (the reference's settings) decode with the oracle / the HIP decoder,
  size            : our compressed size stays within 2 % of libFLAC's on the benchmark's signal.
"""
import numpy as np
import pytest

from tests.conftest import full_range_i32, sinusoid_noise_i32

harness = pytest.importorskip("oracle.libflac_harness")
pytestmark = pytest.mark.skipif(not harness.available(), reason="no system libFLAC on this machine (parity with libFLAC stays unpinned)")


@pytest.fixture(scope="module")
def fa():
    import flacarray_amd

    return flacarray_amd


def _cases():
    return [
        sinusoid_noise_i32(4, 20000, seed=1),
        full_range_i32((2, 9000), seed=2),
        np.zeros((2, 5000), np.int32),
        (np.arange(3 * 4097, dtype=np.int64).reshape(3, 4097) * 977 % 100003 - 50000).astype(np.int32),
    ]


@pytest.mark.parametrize("level", [0, 3, 5, 8])
def test_oracle_streams_decode_with_libflac(oracle, level):
    for x in _cases():
        blob, st, nb = oracle.encode_i32(x, level)
        assert np.array_equal(harness.decode_i32(blob, st, nb, x.shape[1]), x)


@pytest.mark.parametrize("level", [0, 3, 5, 8])
def test_libflac_streams_decode_with_oracle(oracle, level):
    for x in _cases():
        blob, st, nb = harness.encode_i32(x, level)
        assert np.array_equal(oracle.decode_i32(blob, st, nb, x.shape[1]), x)
        f0, f1 = 17, min(x.shape[1], 4200)
        assert np.array_equal(oracle.decode_i32(blob, st, nb, x.shape[1], f0, f1), x[:, f0:f1])


def test_compressed_size_close_to_libflac(oracle):
    x = sinusoid_noise_i32(8, 1 << 16, seed=3)
    ours = oracle.encode_i32(x, 5)[0].size
    theirs = harness.encode_i32(x, 5)[0].size
    assert abs(ours - theirs) <= 0.02 * theirs, (ours, theirs)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [3, 5, 8])
def test_hip_and_libflac_read_each_other(fa, level):
    import torch

    for x in _cases():
        blob, st, nb = fa.encode_flac(x, level)
        assert np.array_equal(harness.decode_i32(blob, st, nb, x.shape[1]), x)
        lb, ls, ln = harness.encode_i32(x, level)
        dev = torch.device("cuda", 0)
        y = fa.decode_flac_device(torch.from_numpy(lb).to(dev), torch.from_numpy(ls).to(dev), torch.from_numpy(ln).to(dev), x.shape[1])
        assert np.array_equal(y.cpu().numpy(), x)
