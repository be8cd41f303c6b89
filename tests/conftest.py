import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the checker is compiled here, before any test can have initialised the GPU (oracle.lib() never builds)
    from oracle import oracle as O

    O.build()


def sinusoid_noise_i32(n_ch, n_samp, seed=123456789, amp=2**16):
    """S2 of SURVEY.md 8(d): int version of the reference generator (src/flacarray/demo.py:75-97):
    rint(A*(dc_c + s_c*(2 sin(2pi 3f t) + 6 sin(2pi f t)) + N(0,1))), f = 5/T."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samp)
    f = 5.0 / n_samp
    dc = 5.0 * (rng.random((n_ch, 1)) - 0.5)
    wave = 2.0 * np.sin(2 * np.pi * 3 * f * t) + 6.0 * np.sin(2 * np.pi * f * t)
    scale = rng.random((n_ch, 1))
    x = dc + scale * wave + rng.normal(0.0, 1.0, (n_ch, n_samp))
    return np.rint(amp * x).astype(np.int32)


def sinusoid_noise_f32(n_ch, n_samp, seed=123456789):
    """S3: the same field before rint, float32, amplitude 1."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samp)
    f = 5.0 / n_samp
    dc = 5.0 * (rng.random((n_ch, 1)) - 0.5)
    wave = 2.0 * np.sin(2 * np.pi * 3 * f * t) + 6.0 * np.sin(2 * np.pi * f * t)
    scale = rng.random((n_ch, 1))
    return (dc + scale * wave + rng.normal(0.0, 1.0, (n_ch, n_samp))).astype(np.float32)


def full_range_i32(shape, seed=123456789):
    """Uniform full-range int32 with INT32_MIN / INT32_MAX at flat positions 0 / 1
    (reference recipe: src/flacarray/demo.py:58-68, tests/bindings.py:34-39)."""
    rng = np.random.default_rng(seed)
    flat = rng.integers(-(2**31), 2**31 - 1, size=int(np.prod(shape)), dtype=np.int64).astype(np.int32)
    flat[0] = -(2**31)
    if flat.size > 1:
        flat[1] = 2**31 - 1
    return flat.reshape(shape)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


def strip_seektable(blob, starts, nbytes):
    """Rewrite every stream of an encoded triple without its SEEKTABLE block (fLaC, STREAMINFO
    marked last, frames): the metadata layout libFLAC-written streams present to the decoder's
    frame-location logic."""
    blob = np.asarray(blob)
    parts, new_nb = [], []
    for s, n in zip(np.asarray(starts).reshape(-1), np.asarray(nbytes).reshape(-1)):
        st = blob[int(s) : int(s) + int(n)]
        assert bytes(st[:4]) == b"fLaC" and (st[4] & 0x7F) == 0
        off = 4
        while True:
            last = st[off] >> 7
            ln = (int(st[off + 1]) << 16) | (int(st[off + 2]) << 8) | int(st[off + 3])
            off += 4 + ln
            if last:
                break
        si = st[4:42].copy()
        si[0] = 0x80  # STREAMINFO is now the last metadata block
        out = np.concatenate([st[:4], si, st[off:]])
        parts.append(out)
        new_nb.append(out.shape[0])
    new_nb = np.array(new_nb, dtype=np.int64)
    new_st = np.concatenate([[0], np.cumsum(new_nb)[:-1]]).astype(np.int64)
    return np.concatenate(parts), new_st, new_nb


class FakeH5Dataset:
    """The slice of the h5py.Dataset API that flacarray_amd.hdf5 uses (h5py is absent from this image)."""

    def __init__(self, shape, dtype):
        self._a = np.zeros(shape, dtype=dtype)
        self.attrs = {}

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)
    size = property(lambda self: self._a.size)

    def __getitem__(self, k):
        return self._a[k]

    def __setitem__(self, k, v):
        self._a[k] = v


class FakeH5Group(dict):
    def __init__(self):
        super().__init__()
        self.attrs = {}

    def create_dataset(self, name, shape, dtype=None):
        self[name] = FakeH5Dataset(shape, dtype)
        return self[name]


class FakeZarr3Group(dict):
    """zarr-3 style group: create_array(name, shape=, dtype=) instead of create_dataset."""

    def __init__(self):
        super().__init__()
        self.attrs = {}

    def create_array(self, name, shape=None, dtype=None):
        self[name] = FakeH5Dataset(shape, dtype)
        return self[name]
