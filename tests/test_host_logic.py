"""CPU tests of the host side: C-ABI exports, validation and error text, selection helpers, the
FlacArray surface.  Compute calls need a GPU; here the four binding-level wrappers are replaced
by a TEST-ONLY stand-in built on the oracle so the Python logic above them runs on the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import flacarray_amd as fa
from tests.conftest import ROOT, full_range_i32, sinusoid_noise_i32
from flacarray_amd import _lib, libflacarray


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "flacarray_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int64_t|int|void\*?|const char\*)\s+(\w+)\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert b"gfx950" in _lib.lib().fa_version()


def test_single_pass_queries_cover_every_geometry():
    """The capacity / workspace / "is there a single-pass encoder" queries are pure arithmetic (no device): every valid
    geometry has one since round 4 -- the reference's own test shapes (tests/bindings.py:165-230), levels 0-2, int64 --
    the worst-case capacity is one 16 640-byte slot per frame and channel + the stream headers, and the workspace of the
    geometries K3G serves holds a few thousand slots, not one per frame."""
    L = _lib.lib()
    for n_stream, size, level in ((12, 1000, 5), (1, 10000, 5), (12, 1000, 0), (3, 1152 * 5 + 1, 2), (1000, 100000, 5), (4096, 1 << 20, 5)):
        assert L.fa_encode_single_pass_supported(n_stream, size, level) == 1
        block = 1152 if level <= 2 else 4096
        nf = -(-size // block)
        hb = 46 + 18 * nf
        assert L.fa_encode_capacity_bytes(n_stream, size, level) == n_stream * (nf * 16640 + hb)
        assert L.fa_encode_capacity_bytes_i64(n_stream, size, level) == n_stream * (nf * 2 * 16640 + hb)
        for ws in (L.fa_encode_single_pass_workspace_bytes(n_stream, size, level), L.fa_encode_single_pass_workspace_bytes_i64(n_stream, size, level)):
            assert 0 < ws <= 28 * n_stream * nf + 2 * 2049 * 2 * 16640 + 65536
    assert L.fa_encode_single_pass_workspace_bytes_i64(1024, 1 << 20, 5) < 150e6 < L.fa_encode_workspace_bytes_i64(1024, 1 << 20, 5)
    assert L.fa_encode_single_pass_supported(12, 1000, 9) == 0 and L.fa_encode_single_pass_supported(0, 1000, 5) == 0
    assert L.fa_encode_capacity_bytes(12, 1000, 9) == -1


def test_reference_named_header_compiles_without_libflac(tmp_path):
    """include/flacarray.h is what `cdef extern from "flacarray.h"` (libflacarray.pyx:18) finds when the binding
    is built against this library: the ten prototypes the binding declares (libflacarray.pyx:19-110) and the
    reference's unprefixed error names, with no <FLAC/...> include."""
    import subprocess

    text = open(os.path.join(ROOT, "include", "flacarray.h")).read()
    assert "#include <FLAC" not in text and "#include" in text
    src = tmp_path / "use.c"
    src.write_text(
        '#include "flacarray.h"\n'
        "int use(void) {\n"
        "  void* f[] = {(void*)encode_i32, (void*)encode_i32_threaded, (void*)encode_i64, (void*)encode_i64_threaded,\n"
        "               (void*)decode_i32, (void*)decode_i64, (void*)float32_to_int32, (void*)float64_to_int64,\n"
        "               (void*)int32_to_float32, (void*)int64_to_float64};\n"
        "  return (int)sizeof f + ERROR_NONE + ERROR_ALLOC + ERROR_INVALID_LEVEL + ERROR_ZERO_NSTREAM + ERROR_ZERO_STREAMSIZE +\n"
        "         ERROR_ENCODE_INIT + ERROR_ENCODE_PROCESS + ERROR_DECODE_INIT + ERROR_DECODE_PROCESS + ERROR_DECODE_STREAMSIZE +\n"
        "         ERROR_DECODE_SAMPLE_RANGE + ERROR_DECODE_SEEK + ERROR_CONVERT_TYPE + ERROR_ENCODE_FINISH + ERROR_DECODE_FINISH;\n"
        "}\n"
    )
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-Wno-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    # same bit values as the reference's table (flacarray.h:20-40)
    vals = dict(re.findall(r"#define (FA_ERROR_\w+) \(1 << (\d+)\)", open(os.path.join(ROOT, "include", "flacarray_hip.h")).read()))
    assert vals["FA_ERROR_DECODE_INIT"] == "13" and vals["FA_ERROR_DECODE_SAMPLE_RANGE"] == "17" and vals["FA_ERROR_CONVERT_TYPE"] == "19"


def test_no_silent_cpu_fallback():
    if _lib.lib().fa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        fa.array_compress(np.zeros((2, 100), np.int32))
    with pytest.raises(RuntimeError, match="no HIP device"):
        fa.decode_flac(np.zeros(10, np.uint8), np.zeros(1, np.int64), np.zeros(1, np.int64), 10)
    # the raw C ABI reports FA_ERROR_DEVICE instead of computing on the host
    x = np.zeros(8, np.int32)
    nb, raw, st = ctypes.c_int64(0), ctypes.c_void_p(None), np.zeros(1, np.int64)
    rc = _lib.lib().encode_i32(x.ctypes.data, 1, 8, 5, ctypes.byref(nb), st.ctypes.data, ctypes.byref(raw))
    assert rc == _lib.ERROR_DEVICE and raw.value is None


def test_encode_decode_validation_text():
    """libflacarray.pyx:551-559, :751-789"""
    with pytest.raises(RuntimeError, match="Only 32bit or 64bit integer data is supported"):
        fa.encode_flac(np.zeros((2, 8), np.float32), 5)
    with pytest.raises(RuntimeError, match="Only C-contiguous arrays are supported"):
        fa.encode_flac(np.zeros((8, 2), np.int32).T, 5)
    with pytest.raises(RuntimeError, match="FLAC only supports compression levels 0-8"):
        fa.encode_flac(np.zeros((2, 8), np.int32), 9)
    c, s, n = np.zeros(10, np.uint8), np.zeros(1, np.int64), np.zeros(1, np.int64)
    with pytest.raises(RuntimeError, match="Compressed data should be of type uint8"):
        fa.decode_flac(c.astype(np.int8), s, n, 10)
    with pytest.raises(RuntimeError, match="starts data should be of type int64"):
        fa.decode_flac(c, s.astype(np.int32), n, 10)
    with pytest.raises(RuntimeError, match="nbytes data should be of type int64"):
        fa.decode_flac(c, s, n.astype(np.int32), 10)
    with pytest.raises(RuntimeError, match="non-zero output stream size"):
        fa.decode_flac(c, s, n, 0)
    with pytest.raises(RuntimeError, match="one dimensional"):
        fa.decode_flac(c.reshape(2, 5), s, n, 10)
    with pytest.raises(RuntimeError, match="last_sample is beyond end of stream"):
        fa.decode_flac(c, s, n, 10, 0, 11)
    with pytest.raises(RuntimeError, match="first_sample is beyond last element of stream"):
        fa.decode_flac(c, s, n, 10, 10, 10)
    with pytest.raises(RuntimeError, match="first_sample is larger than last_sample"):
        fa.decode_flac(c, s, n, 10, 5, 5)


def test_array_compress_argument_errors():
    """compress.py:47-59,84"""
    with pytest.raises(ValueError, match="zero-sized"):
        fa.array_compress(np.zeros((0, 4), np.int32))
    with pytest.raises(RuntimeError, match="requires specifying either quanta or precision"):
        fa.array_compress(np.zeros((2, 4), np.float32))
    with pytest.raises(RuntimeError, match="Cannot set both quanta and precision"):
        fa.array_compress(np.zeros((2, 4), np.float32), quanta=1e-3, precision=3)
    with pytest.raises(ValueError, match="Unsupported data type"):
        fa.array_compress(np.zeros((2, 4), np.int16))
    with pytest.raises(RuntimeError, match="no HIP device"):  # int64 is supported; without a GPU it fails loudly
        fa.encode_flac(np.zeros((2, 4), np.int64), 5)


def test_keep_select_matches_reference_semantics():
    rng = np.random.default_rng(1)
    starts = rng.integers(0, 1000, (4, 3)).astype(np.int64)
    nbytes = rng.integers(1, 50, (4, 3)).astype(np.int64)
    keep = rng.random((4, 3)) > 0.5
    s, n, idx = fa.keep_select(keep, starts, nbytes)
    # the reference walks np.nditer in C order with multi_index (utils.py:432-449)
    exp = [(i, j) for i in range(4) for j in range(3) if keep[i, j]]
    assert idx == exp and s.dtype == np.int64 and s.tolist() == [starts[i] for i in exp] and n.tolist() == [nbytes[i] for i in exp]
    assert fa.keep_select(None, starts, nbytes) == (starts, nbytes, None)
    with pytest.raises(RuntimeError, match="same shape as stream_starts"):
        fa.keep_select(keep[:2], starts, nbytes)
    from flacarray_amd.utils import select_keep_indices

    off = rng.random((4, 3)).astype(np.float32)
    assert select_keep_indices(off, idx).tolist() == [off[i] for i in exp]
    assert select_keep_indices(None, idx) is None and select_keep_indices(off, None) is off


def test_shard_range_is_array_split():
    from flacarray_amd.dist import shard_counts, shard_range

    for n in (1, 7, 8, 4096, 32768, 10):
        for w in (1, 2, 3, 8):
            parts = np.array_split(np.arange(n), w)  # mpi.py:84
            for r in range(w):
                lo, hi = shard_range(n, w, r)
                assert hi - lo == len(parts[r]) and (hi == lo or parts[r][0] == lo)
            assert sum(shard_counts(n, w)) == n


# ---- TEST-ONLY CPU stand-in for the binding layer (the oracle), to exercise the Python logic ----
@pytest.fixture
def cpu_backend(monkeypatch, oracle):
    def enc(flat, n_stream, stream_size, level):
        return oracle.encode_i32(np.asarray(flat).reshape(n_stream, stream_size), level)

    def dec(comp, starts, nbytes, n_stream, stream_size, first, last, use_threads):
        return oracle.decode_i32(comp, starts, nbytes, stream_size, first, last).reshape(-1)

    def f2i(flat, n_stream, stream_size, quanta):
        q = quanta if len(quanta) == n_stream else None
        o, off, g = oracle.float32_to_int32(np.asarray(flat).reshape(n_stream, stream_size), q)
        return o.reshape(-1), off, g

    def i2f(idata, n_stream, stream_size, offsets, gains):
        return oracle.int32_to_float32(np.asarray(idata).reshape(n_stream, stream_size), offsets, gains).reshape(-1)

    def enc_f32(data, quanta, level):  # the one-trip float32 entry: quantise, then encode
        lead = data.shape[:-1] if data.ndim > 1 else (1,)
        flat = np.ascontiguousarray(data).reshape(int(np.prod(lead)), data.shape[-1])
        ints, off, g = oracle.float32_to_int32(flat, quanta)
        blob, st, nb = oracle.encode_i32(ints, level)
        return blob, st.reshape(lead), nb.reshape(lead), off.reshape(lead), g.reshape(lead)

    import flacarray_amd.compress as C
    import flacarray_amd.utils as U

    def enc_f64(data, quanta, level):  # the float64 twin
        lead = data.shape[:-1] if data.ndim > 1 else (1,)
        flat = np.ascontiguousarray(data).reshape(int(np.prod(lead)), data.shape[-1])
        ints, off, g = oracle.float64_to_int64(flat, quanta)
        blob, st, nb = oracle.encode_i64(ints, level)
        return blob, st.reshape(lead), nb.reshape(lead), off.reshape(lead), g.reshape(lead)

    import flacarray_amd.decompress as D

    monkeypatch.setattr(C, "encode_flac_f32", enc_f32)
    monkeypatch.setattr(C, "encode_flac_f64", enc_f64)
    monkeypatch.setattr(D, "decode_flac_restore", lambda *a, **k: None)  # (the fused decode + restore: decode, then int_to_float)
    monkeypatch.setattr(libflacarray, "wrap_encode_i32", enc)
    monkeypatch.setattr(libflacarray, "wrap_encode_i32_threaded", enc)
    monkeypatch.setattr(libflacarray, "wrap_decode_i32", dec)
    monkeypatch.setattr(U, "wrap_float32_to_int32", f2i)
    monkeypatch.setattr(U, "wrap_int32_to_float32", i2f)
    return oracle


def _check_array_surface():
    """tests/array.py:26-146 (helpers), :167-232 (slicing shapes) restated."""
    for shape in ((4, 3, 1000), (10000,)):
        x = full_range_i32(shape)
        comp, starts, nbytes, off, gain = fa.array_compress(x, level=5)
        assert off is None and gain is None  # tests/array.py:57-60
        lead = shape[:-1] if len(shape) > 1 else (1,)
        assert starts.shape == lead and nbytes.shape == lead and comp.dtype == np.uint8
        y = fa.array_decompress(comp, shape[-1], starts, nbytes)
        assert y.shape == x.shape and np.array_equal(y, x)
        n = shape[-1]
        y = fa.array_decompress(comp, n, starts, nbytes, first_stream_sample=n // 2 - 5, last_stream_sample=n // 2 + 5)
        assert np.array_equal(y, x[..., n // 2 - 5 : n // 2 + 5])
        xf = (np.random.default_rng(0).normal(0, 1, shape)).astype(np.float32)
        comp, starts, nbytes, off, gain = fa.array_compress(xf, level=5, quanta=1e-6)
        assert off.shape == lead and gain.shape == lead and off.dtype == np.float32
        yf = fa.array_decompress(comp, n, starts, nbytes, stream_offsets=off, stream_gains=gain)
        assert yf.dtype == np.float32 and np.allclose(yf, xf, atol=1e-5)  # tests/array.py: atol = 10 * quanta
    # numpy-style slicing: shapes must equal numpy's (tests/array.py:167-232)
    x = sinusoid_noise_i32(120, 100, seed=8).reshape(4, 3, 10, 100)
    f = fa.FlacArray.from_array(x)
    assert f.shape == x.shape and f.stream_size == 100 and f.leading_shape == (4, 3, 10) and f.dtype == np.int32
    keys = [
        (slice(None), slice(None), slice(None), slice(None)), (1, slice(None)), (slice(1, 3), 2), (0, 1, 2),
        (0, 1, 2, 5), (slice(None), 1, slice(2, 8), slice(10, 20)), (3, slice(None), slice(None), slice(50, None)),
        (slice(0, 0),), (1, 2, slice(None), slice(99, 100)), 2, (slice(None), slice(None), 9, slice(None, 10)),
        (0, 0, 0, slice(40, 40)),
    ]
    for key in keys:
        assert f[key].shape == x[key].shape, key
        assert np.array_equal(f[key], x[key]), key
    # the reference's own key list (tests/array.py:181-194), shapes AND values
    for key in [
        (slice(0)), (slice(1, 3)), (slice(3, 1)), (slice(3, 1, -1)), (1, 2, 5, 50), (1, 2, 5),
        (2, slice(0, 1, 1), slice(0, 1, 1), slice(None)), (1, slice(1, 3, 1), slice(6, 8, 1), 50),
        (slice(1, 3, 1), 2, slice(6, 8, 1), slice(60, 80, 1)), (2, 1, slice(2, 8, 2), slice(80, 120, 1)),
        (2, 1, slice(2, 8, 2), slice(80, None)), (2, 1, slice(2, 8, 2), slice(None, 10)),
    ]:
        assert np.shape(f[key]) == np.shape(x[key]), key
        if key == slice(3, 1, -1):
            # streams are selected through a keep mask (array.py:339-378), so a negative leading step
            # keeps the ascending order: the reference pins only the shape here
            assert np.array_equal(f[key], x[2:4]), key
        else:
            assert np.array_equal(f[key], x[key]), key
    x1 = sinusoid_noise_i32(1, 10000, seed=9)[0]
    f1 = fa.FlacArray.from_array(x1)
    for key in [(slice(0)), (slice(1, 3)), (100,)]:  # tests/array.py:219
        assert np.shape(f1[key]) == np.shape(x1[key]) and np.array_equal(f1[key], x1[key]), key
    for key in (slice(None), slice(100, 200), 77):
        assert np.array_equal(f1[key], x1[key]) and np.shape(f1[key]) == np.shape(x1[key])
    # to_array with keep mask + indices (array.py:518-584)
    keep = np.zeros((4, 3, 10), bool)
    keep[1, 2, 3] = keep[3, 0, 9] = True
    arr, idx = f.to_array(keep=keep, keep_indices=True)
    assert idx == [(1, 2, 3), (3, 0, 9)] and np.array_equal(arr, np.stack([x[1, 2, 3], x[3, 0, 9]]))
    assert np.array_equal(f.to_array(stream_slice=slice(10, 20)), x[..., 10:20])
    assert np.array_equal(f.to_array(), x) and f == fa.FlacArray(f)
    with pytest.raises(RuntimeError):
        f[0] = 1
    with pytest.raises(ValueError, match="stride==1"):
        f[0, 0, 0, ::2]


def test_array_surface_cpu_standin(cpu_backend):
    _check_array_surface()


@pytest.mark.gpu
def test_array_surface_gpu():
    _check_array_surface()


def test_hdf5_v1_layout_roundtrip(oracle):
    """write_compressed -> read_compressed on the reference's format-version-1 layout
    (hdf5.py:195-245, hdf5_load_v1.py:136-157): names, dtypes, attrs, keep selection."""
    from flacarray_amd import hdf5 as H
    from tests.conftest import FakeH5Group, sinusoid_noise_i32

    x = sinusoid_noise_i32(6, 5000, seed=3).reshape(2, 3, 5000)
    blob, st, nb = oracle.encode_i32(x.reshape(6, 5000), 5)
    st, nb = st.reshape(2, 3), nb.reshape(2, 3)
    off = np.arange(6, dtype=np.float32).reshape(2, 3)
    gain = (1.0 + np.arange(6, dtype=np.float32)).reshape(2, 3)
    g = FakeH5Group()
    H.write_compressed(g, (2, 3), (2, 3), 5000, st, st, nb, off, gain, blob, 1)
    assert g.attrs["flacarray_format_version"] == "1" and g.attrs["flac_channels"] == "1"
    assert set(g) == {"stream_starts", "stream_bytes", "stream_offsets", "stream_gains", "compressed"}
    assert g["stream_starts"].attrs["stream_size"] == 5000 and g["stream_starts"].dtype == np.int64
    assert g["compressed"].dtype == np.uint8 and g["compressed"].shape == (blob.shape[0],)

    ls, gs, comp, nch, s2, n2, o2, g2, dist, idx = H.read_compressed(g)
    assert ls == (2, 3, 5000) and gs == (2, 3, 5000) and nch == 1 and idx is None
    assert np.array_equal(comp, blob) and np.array_equal(s2, st) and np.array_equal(n2, nb)
    assert np.array_equal(o2, off) and np.array_equal(g2, gain)
    assert np.array_equal(oracle.decode_i32(comp, s2.reshape(-1), n2.reshape(-1), 5000), x.reshape(6, 5000))

    keep = np.zeros((2, 3), dtype=bool)
    keep[0, 2] = keep[1, 1] = True
    ls, gs, comp, nch, s2, n2, o2, g2, dist, idx = H.read_compressed(g, keep=keep)
    assert ls == (2, 5000) and idx == [(0, 2), (1, 1)]
    assert np.array_equal(n2, [nb[0, 2], nb[1, 1]]) and np.array_equal(s2, [0, nb[0, 2]])
    assert np.array_equal(o2, [off[0, 2], off[1, 1]])
    assert np.array_equal(oracle.decode_i32(comp, s2, n2, 5000), x[[0, 1], [2, 1]])

    g.attrs["flacarray_format_version"] = "0"
    with pytest.raises(RuntimeError):
        H.read_compressed(g)


def test_zarr_v1_layout_roundtrip(oracle):
    """The Zarr layout is the HDF5 one; zarr-3 groups create arrays with create_array (zarr.py:236-241)."""
    from flacarray_amd import zarr as Z  # (the reference module name, zarr.py:145-447)
    from flacarray_amd.zarr import read_array, write_array  # noqa: F401  (the names the reference exports)
    from tests.conftest import FakeZarr3Group, sinusoid_noise_i32

    x = sinusoid_noise_i32(4, 3000, seed=8)
    blob, st, nb = oracle.encode_i32(x, 5)
    g = FakeZarr3Group()
    Z.write_compressed(g, (4,), (4,), 3000, st, st, nb, None, None, blob, 1)
    assert set(g) == {"stream_starts", "stream_bytes", "compressed"} and g.attrs["flacarray_format_version"] == "1"
    ls, gs, comp, nch, s2, n2, o2, g2, dist, idx = Z.read_compressed(g)
    assert ls == (4, 3000) and o2 is None and g2 is None
    assert np.array_equal(oracle.decode_i32(comp, s2, n2, 3000), x)


def test_pinned_result_pool_recycles_blocks(monkeypatch):
    """The pool behind host results of small reads (libflacarray._PinnedPool): power-of-two blocks, handed back when the
    last view of the array dies, re-used by the next request of the class, bounded idle and outstanding totals.  The
    allocator is replaced by malloc here (pinned memory needs a GPU)."""
    import gc

    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    calls = {"alloc": 0, "free": 0}

    class FakeLib:
        def fa_pinned_alloc(self, cap):
            calls["alloc"] += 1
            return libc.malloc(cap)

        def fa_pinned_free(self, addr):
            calls["free"] += 1
            libc.free(addr)

    monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
    monkeypatch.setenv("FLACARRAY_HIP_PINNED_POOL_MB", "1")
    monkeypatch.setenv("FLACARRAY_HIP_PINNED_MAX_MB", "2")
    pool = libflacarray._PinnedPool()
    for i in range(50):  # one class, one live array at a time: one allocation in all
        a = pool.empty(20000 + i, np.int32)
        a[:] = i
        view = a.reshape(1, -1)[:, 5:]
        del a
        assert int(view[0, 0]) == i  # (the block lives as long as any view of it)
        del view
    gc.collect()
    assert calls == {"alloc": 1, "free": 0} and pool._out == 0 and pool._idle == 131072
    held = [pool.empty(100000, np.float32) for _ in range(4)]  # 512 KB blocks: four fit the 2 MB that may be out
    assert all(h is not None for h in held) and pool.empty(100000, np.float32) is None
    del held
    gc.collect()
    assert pool._out == 0 and pool._idle <= (1 << 20) and calls["free"] >= 2  # (idle blocks beyond 1 MB are freed)
    assert pool.empty(0, np.int32) is None and pool.empty((64 << 20) // 4 + 1, np.int32) is None
    pool.drain()
    assert pool._idle == 0
