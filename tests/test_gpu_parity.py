"""GPU parity: the HIP path against the CPU oracle on the same seeded inputs (bit-exact)."""
import os

import numpy as np
import pytest

from tests.conftest import full_range_i32, sinusoid_noise_f32, sinusoid_noise_i32

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flacarray_amd

    return flacarray_amd


@pytest.fixture(autouse=True, params=["auto", "k7"])
def decoder_dispatch(request, monkeypatch):
    """Launches of up to 4096 frames go to the wave-per-frame decoder K7L, larger ones to K7 (lane per frame, the
    headline kernel); arrays of fewer than 1024 / 4096 frames are encoded by the placing encoder K3G, larger ones of the
    headline geometry by K3F.  The cases of this file are small, so left alone they would all exercise K7L and K3G: every
    test runs twice, once with the library's own dispatch and once with K7L switched off and K3F taking every array of
    its geometry (the variables are read per call)."""
    if request.param == "k7":
        monkeypatch.setenv("FLACARRAY_HIP_LATENCY", "0")
        monkeypatch.setenv("FLACARRAY_HIP_PLACED_BELOW", "0")
    else:
        monkeypatch.delenv("FLACARRAY_HIP_LATENCY", raising=False)
        monkeypatch.delenv("FLACARRAY_HIP_PLACED_BELOW", raising=False)
    return request.param


def _frame_diff(oracle, x, level, info_gpu, nf):
    """Describe the first frame whose decisions differ (for assertion messages)."""
    msgs = []
    keys = ["type", "order", "porder", "wasted", "shift", "precision", "nbytes", "blocksize"]
    for s in range(x.shape[0]):
        oi = oracle.stream_info(x[s], level)
        for f in range(nf):
            g = info_gpu[s * nf + f]
            o = [oi[f][k] for k in keys]
            if list(g) != o:
                msgs.append(f"stream {s} frame {f}: gpu {dict(zip(keys, g))} oracle {dict(zip(keys, o))}")
                if len(msgs) >= 5:
                    return "\n".join(msgs)
    return "\n".join(msgs)


def _encode_device(fa, x, level):
    import torch

    d = torch.from_numpy(x).cuda()
    comp, st, nb, info = fa.encode_flac_device(d, level=level, return_info=True)
    torch.cuda.synchronize()
    return comp.cpu().numpy(), st.cpu().numpy(), nb.cpu().numpy(), info.cpu().numpy()


CASES = [
    ("sinus16x65536", lambda: sinusoid_noise_i32(16, 65536)),
    ("sinus3x10000", lambda: sinusoid_noise_i32(3, 10000, seed=7)),
    ("fullrange3x10000", lambda: full_range_i32((3, 10000))),
    ("small_amp", lambda: sinusoid_noise_i32(4, 12288, seed=11, amp=8)),
    ("wasted_bits", lambda: (sinusoid_noise_i32(4, 9000, seed=5, amp=64) * 16).astype(np.int32)),
    ("zeros", lambda: np.zeros((2, 5000), dtype=np.int32)),
    ("const", lambda: np.full((2, 4097), -77777, dtype=np.int32)),
    ("ramp", lambda: (np.arange(3 * 8192, dtype=np.int64).reshape(3, 8192) * 3 - 999).astype(np.int32)),
    ("spikes", lambda: _spikes()),
    # whole frames only (the single-pass kernel's geometry): verbatim, wasted bits, constant, mixed frame types
    ("fullrange2x8192", lambda: full_range_i32((2, 8192))),
    ("wasted_whole_frames", lambda: (sinusoid_noise_i32(3, 8192, seed=6, amp=64) * 32).astype(np.int32)),
    ("zeros_whole_frames", lambda: np.zeros((2, 8192), dtype=np.int32)),
    ("const_then_noise", lambda: np.concatenate([np.full((3, 4096), 12345, np.int32), sinusoid_noise_i32(3, 8192, seed=2)], axis=1)),
    ("walk_whole_frames", lambda: np.cumsum(np.random.default_rng(3).integers(-2000, 2001, size=(5, 16384)), axis=1).astype(np.int32)),
    ("wide_5x12288", lambda: sinusoid_noise_i32(5, 12288, seed=13, amp=2**27)),
]


def _spikes():
    rng = np.random.default_rng(99)
    x = rng.integers(-3, 4, size=(4, 8192)).astype(np.int32)
    x[0, 100] = 2**31 - 1
    x[1, 5000] = -(2**31)
    x[2, ::512] = 2**28
    return x


@pytest.mark.parametrize("name,gen", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("level", [0, 3, 5, 8])
def test_encode_bytes_match_oracle(fa, oracle, name, gen, level):
    x = np.ascontiguousarray(gen())
    blob_o, st_o, nb_o = oracle.encode_i32(x, level)
    blob_g, st_g, nb_g, info = _encode_device(fa, x, level)
    bs = 1152 if level <= 2 else 4096
    nf = (x.shape[1] + bs - 1) // bs
    if not (np.array_equal(nb_g.reshape(-1), nb_o) and np.array_equal(blob_g, blob_o)):
        pytest.fail("compressed bytes differ from the oracle\n" + _frame_diff(oracle, x, level, info, nf))
    assert np.array_equal(st_g.reshape(-1), st_o)


@pytest.mark.parametrize("n", [1, 2, 4, 5, 15, 16, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 10000])
def test_encode_lengths(fa, oracle, n):
    rng = np.random.default_rng(n)
    x = np.stack([rng.integers(-(2**17), 2**17, n), np.cumsum(rng.integers(-50, 51, n))]).astype(np.int32)
    for level in (2, 5):
        blob_o, st_o, nb_o = oracle.encode_i32(x, level)
        blob_g, st_g, nb_g, info = _encode_device(fa, x, level)
        bs = 1152 if level <= 2 else 4096
        nf = (n + bs - 1) // bs
        if not np.array_equal(blob_g, blob_o):
            pytest.fail(f"n={n} level={level}\n" + _frame_diff(oracle, x, level, info, nf))
        assert np.array_equal(nb_g.reshape(-1), nb_o)


def test_decode_oracle_streams(fa, oracle):
    """Streams written by the oracle encoder decode bit-exactly on the GPU (full + slices)."""
    import torch

    for level in (0, 5, 8):
        for x in (sinusoid_noise_i32(8, 20000, seed=3), full_range_i32((3, 10000)), _spikes()):
            blob, st, nb = oracle.encode_i32(x, level)
            d = fa.decode_flac_device(torch.from_numpy(blob).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda(), x.shape[1])
            assert np.array_equal(d.cpu().numpy(), x)
            n = x.shape[1]
            for first, last in ((0, 1), (n // 2 - 5, n // 2 + 5), (4090, min(8200, n)), (n - 1, n), (1, n)):
                d = fa.decode_flac_device(
                    torch.from_numpy(blob).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda(), n, first, last
                )
                assert np.array_equal(d.cpu().numpy(), x[:, first:last]), (level, first, last)


def test_host_abi_roundtrip(fa, oracle):
    """numpy in / numpy out through the reference-shaped C ABI (encode_i32 / decode_i32)."""
    x = sinusoid_noise_i32(12, 30000, seed=21).reshape(4, 3, 30000)
    comp, starts, nbytes = fa.encode_flac(x, 5, use_threads=True)
    assert starts.shape == (4, 3) and nbytes.shape == (4, 3) and comp.dtype == np.uint8
    blob_o, st_o, nb_o = oracle.encode_i32(x.reshape(12, -1), 5)
    assert np.array_equal(comp, blob_o) and np.array_equal(starts.reshape(-1), st_o)
    y = fa.decode_flac(comp, starts, nbytes, 30000)
    assert y.shape == x.shape and np.array_equal(y, x)
    y = fa.decode_flac(comp, starts, nbytes, 30000, first_sample=14995, last_sample=15005)
    assert np.array_equal(y, x[..., 14995:15005])
    # oracle decodes GPU bytes
    assert np.array_equal(oracle.decode_i32(comp, starts, nbytes, 30000), x.reshape(12, -1))


def test_float_quantise_matches_oracle(fa, oracle):
    import torch

    x = sinusoid_noise_f32(9, 50001, seed=17)
    x[3] = 0.0  # all-zero stream: gain 1, offset +-0
    x[4] += 10.51
    q = (2.0**-16 * (1 + np.arange(9) % 4)).astype(np.float32)
    for quanta in (None, q):
        io, offo, go = oracle.float32_to_int32(x, quanta)
        ig, offg, gg = fa.float32_to_int32_device(torch.from_numpy(x).cuda(), None if quanta is None else torch.from_numpy(quanta))
        assert np.array_equal(ig.cpu().numpy(), io)
        assert np.array_equal(offg.cpu().numpy().view(np.uint32), offo.view(np.uint32))
        assert np.array_equal(gg.cpu().numpy().view(np.uint32), go.view(np.uint32))
        fo = oracle.int32_to_float32(io, offo, go)
        fg = fa.int_to_float(io, offo, go)
        assert np.array_equal(fg.view(np.uint32), fo.view(np.uint32))
    # fused decode + dequantise == separate steps
    io, offo, go = oracle.float32_to_int32(x, q)
    blob, st, nb = oracle.encode_i32(io, 5)
    d = fa.decode_flac_device(
        torch.from_numpy(blob).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda(), x.shape[1],
        offsets=torch.from_numpy(offo), gains=torch.from_numpy(go),
    )
    assert np.array_equal(d.cpu().numpy().view(np.uint32), oracle.int32_to_float32(io, offo, go).view(np.uint32))
    # |x^ - x| <= quanta/2 (+ ulps of |x|, |offset| from the float32 subtract/add), as tests/array.py:251-260
    tol = 0.5 * q[:, None] + 4 * np.finfo(np.float32).eps * (np.abs(x) + np.abs(offo[:, None]))
    assert np.all(np.abs(d.cpu().numpy() - x) <= tol)


def test_scattered_slices(fa, oracle):
    import torch

    x = sinusoid_noise_i32(32, 40000, seed=4)
    blob, st, nb = oracle.encode_i32(x, 5)
    rng = np.random.default_rng(987654321)
    n = 500
    ch = rng.integers(0, 32, n)
    cnt = rng.integers(1, 8193, n)
    first = np.array([rng.integers(0, 40000 - c + 1) for c in cnt])
    first[0], cnt[0] = 40000 - 1, 1
    out, off = fa.decode_slices_device(torch.from_numpy(blob).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda(), 40000, ch, first, cnt)
    out = out.cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[off[i] : off[i] + cnt[i]], x[ch[i], first[i] : first[i] + cnt[i]]), i


GOLDEN_NAMES = ["g1_const", "g2_verbatim", "g3_fixed", "g4_lpc", "g5_wasted", "g6_16bit", "g7_deep"]


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_decode_hand_assembled_streams(fa, name):
    """Streams assembled field by field from RFC 9639 (tests/golden/make_golden.py): no SEEKTABLE
    (frame walk), 8/16-bit blocksize codes, escapes, Rice2, wasted bits, 16/24-bit samples,
    predictor orders 12 and 20 (deeper-history passes), metadata blocks to skip."""
    import os

    import torch

    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flac_vectors.npz"))
    s, st, n = v[name + "_samples"], v[name + "_stream"], int(v[name + "_size"])
    # three copies in one blob, addressed out of order, the blob deliberately not 16-byte aligned
    blob = np.concatenate([np.zeros(3, np.uint8), st, st, st])
    starts = np.array([3 + 2 * st.size, 3, 3 + st.size], dtype=np.int64)
    nbytes = np.full(3, st.size, dtype=np.int64)
    d = fa.decode_flac_device(torch.from_numpy(blob).cuda()[3:], torch.from_numpy(starts - 3).cuda(), torch.from_numpy(nbytes).cuda(), n)
    assert np.array_equal(d.cpu().numpy(), np.stack([s, s, s]))
    y = fa.decode_flac(blob, starts, nbytes, n)  # host C ABI
    assert np.array_equal(y, np.stack([s, s, s]))
    for first, last in ((0, 1), (n // 2, n), (max(n - 3, 0), n)):
        if first < last:
            y = fa.decode_flac(blob, starts, nbytes, n, first_sample=first, last_sample=last)
            assert np.array_equal(y, np.stack([s[first:last]] * 3)), (first, last)


def test_corrupt_stream_reports_error(fa, oracle):
    x = sinusoid_noise_i32(2, 9000, seed=1)
    blob, st, nb = oracle.encode_i32(x, 5)
    bad = blob.copy()
    bad[int(st[1]) + 0] ^= 0xFF  # break the "fLaC" marker of stream 1
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac(bad, st, nb, 9000)
    bad = blob.copy()
    bad[int(st[0]) + 100 + 2] ^= 0x10  # frame header byte of stream 0's first frame: CRC-8 mismatch
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac(bad, st, nb, 9000)


def test_frame_crc16_verification(fa, oracle):
    """A flipped sample bit passes every header check: without the integrity pass the decoder returns (wrong)
    samples with code 0, as DESIGN.md says; with fa.set_decode_verify(True) it is ERROR_DECODE_PROCESS.  Intact
    streams -- own, oracle-written and without a SEEKTABLE -- pass the check, whole and in slices."""
    import torch

    from tests.conftest import strip_seektable

    x = full_range_i32((3, 20000), seed=41)  # VERBATIM frames: a flipped sample bit cannot desynchronise the parse
    blob, st, nb = oracle.encode_i32(x, 5)
    nf = 5
    hb = 4 + 4 + 34 + 4 + 18 * nf
    bad = blob.copy()
    bad[int(st[1]) + hb + 3000] ^= 0x04  # inside frame 0 of stream 1, far from any header
    d = lambda b: (torch.from_numpy(b).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda())  # noqa: E731
    assert fa.set_decode_verify(False) is False
    y = fa.decode_flac_device(*d(bad), 20000).cpu().numpy()
    assert np.array_equal(y[0], x[0]) and np.array_equal(y[2], x[2]) and not np.array_equal(y[1], x[1])
    try:
        assert fa.set_decode_verify(True) is False
        with pytest.raises(RuntimeError, match="Decoding failed"):
            fa.decode_flac_device(*d(bad), 20000)
        with pytest.raises(RuntimeError, match="Decoding failed"):
            fa.decode_flac_device(*d(bad), 20000, 100, 200)  # the damaged frame is the one the slice reads
        assert np.array_equal(fa.decode_flac_device(*d(bad), 20000, 9000, 9500).cpu().numpy(), x[:, 9000:9500])  # others are untouched
        assert np.array_equal(fa.decode_flac_device(*d(blob), 20000).cpu().numpy(), x)
        b2, s2, n2 = strip_seektable(blob, st, nb)
        assert np.array_equal(fa.decode_flac_device(torch.from_numpy(b2).cuda(), torch.from_numpy(s2).cuda(), torch.from_numpy(n2).cuda(), 20000).cpu().numpy(), x)
        xg = sinusoid_noise_i32(4, 8192, seed=42)
        cg, sg, ng = fa.encode_flac_device(torch.from_numpy(xg).cuda())  # single-pass encoder's own CRCs
        assert np.array_equal(fa.decode_flac_device(cg, sg, ng, 8192).cpu().numpy(), xg)
        assert np.array_equal(fa.decode_flac(bad, st, nb, 20000, first_sample=9000, last_sample=9500), x[:, 9000:9500])  # host ABI
        with pytest.raises(RuntimeError, match="Decoding failed"):
            fa.decode_flac(bad, st, nb, 20000)
    finally:
        assert fa.set_decode_verify(False) is True
    # per call: the argument wins over the process default (now off); the host-pointer ABI always checks, as libFLAC does
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac_device(*d(bad), 20000, verify=True)
    assert not np.array_equal(fa.decode_flac_device(*d(bad), 20000, verify=False).cpu().numpy(), x)
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac(bad, st, nb, 20000)
    ix = fa.DeviceDecodeIndex(*d(bad), 20000)
    try:
        with pytest.raises(RuntimeError, match="Decoding failed"):
            ix.decode(verify=True)
        assert np.array_equal(ix.decode(9000, 9500, verify=True).cpu().numpy(), x[:, 9000:9500])
        out, _ = ix.decode_slices([0, 2], [0, 10], [50, 60], verify=True)  # streams 0 and 2 are intact
        assert np.array_equal(out.cpu().numpy(), np.concatenate([x[0, :50], x[2, 10:70]]))
        with pytest.raises(RuntimeError, match="Decoding failed"):
            ix.decode_slices([1], [0], [50], verify=True)
    finally:
        ix.close()


def test_float_array_path(fa, oracle):
    """array_compress / array_decompress on float32 (quanta scalar, per-stream array, precision)."""
    x = sinusoid_noise_f32(6, 30000, seed=12).reshape(2, 3, 30000)
    for kw in ({"quanta": 1e-4}, {"quanta": np.full((2, 3), 2e-4, np.float32)}, {"precision": 4}):
        comp, starts, nbytes, off, gain = fa.array_compress(x, level=5, **kw)
        y = fa.array_decompress(comp, 30000, starts, nbytes, stream_offsets=off, stream_gains=gain)
        q = 1.0 / gain
        assert np.all(np.abs(y - x) <= 0.5 * q[..., None] + 8 * np.finfo(np.float32).eps * (np.abs(x) + np.abs(off[..., None])))
    with pytest.raises(RuntimeError, match="NaNs"):
        bad = x.copy()
        bad[1, 1, 77] = np.nan
        fa.array_compress(bad, quanta=1e-4)


def test_read_slices_batched(fa):
    x = sinusoid_noise_i32(24, 20000, seed=31).reshape(4, 6, 20000)
    f = fa.FlacArray.from_array(x)
    rng = np.random.default_rng(2)
    streams = rng.integers(0, 24, 64)
    cnt = rng.integers(1, 9000, 64)
    first = np.array([rng.integers(0, 20000 - c + 1) for c in cnt])
    outs = f.read_slices(streams, first, cnt)
    flat = x.reshape(24, -1)
    for i in range(64):
        assert np.array_equal(outs[i], flat[streams[i], first[i] : first[i] + cnt[i]])


def test_full_size_properties(fa):
    """BASELINE.json's configuration 2 at its stated size -- 4096 channels x 2^20 int32: 16 GiB of input,
    17.4 GB of frame slots and a 9.8 GB blob, so slot and blob byte offsets pass 2^32 -- through
    size-independent properties: decode(encode(x)) == x, nbytes/starts are a
    consistent exclusive scan, every stream is self-contained (decoding a shuffled subset through
    its own starts/nbytes gives the same rows), slices equal the corresponding part of the full
    decode, and a 64-bit checksum of the decoded matrix equals the input's."""
    import torch

    import bench

    dev = torch.device("cuda", 0)
    n_ch, n = 4096, 1 << 20
    x = bench.make_data(torch, n_ch, n, 2024, dev)
    comp, st, nb = fa.encode_flac_device(x, level=5)
    assert comp.numel() > 2**32 and int(st[-1]) > 2**32  # the offsets this test exists for
    assert int(st[0]) == 0 and torch.equal(torch.cumsum(nb, 0) - nb, st) and int(st[-1] + nb[-1]) == comp.numel()
    y = fa.decode_flac_device(comp, st, nb, n)
    assert torch.equal(y, x)
    assert int(y.to(torch.int64).sum()) == int(x.to(torch.int64).sum())
    sel = torch.randperm(n_ch, device=dev)[:37]
    ys = fa.decode_flac_device(comp, st[sel].contiguous(), nb[sel].contiguous(), n, 500000, 500000 + 70001)
    assert torch.equal(ys, x[sel, 500000 : 500000 + 70001])
    ratio = comp.numel() / (4.0 * x.numel())
    assert 0.5 < ratio < 0.65  # c/4 of the sinusoid+noise workload (SURVEY.md 8d expects 0.55-0.6)
    # every stream is a complete FLAC stream of its own: marker, and the oracle decodes a few of them
    # (first, last, one past the 2^32-byte mark) from nothing but their own bytes
    from oracle import oracle as O

    past = int(torch.searchsorted(st, torch.tensor(2**32, device=dev)))
    for s_i in (0, past, n_ch - 1):
        b0, b1 = int(st[s_i]), int(st[s_i] + nb[s_i])
        one = comp[b0:b1].cpu().numpy()
        assert bytes(one[:4]) == b"fLaC"
        yo = O.decode_i32(one, np.zeros(1, np.int64), np.array([b1 - b0], np.int64), n)
        assert np.array_equal(yo.reshape(-1), x[s_i].cpu().numpy())
    del y, ys, comp
    torch.cuda.empty_cache()


def _make_float_field(torch, n_ch, n_samp, seed, dev):
    """S3 of SURVEY.md 8(d): the sinusoid+noise field before rint, float32, amplitude 1, generated on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n_ch, n_samp), dtype=torch.float32, device=dev)
    t = torch.arange(n_samp, device=dev, dtype=torch.float32)
    f = 5.0 / n_samp
    wave = 2.0 * torch.sin(2 * np.pi * 3 * f * t) + 6.0 * torch.sin(2 * np.pi * f * t)
    for c0 in range(0, n_ch, 64):
        c1 = min(n_ch, c0 + 64)
        dc = 5.0 * (torch.rand((c1 - c0, 1), generator=g, device=dev) - 0.5)
        sc = torch.rand((c1 - c0, 1), generator=g, device=dev)
        out[c0:c1] = dc + sc * wave + torch.randn((c1 - c0, n_samp), generator=g, device=dev)
    return out


def test_cfg3_full_length_float(fa, oracle):
    """BASELINE.json's configuration 3 at full stream length: 1024 channels x 2^20 float32 with per-channel
    quanta 2^-16 (1 + c mod 4): quantise -> encode -> decode with the fused restore.
    Properties: |x^ - x| <= quanta/2 (+ float32 rounding of the subtract / add, as the reference's own
    bound, tests/array.py:251-260); gains == 1/quanta; offsets are whole multiples of the quanta; and the
    first / last channel's integers, offset and gain are bit-equal to the oracle's (utils.c:160-243)."""
    import torch

    dev = torch.device("cuda", 0)
    n_ch, n = 1024, 1 << 20
    x = _make_float_field(torch, n_ch, n, 31337, dev)
    q = (2.0**-16 * (1 + torch.arange(n_ch, device=dev) % 4)).to(torch.float32)
    ints, off2, gain2 = fa.float32_to_int32_device(x, q)
    comp, st, nb, off, gain = fa.encode_flac_device_f32(x, q, level=5)  # quantisation fused into the encoder's load
    assert torch.equal(off, off2) and torch.equal(gain, gain2)
    y = fa.decode_flac_device(comp, st, nb, n, offsets=off, gains=gain)
    assert y.dtype == torch.float32 and y.shape == x.shape
    eps = float(np.finfo(np.float32).eps)
    tol = 0.5 * q[:, None] + 4 * eps * (x.abs() + off.abs()[:, None])
    assert bool(((y - x).abs() <= tol).all())
    assert torch.equal(gain, (1.0 / q.double()).float())
    ratio = off.double() / q.double()
    assert bool((ratio == ratio.round()).all())
    # the decoded integers are the quantised ones (the int path under the float path is lossless)
    assert torch.equal(fa.decode_flac_device(comp, st, nb, n), ints)
    for c in (0, n_ch - 1):
        io, offo, go = oracle.float32_to_int32(x[c : c + 1].cpu().numpy(), q[c : c + 1].cpu().numpy())
        assert np.array_equal(ints[c].cpu().numpy(), io.reshape(-1))
        assert offo.view(np.uint32)[0] == off[c : c + 1].cpu().numpy().view(np.uint32)[0]
        assert go.view(np.uint32)[0] == gain[c : c + 1].cpu().numpy().view(np.uint32)[0]


def test_float_input_fused_into_the_encoder(fa, oracle):
    """encode_flac_device_f32 (range pre-pass + quantisation in the single-pass encoder's staging load) gives the
    bytes, offsets and gains of the two separate steps -- which are bit-equal to the oracle's float32_to_int32
    (utils.c:160-243) followed by its encoder -- for explicit per-stream quanta and for quanta derived from the data."""
    import torch

    x = sinusoid_noise_f32(7, 3 * 4096, seed=23)
    x[2] = 0.0          # all-zero stream: gain 1
    x[3] += 10.51
    x[5, :4096] = 3.25  # a constant frame inside a stream
    q = (2.0**-16 * (1 + np.arange(7) % 4)).astype(np.float32)
    d = torch.from_numpy(x).cuda()
    for quanta in (q, None):
        for level in (3, 5, 8):
            io, offo, go = oracle.float32_to_int32(x, quanta)
            blob_o, st_o, nb_o = oracle.encode_i32(io, level)
            comp, st, nb, off, gain = fa.encode_flac_device_f32(d, None if quanta is None else torch.from_numpy(quanta), level=level)
            assert np.array_equal(off.cpu().numpy().view(np.uint32), offo.view(np.uint32))
            assert np.array_equal(gain.cpu().numpy().view(np.uint32), go.view(np.uint32))
            assert np.array_equal(comp.cpu().numpy(), blob_o) and np.array_equal(st.cpu().numpy(), st_o) and np.array_equal(nb.cpu().numpy(), nb_o)
            y = fa.decode_flac_device(comp, st, nb, x.shape[1], offsets=off, gains=gain)
            assert np.array_equal(y.cpu().numpy().view(np.uint32), oracle.int32_to_float32(io, offo, go).view(np.uint32))
    # a geometry the single-pass kernel does not take (tail frame): the two-step route, same contract
    xs = x[:, :10000].copy()
    io, offo, go = oracle.float32_to_int32(xs, q)
    comp, st, nb, off, gain = fa.encode_flac_device_f32(torch.from_numpy(xs).cuda(), torch.from_numpy(q))
    assert np.array_equal(comp.cpu().numpy(), oracle.encode_i32(io, 5)[0]) and np.array_equal(off.cpu().numpy().view(np.uint32), offo.view(np.uint32))
    bad = x.copy()
    bad[1, 5000] = np.nan
    with pytest.raises(RuntimeError, match="NaNs"):
        fa.encode_flac_device_f32(torch.from_numpy(bad).cuda(), torch.from_numpy(q))


def test_single_pass_and_slot_sequence_write_the_same_bytes(fa):
    """The single-pass kernel (K3F) and the slot sequence (K3 + K4 + K5, forced with FLACARRAY_HIP_SLOTS) are two
    implementations of one specification: same blob, starts and nbytes on frames of every kind."""
    import torch

    x = np.concatenate(
        [sinusoid_noise_i32(6, 8192, seed=71), full_range_i32((6, 4096)), np.zeros((6, 4096), np.int32),
         (sinusoid_noise_i32(6, 4096, seed=72, amp=64) * 8).astype(np.int32)], axis=1)
    for xx, level in [(x, 3), (x, 5), (x, 7), (x[:, :20000 + 4], 5), (x[:, : 2 * 4096 + 4], 8), (x[:, :12288 + 4092], 4)]:
        # (the shortened arrays end in a short last frame: slot encoder + scanner placement + compaction of the tails)
        x_ = np.ascontiguousarray(xx)
        d = torch.from_numpy(x_).cuda()
        a = fa.encode_flac_device(d, level=level)
        os.environ["FLACARRAY_HIP_SLOTS"] = "1"
        try:
            b = fa.encode_flac_device(d, level=level)
        finally:
            del os.environ["FLACARRAY_HIP_SLOTS"]
        assert a[0].untyped_storage().size() > a[0].numel() == b[0].numel() == b[0].untyped_storage().size()  # a view of the capacity buffer vs an exact tensor
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        assert torch.equal(fa.decode_flac_device(*a, x_.shape[1]).cpu(), torch.from_numpy(x_))


def test_placing_encoder_and_slot_sequence_write_the_same_bytes(fa, oracle):
    """K3G (K3's frame body in a ticket loop, frames moved into place by the waves that packed them) against the slot
    sequence and the oracle on the geometries K3F does not take -- the reference's own test shapes ((12, 1000)-class
    arrays and single 10 000-sample streams, tests/bindings.py:165-230), levels 0-2 (1152-sample blocks), lengths that
    are not a multiple of 4, rows that are not 16-byte aligned, int64 (two-channel frames) -- and on more frames than the
    persistent grid has workgroups, so that every slot is used many times."""
    import torch

    L = fa._lib.lib()
    rng = np.random.default_rng(4242)
    base = np.concatenate(
        [sinusoid_noise_i32(12, 5000, seed=81), full_range_i32((12, 1500)), np.zeros((12, 1200), np.int32),
         (sinusoid_noise_i32(12, 2304, seed=82, amp=64) * 8).astype(np.int32)], axis=1)  # (12, 10004)
    wide = base.astype(np.int64) * 8192 + rng.integers(-4096, 4096, base.shape)
    small64 = rng.integers(-100, 100, (5, 9001)).astype(np.int64)  # (side + right frames)
    many = sinusoid_noise_i32(2600, 1152 * 2 + 7, seed=83)  # level 1: 7800 frames > 2048 workgroups
    cases = [(base[:, :1000], 5), (base[:1, :10000], 5), (base[:, :1000], 0), (base, 1), (base, 2), (base[:, :4097], 3), (base[:, :8191], 8),
             (base[:, :10001], 6), (base[:, :4096], 5), (wide[:, :1000], 5), (wide[:1, :10000], 5), (wide, 0), (wide, 3), (wide[:, :8192], 7),
             (small64, 5), (many, 1), (many[:, :1000].astype(np.int64) << 20, 2)]
    for xx, level in cases:
        x_ = np.ascontiguousarray(xx)
        d = torch.from_numpy(x_).cuda()
        assert L.fa_encode_single_pass_supported(int(np.prod(x_.shape[:-1])), x_.shape[-1], level) == 1
        a = fa.encode_flac_device(d, level=level)
        os.environ["FLACARRAY_HIP_SLOTS"] = "1"
        try:
            assert L.fa_encode_single_pass_supported(int(np.prod(x_.shape[:-1])), x_.shape[-1], level) == 0
            b = fa.encode_flac_device(d, level=level)
        finally:
            del os.environ["FLACARRAY_HIP_SLOTS"]
        assert a[0].untyped_storage().size() > a[0].numel() == b[0].numel() == b[0].untyped_storage().size()
        assert all(torch.equal(u, v) for u, v in zip(a, b)), (x_.shape, x_.dtype, level)
        blob, st, nb = (oracle.encode_i64 if x_.dtype == np.int64 else oracle.encode_i32)(x_, level)
        assert np.array_equal(a[0].cpu().numpy(), blob) and np.array_equal(a[1].cpu().numpy(), st) and np.array_equal(a[2].cpu().numpy(), nb)
        assert torch.equal(fa.decode_flac_device(*a, x_.shape[1], is_int64=x_.dtype == np.int64).cpu(), torch.from_numpy(x_))
    # the persistent grid is a performance choice, not a correctness one: 64 workgroups (far fewer than frames or than the
    # chip holds) and 16 384 (most of them start when the tickets are gone, or find the chip full) write the same bytes
    d = torch.from_numpy(many).cuda()
    ref = fa.encode_flac_device(d, level=1)
    for grid in ("64", "16384"):
        os.environ["FLACARRAY_HIP_PLACED_GRID"] = grid
        try:
            got = fa.encode_flac_device(d, level=1)
        finally:
            del os.environ["FLACARRAY_HIP_PLACED_GRID"]
        assert all(torch.equal(u, v) for u, v in zip(ref, got)), grid
    # rows that start 4 bytes off a 16-byte boundary (a view into a larger tensor): K3G instead of K3F, same bytes
    big = torch.from_numpy(np.concatenate([np.zeros(1, np.int32), sinusoid_noise_i32(3, 8192, seed=84).reshape(-1)])).cuda()
    view = big[1:].reshape(3, 8192)
    assert view.data_ptr() % 16 == 4
    a = fa.encode_flac_device(view, level=5)
    b = fa.encode_flac_device(view.clone(), level=5)  # (aligned: K3F)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    # an undersized buffer is refused cleanly, as for K3F
    with pytest.raises(RuntimeError, match="return code = 1"):
        fa.encode_flac_device(torch.from_numpy(np.ascontiguousarray(full_range_i32((12, 1500)))).cuda(), level=1, capacity_bytes=12 * 100 + 4096)


def test_corrupt_index_rejected(fa, oracle):
    """A damaged index (stream_starts / stream_nbytes pointing outside the blob, negative entries) must come
    back as ERROR_DECODE_INIT, never as an out-of-range device read."""
    import torch

    x = sinusoid_noise_i32(3, 9000, seed=8)
    blob, st, nb = oracle.encode_i32(x, 5)
    d_blob = torch.from_numpy(blob).cuda()
    for bad_st, bad_nb in (
        (st + np.array([0, 10**12, 0]), nb),           # start far past the end
        (st, nb + np.array([0, 0, 10**9])),            # length runs off the end
        (st * np.array([1, -1, 1]) - np.array([0, 1, 0]), nb),  # negative start
        (st, nb * np.array([1, -1, 1])),               # negative length
        (np.full(3, blob.size, np.int64), nb),         # start == blob size
    ):
        with pytest.raises(RuntimeError, match="Decoding failed"):
            fa.decode_flac_device(d_blob, torch.from_numpy(bad_st.astype(np.int64)).cuda(), torch.from_numpy(bad_nb.astype(np.int64)).cuda(), 9000)
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac(blob, st * np.array([1, -1, 1]) - np.array([0, 1, 0]), nb, 9000)  # host ABI: negative start
    # the intact index still decodes
    assert np.array_equal(fa.decode_flac_device(d_blob, torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda(), 9000).cpu().numpy(), x)


def test_resident_flacarray(fa):
    """A FlacArray made resident with to_device() answers every read like the host-backed one (and like numpy),
    int32 and quantised float32, without re-uploading the store."""
    import torch

    x = sinusoid_noise_i32(24, 20000, seed=31).reshape(4, 6, 20000)
    xf = sinusoid_noise_f32(24, 20000, seed=32).reshape(4, 6, 20000)
    for arr, kw in ((x, {}), (xf, {"quanta": 1e-4})):
        host = fa.FlacArray.from_array(arr, **kw)
        ref = host.to_array()
        res = fa.FlacArray(host).to_device()
        assert res.is_resident and not host.is_resident
        keys = [
            (slice(None),), (1,), (1, 2), (1, 2, slice(5, 9000)), (slice(1, 3), slice(None), slice(100, 101)),
            (slice(None), 3, 19999), (slice(0, 4, 2), slice(5, 0, -2)), (3, 5, slice(4096, 8192)),
        ]
        for key in keys:
            a, b = host[key], res[key]
            assert a.shape == b.shape == ref[key].shape, key
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
        assert np.array_equal(res.to_array().view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(res.to_array(stream_slice=slice(7, 4500)).view(np.uint32), ref[..., 7:4500].view(np.uint32))
        keep = np.zeros((4, 6), bool)
        keep[0, 1] = keep[3, 5] = keep[2, 2] = True
        a, ia = host.to_array(keep=keep, keep_indices=True)
        b, ib = res.to_array(keep=keep, keep_indices=True)
        assert ia == ib and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        rng = np.random.default_rng(5)
        streams = rng.integers(0, 24, 200)
        cnt = rng.integers(1, 9000, 200)
        first = np.array([rng.integers(0, 20000 - c + 1) for c in cnt])
        flat = ref.reshape(24, -1)
        for o, s_i, f0, c in zip(res.read_slices(streams, first, cnt), streams, first, cnt):
            assert np.array_equal(o.view(np.uint32), flat[s_i, f0 : f0 + c].view(np.uint32))
    # built on the device, never leaves it: from_device_array
    d = torch.from_numpy(x.reshape(24, -1)).cuda()
    r2 = fa.FlacArray.from_device_array(d)
    assert r2.is_resident and r2 == fa.FlacArray.from_array(x.reshape(24, -1))
    assert np.array_equal(r2[5, 100:300], x.reshape(24, -1)[5, 100:300])


@pytest.mark.parametrize("kind", ["i32", "f32", "i64", "noseek"])
def test_decode_index_matches_plain_decode(fa, oracle, kind):
    """fa_decode_index_create / fa_decode_indexed: the index (headers parsed, frame offsets tabulated once) must
    answer whole decodes, sample ranges and slice batches exactly like the one-shot entry points, any number of
    times, and reject what they reject."""
    import torch

    from tests.conftest import strip_seektable

    n = 30011
    dev = torch.device("cuda", 0)
    off = gain = None
    if kind == "i64":
        rng = np.random.default_rng(3)
        x = (rng.integers(-(2**40), 2**40, (5, n)) + np.cumsum(rng.integers(-900, 900, (5, n)), axis=1)).astype(np.int64)
        blob, st, nb = fa.encode_flac(x, 5)
    else:
        x = np.concatenate([sinusoid_noise_i32(9, n, seed=15), full_range_i32((2, n), seed=16)])
        blob, st, nb = oracle.encode_i32(x, 5)
        if kind == "noseek":
            blob, st, nb = strip_seektable(blob, st, nb)
        if kind == "f32":
            off = torch.linspace(-3, 3, x.shape[0])
            gain = torch.linspace(1e-3, 2e-3, x.shape[0])
    tb, ts, tn = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (blob, st, nb))
    idx = fa.DeviceDecodeIndex(tb, ts, tn, n, is_int64=(kind == "i64"))
    plain = lambda f, l: fa.decode_flac_device(tb, ts, tn, n, f, l, offsets=off, gains=gain, is_int64=(kind == "i64"))  # noqa: E731
    for rep in range(2):
        for f, l in ((-1, -1), (0, n), (5, 6), (4095, 4097), (n - 1, n), (12345, 29000)):
            assert torch.equal(idx.decode(f, l, offsets=off, gains=gain), plain(f, l)), (rep, f, l)
        rng = np.random.default_rng(77 + rep)
        ss = rng.integers(0, x.shape[0], 300)
        cnt = rng.integers(1, 9000, 300)
        first = np.array([rng.integers(0, n - c + 1) for c in cnt])
        a, ao = idx.decode_slices(ss, first, cnt, offsets=off, gains=gain)
        b, bo = fa.decode_slices_device(tb, ts, tn, n, ss, first, cnt, offsets=off, gains=gain, is_int64=(kind == "i64"))
        assert np.array_equal(ao, bo) and torch.equal(a, b)
        if off is None:
            flat = a.cpu().numpy()
            for o, s_i, f0, c in zip(ao, ss, first, cnt):
                assert np.array_equal(flat[o : o + c], x[s_i, f0 : f0 + c])
    with pytest.raises(RuntimeError):
        idx.decode(10, n + 1)
    with pytest.raises(RuntimeError, match="Decoding failed"):
        idx.decode_slices(np.array([0, x.shape[0]]), np.array([0, 0]), np.array([5, 5]))
    with pytest.raises(RuntimeError, match="Decoding failed"):
        idx.decode_slices(np.array([0]), np.array([n - 3]), np.array([5]))
    assert torch.equal(idx.decode(offsets=off, gains=gain), plain(-1, -1))  # still usable after rejected calls
    idx.close()
    idx.close()
    # a damaged store is rejected when the index is built
    bad = tb.clone()
    bad[int(st[1]) : int(st[1]) + 4] = 0
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.DeviceDecodeIndex(bad, ts, tn, n, is_int64=(kind == "i64"))
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.DeviceDecodeIndex(tb, ts + 10**12, tn, n, is_int64=(kind == "i64"))


@pytest.mark.parametrize("name", ["example1", "example2", "example3"])
def test_rfc9639_worked_examples_on_the_gpu(fa, name):
    """The worked examples of RFC 9639 Appendix D (tests/golden/rfc9639.py: self-verifying through their CRC-8s,
    CRC-16s and MD5, checked in tests/test_oracle.py) through the HIP decoder and its CRC-16 verifier: VERBATIM with
    wasted bits, side/right stereo with FIXED subframes and a short last frame behind three metadata blocks, 8-bit LPC
    with an escaped partition -- streams this repository neither wrote nor assembled."""
    from tests.test_oracle import _rfc_samples

    data, x = _rfc_samples(name)
    n = x.shape[1]
    blob = np.frombuffer(data, dtype=np.uint8).copy()
    st, nb = np.array([0], np.int64), np.array([blob.size], np.int64)
    i64 = x.shape[0] == 2
    want = ((x[1] << 32) | (x[0] & 0xFFFFFFFF)) if i64 else x[0].astype(np.int32)
    for verify in (False, True):
        fa.set_decode_verify(verify)
        try:
            assert np.array_equal(fa.decode_flac(blob, st, nb, n, is_int64=i64).reshape(-1), want)
            if n > 2:
                assert np.array_equal(fa.decode_flac(blob, st, nb, n, 2, n - 1, is_int64=i64).reshape(-1), want[2 : n - 1])
            if verify:
                bad = blob.copy()
                bad[blob.size - 5] ^= 0x04  # a residual / sample bit of the last frame: only its CRC-16 can tell
                with pytest.raises(RuntimeError, match="Decoding failed"):
                    fa.decode_flac(bad, st, nb, n, is_int64=i64)
        finally:
            fa.set_decode_verify(False)


def test_streams_of_different_block_sizes_in_one_call(fa, oracle):
    """A store put together from encodes at level 0-2 (1152-sample blocks) and 3-8 (4096): libFLAC decodes every
    stream on its own terms (decompress.c:256-305); here the call falls back to one launch per block size."""
    import torch

    n = 20000
    xa, xb = sinusoid_noise_i32(3, n, seed=61), sinusoid_noise_i32(4, n, seed=62)
    ba, sa, na = oracle.encode_i32(xa, 1)
    bb, sb, nb = oracle.encode_i32(xb, 5)
    blob = np.concatenate([ba, bb])
    st = np.concatenate([sa, sb + ba.size])
    nbs = np.concatenate([na, nb])
    order = np.array([4, 0, 5, 1, 6, 2, 3])  # interleave the two kinds
    st, nbs = st[order].copy(), nbs[order].copy()
    x = np.concatenate([xa, xb])[order]
    assert np.array_equal(fa.decode_flac(blob, st, nbs, n), x)
    assert np.array_equal(fa.decode_flac(blob, st, nbs, n, 1000, 9000), x[:, 1000:9000])
    dev = torch.device("cuda", 0)
    tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (blob, st, nbs))
    assert np.array_equal(fa.decode_flac_device(tb, ts, tn, n).cpu().numpy(), x)
    off, gain = torch.linspace(-1, 1, 7), torch.linspace(1e-3, 2e-3, 7)
    got = fa.decode_flac_device(tb, ts, tn, n, 5, 4100, offsets=off, gains=gain)
    ref = torch.stack([fa.decode_flac_device(tb, ts[i : i + 1], tn[i : i + 1], n, 5, 4100, offsets=off[i : i + 1], gains=gain[i : i + 1])[0] for i in range(7)])
    assert torch.equal(got, ref)
    # a store that is damaged rather than mixed still fails
    bad = blob.copy()
    bad[int(st[2]) : int(st[2]) + 4] = 0
    with pytest.raises(RuntimeError, match="Decoding failed"):
        fa.decode_flac(bad, st, nbs, n)


@pytest.mark.parametrize("level,n", [(5, 100000), (0, 50001), (8, 4096 * 3)])
def test_decode_without_seektable(fa, oracle, level, n):
    """Streams without a SEEKTABLE (what libFLAC writes through the reference, compress.c:337-390)
    take the parallel sync-code scan; full decode and slices must equal the input."""
    import torch

    from tests.conftest import strip_seektable

    x = np.concatenate([sinusoid_noise_i32(19, n, seed=5), full_range_i32((3, n), seed=6)])
    blob, st, nb = oracle.encode_i32(x, level)
    b2, s2, n2 = strip_seektable(blob, st, nb)
    dev = torch.device("cuda", 0)
    tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (b2, s2, n2))
    y = fa.decode_flac_device(tb, ts, tn, n)
    assert np.array_equal(y.cpu().numpy(), x)
    lo, hi = n // 3, n // 3 + 5000
    ys = fa.decode_flac_device(tb, ts, tn, n, lo, hi)
    assert np.array_equal(ys.cpu().numpy(), x[:, lo:hi])
    # the oracle's general decoder agrees on the rewritten streams
    assert np.array_equal(oracle.decode_i32(b2, s2, n2, n), x)


def _crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


@pytest.mark.parametrize("fake_number", [3, 200])
def test_false_sync_inside_verbatim_frame(fa, oracle, fake_number):
    """A VERBATIM frame whose sample bytes spell a complete, CRC-8-valid frame header.  With a frame
    number below the frame count the false candidate collides with the true frame and the stream
    must fall back to the serial walk; above it, it must be ignored.  Either way the decode is exact."""
    import torch

    from tests.conftest import strip_seektable

    n = 8 * 4096
    x = full_range_i32((2, n), seed=77)  # incompressible: every frame is VERBATIM, samples byte aligned
    num = [fake_number] if fake_number < 0x80 else [0xC0 | (fake_number >> 6), 0x80 | (fake_number & 0x3F)]
    hdr = [0xFF, 0xF8, 0xC9, 0x0E] + num
    hdr.append(_crc8(hdr))
    hdr += [0] * (-len(hdr) % 4)
    words = np.frombuffer(bytes(hdr), dtype=">i4").astype(np.int32)
    x[1, 5 * 4096 + 100 : 5 * 4096 + 100 + len(words)] = words
    blob, st, nb = oracle.encode_i32(x, 5)
    info = oracle.stream_info(x[1], 5)
    assert info[5]["type"] == 1  # VERBATIM
    b2, s2, n2 = strip_seektable(blob, st, nb)
    assert bytes(hdr[:6]) in bytes(b2)
    dev = torch.device("cuda", 0)
    y = fa.decode_flac_device(torch.from_numpy(b2).to(dev), torch.from_numpy(s2).to(dev), torch.from_numpy(n2).to(dev), n)
    assert np.array_equal(y.cpu().numpy(), x)


def test_hdf5_write_read_array(fa):
    """hdf5.write_array / read_array and FlacArray.write_hdf5 / read_hdf5 on the version-1 layout
    (in-memory stand-in for h5py.Group: h5py is absent from this image)."""
    from flacarray_amd import hdf5 as H
    from tests.conftest import FakeH5Group

    x = sinusoid_noise_i32(6, 20000, seed=21).reshape(3, 2, 20000)
    g = FakeH5Group()
    H.write_array(x, g, level=5)
    assert np.array_equal(H.read_array(g), x)
    assert np.array_equal(H.read_array(g, stream_slice=slice(100, 9000)), x[..., 100:9000])
    keep = np.array([[True, False], [False, False], [True, True]])
    arr, idx = H.read_array(g, keep=keep, keep_indices=True)
    assert idx == [(0, 0), (2, 0), (2, 1)] and np.array_equal(arr, x[keep])

    xf = sinusoid_noise_f32(4, 10000, seed=22)
    gf = FakeH5Group()
    H.write_array(xf, gf, quanta=1e-4)
    assert gf["stream_offsets"].dtype == np.float32 and gf.attrs["flac_channels"] == "1"
    assert np.max(np.abs(H.read_array(gf) - xf)) <= 0.5e-4 + 1e-5

    fl = fa.FlacArray.from_array(x)
    g2 = FakeH5Group()
    fl.write_hdf5(g2)
    back = fa.FlacArray.read_hdf5(g2)
    assert back == fl and np.array_equal(back.to_array(), x)
    one = fa.FlacArray.from_array(x[0, 0])
    g3 = FakeH5Group()
    one.write_hdf5(g3)
    assert g3["stream_starts"].shape == (1,)
    assert fa.FlacArray.read_hdf5(g3).shape == (20000,)
    assert fa.FlacArray.read_hdf5(g3, no_flatten=True).shape == (1, 20000)


# ---- two-channel (int64 / float64) arrays ----------------------------------------------------
STEREO_GOLDEN = ["g8_stereo_lr", "g9_stereo_ls", "g10_stereo_sr", "g11_stereo_ms", "g12_stereo16", "g13_stereo_pred"]


def _i64_cases():
    rng = np.random.default_rng(64)
    n = 3 * 4096 + 777
    x = np.empty((6, n), dtype=np.int64)
    x[0] = np.cumsum(rng.integers(-(2**20), 2**20, n)) + 2**40           # counter above 2^32: the low word wraps
    x[1] = rng.integers(-(2**62), 2**62, n)                              # incompressible
    x[2] = -(2**63)                                                      # constant
    x[3] = (np.arange(n, dtype=np.int64) - 4500) * 1000003               # ramp
    x[4] = np.rint(2.0**45 * np.sin(np.arange(n) / 50.0)).astype(np.int64) + rng.integers(-(2**30), 2**30, n)
    x[5] = rng.integers(-5, 6, n) * 2**33                                # low word zero, wasted bits in the high word
    x[1, :4] = [2**63 - 1, -(2**63), 2**32, -1]
    return x


@pytest.mark.parametrize("name", STEREO_GOLDEN)
def test_decode_hand_assembled_stereo_streams(fa, name):
    """Two-channel streams assembled field by field (tests/golden/make_golden.py): every stereo channel
    assignment, 33-bit side channels (VERBATIM, FIXED and LPC), 16-bit stereo; no SEEKTABLE."""
    import torch

    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flac_vectors.npz"))
    s, st, n = v[name + "_samples"], v[name + "_stream"], int(v[name + "_size"])
    dev = torch.device("cuda", 0)
    blob = torch.from_numpy(np.concatenate([st, st])).to(dev)
    starts = torch.tensor([st.size, 0], dtype=torch.int64, device=dev)
    nbytes = torch.tensor([st.size, st.size], dtype=torch.int64, device=dev)
    y = fa.decode_flac_device(blob, starts, nbytes, n, is_int64=True).cpu().numpy()
    assert np.array_equal(y[0], s) and np.array_equal(y[1], s)
    lo, hi = n // 3, n - 1
    y = fa.decode_flac_device(blob, starts, nbytes, n, lo, hi, is_int64=True).cpu().numpy()
    assert np.array_equal(y[0], s[lo:hi])
    with pytest.raises(RuntimeError):  # not an int32 stream
        fa.decode_flac_device(blob, starts, nbytes, n)


@pytest.mark.parametrize("level", [0, 5, 8])
def test_decode_i64_oracle_streams(fa, oracle, level):
    import torch

    from tests.conftest import strip_seektable

    x = _i64_cases()
    n = x.shape[1]
    blob, st, nb = oracle.encode_i64(x, level)
    dev = torch.device("cuda", 0)
    tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (blob, st, nb))
    assert np.array_equal(fa.decode_flac_device(tb, ts, tn, n, is_int64=True).cpu().numpy(), x)
    assert np.array_equal(fa.decode_flac_device(tb, ts, tn, n, 4000, 9001, is_int64=True).cpu().numpy(), x[:, 4000:9001])
    # host ABI (decode_i64) through the reference-named wrapper
    assert np.array_equal(fa.decode_flac(blob, st, nb, n, is_int64=True), x)
    assert np.array_equal(fa.decode_flac(blob, st, nb, n, 10, 20, is_int64=True), x[:, 10:20])
    # without a seek table: sync-code scan on two-channel headers
    b2, s2, n2 = strip_seektable(blob, st, nb)
    t2 = [torch.from_numpy(a).to(dev) for a in (b2, s2, n2)]
    assert np.array_equal(fa.decode_flac_device(*t2, n, is_int64=True).cpu().numpy(), x)
    # float64 restore fused into the final pass
    off = np.linspace(-3, 3, x.shape[0])
    gain = np.full(x.shape[0], 2.0**20)
    yf = fa.decode_flac_device(tb, ts, tn, n, offsets=torch.from_numpy(off), gains=torch.from_numpy(gain), is_int64=True).cpu().numpy()
    assert np.array_equal(yf, oracle.int64_to_float64(x, off, gain))


@pytest.mark.parametrize("level", [0, 3, 5, 8])
def test_encode_i64_bytes_match_oracle(fa, oracle, level):
    """int64 arrays: the HIP encoder's two-channel streams are byte-identical to the oracle's."""
    import torch

    x = _i64_cases()
    n = x.shape[1]
    bo, so, no = oracle.encode_i64(x, level)
    comp, st, nb, info = fa.encode_flac_device(torch.from_numpy(x).cuda(), level=level, return_info=True)
    cg, sg, ng, info = comp.cpu().numpy(), st.cpu().numpy(), nb.cpu().numpy(), info.cpu().numpy()
    if not np.array_equal(cg, bo):
        keys = ["type", "order", "porder", "wasted", "shift", "precision", "nbytes", "blocksize"]
        nf2 = info.shape[0] // x.shape[0]
        for s_ in range(x.shape[0]):
            oi = oracle.stream_info_i64(x[s_], level)
            for k_ in range(nf2):
                g_ = list(info[s_ * nf2 + k_])
                o_ = [oi[k_][kk] for kk in keys]
                assert g_ == o_, f"stream {s_} subframe {k_}: gpu {dict(zip(keys, g_))} oracle {dict(zip(keys, o_))}"
    assert np.array_equal(sg, so) and np.array_equal(ng, no)
    assert np.array_equal(cg, bo)
    y = fa.decode_flac_device(comp, st, nb, n, is_int64=True)
    assert np.array_equal(y.cpu().numpy(), x)
    # host ABI (encode_i64 / decode_i64) through the reference-named wrappers
    c2, s2, n2 = fa.encode_flac(x, level)
    assert np.array_equal(np.asarray(c2), bo) and np.array_equal(s2, so)
    assert np.array_equal(fa.decode_flac(np.asarray(c2), s2, n2, n, is_int64=True), x)


@pytest.mark.parametrize("n", [1, 5, 16, 255, 4095, 4097, 10000])
def test_encode_i64_lengths(fa, oracle, n):
    import torch

    rng = np.random.default_rng(n)
    x = (np.cumsum(rng.integers(-(2**33), 2**33, (3, n)), axis=1)).astype(np.int64)
    bo, so, no = oracle.encode_i64(x, 5)
    comp, st, nb = fa.encode_flac_device(torch.from_numpy(x).cuda(), level=5)
    assert np.array_equal(comp.cpu().numpy(), bo)
    assert np.array_equal(fa.decode_flac_device(comp, st, nb, n, is_int64=True).cpu().numpy(), x)


def test_float64_quantise_matches_oracle(fa, oracle):
    """utils.c:245-348 on the GPU against the operation-by-operation CPU restatement."""
    rng = np.random.default_rng(8)
    x = rng.normal(0, 1, (5, 7001)) + np.array([[0.0], [0.5], [-10.51], [1e6], [-3e-7]])
    x[4] *= 1e-6
    for q in (None, np.array([1e-9, 2e-9, 1e-7, 1e-6, 1e-15])):
        io, oo, go = oracle.float64_to_int64(x, q)
        ig, og, gg = fa.float_to_int(x, quanta=q)
        assert ig.dtype == np.int64 and og.dtype == np.float64
        assert np.array_equal(ig, io) and np.array_equal(og, oo) and np.array_equal(gg, go)
        assert np.array_equal(fa.int_to_float(ig, og, gg), oracle.int64_to_float64(io, oo, go))
    with pytest.raises(RuntimeError, match="NaNs"):
        bad = x.copy()
        bad[2, 77] = np.nan
        fa.float_to_int(bad, quanta=1e-9)


def test_int64_float64_array_path(fa, oracle):
    """tests/array.py:26-146 recipe for the 64-bit dtypes: array_compress / array_decompress(_slice),
    FlacArray and the HDF5 layout (flac_channels = 2, float64 offsets / gains)."""
    from flacarray_amd import hdf5 as H
    from tests.conftest import FakeH5Group

    rng = np.random.default_rng(9)
    xi = (np.cumsum(rng.integers(-(2**34), 2**34, (2, 3, 9000)), axis=-1)).astype(np.int64)
    comp, st, nb, off, gain = fa.array_compress(xi, level=5)
    assert off is None and gain is None and st.shape == (2, 3)
    assert np.array_equal(fa.array_decompress(comp, 9000, st, nb, is_int64=True), xi)
    keep = np.zeros((2, 3), dtype=bool)
    keep[1, 0] = keep[0, 2] = True
    arr, idx = fa.array_decompress_slice(comp, 9000, st, nb, keep=keep, first_stream_sample=100, last_stream_sample=4200, is_int64=True)
    assert idx == [(0, 2), (1, 0)] and np.array_equal(arr, xi[keep][:, 100:4200])

    xf = rng.normal(0, 1, (4, 6000)) * 1e3 + 12345.678
    q = 1e-7
    comp, st, nb, off, gain = fa.array_compress(xf, quanta=q)
    assert off.dtype == np.float64 and gain.dtype == np.float64
    y = fa.array_decompress(comp, 6000, st, nb, stream_offsets=off, stream_gains=gain, is_int64=True)
    assert y.dtype == np.float64 and np.max(np.abs(y - xf)) <= 0.5 * q + 4 * np.finfo(np.float64).eps * 2e4

    fl = fa.FlacArray.from_array(xi)
    assert fl.dtype == np.int64 and np.array_equal(fl.to_array(), xi) and np.array_equal(fl[1, 2, 50:60], xi[1, 2, 50:60])
    ff = fa.FlacArray.from_array(xf, quanta=q)
    assert ff.dtype == np.float64 and np.max(np.abs(ff.to_array() - xf)) <= 0.5 * q + 1e-10
    g = FakeH5Group()
    ff.write_hdf5(g)
    assert g.attrs["flac_channels"] == "2" and g["stream_offsets"].dtype == np.float64
    back = fa.FlacArray.read_hdf5(g)
    assert back.dtype == np.float64 and np.array_equal(back.to_array(), ff.to_array())
    g2 = FakeH5Group()
    H.write_array(xi, g2)
    assert np.array_equal(H.read_array(g2), xi)


def test_read_slices_batched_int64(fa):
    rng = np.random.default_rng(17)
    x = (np.cumsum(rng.integers(-(2**35), 2**35, (5, 20000)), axis=-1)).astype(np.int64)
    fl = fa.FlacArray.from_array(x)
    streams = rng.integers(0, 5, 200)
    count = rng.integers(1, 6000, 200)
    first = np.array([rng.integers(0, 20000 - c + 1) for c in count])
    got = fl.read_slices(streams, first, count)
    for s_, f_, c_, g_ in zip(streams, first, count, got):
        assert g_.dtype == np.int64 and np.array_equal(g_, x[s_, f_ : f_ + c_])


@pytest.mark.parametrize("stereo", [False, True])
def test_corrupted_and_truncated_streams_terminate(fa, oracle, stereo):
    """Damaged input must come back quickly, as an error or as (wrong) samples: zeroed regions, random
    byte flips inside frames, streams cut short, with and without a seek table.  Every reader loop has
    an exit (a unary run longer than 2^20 bits kills the lane's frame)."""
    import time

    import torch

    from tests.conftest import strip_seektable

    rng = np.random.default_rng(123)
    n = 5 * 4096 + 100
    if stereo:
        x = (np.cumsum(rng.integers(-(2**33), 2**33, (3, n)), axis=1)).astype(np.int64)
        blob, st, nb = oracle.encode_i64(x, 5)
    else:
        x = sinusoid_noise_i32(3, n, seed=9)
        blob, st, nb = oracle.encode_i32(x, 5)
    dev = torch.device("cuda", 0)
    variants = []
    for seed in range(4):
        r = np.random.default_rng(seed)
        b = blob.copy()
        lo = int(st[1]) + 300
        for _ in range(20):
            b[r.integers(lo, int(st[1] + nb[1]) - 2)] ^= np.uint8(r.integers(1, 256))
        variants.append((b, st, nb))
    z = blob.copy()
    z[int(st[0]) + 500 : int(st[0]) + 9000] = 0  # a zeroed region: endless unary runs
    variants.append((z, st, nb))
    variants.append((blob, st, np.array([nb[0], nb[1] // 2, nb[2]], dtype=np.int64)))  # stream 1 cut short
    variants += [strip_seektable(*v) for v in variants[:2]] + [strip_seektable(z, st, nb)]
    t0 = time.perf_counter()
    for b, s_, n_ in variants:
        tb, ts, tn = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (b, s_, n_))
        try:
            y = fa.decode_flac_device(tb, ts, tn, n, is_int64=stereo).cpu().numpy()
            assert y.shape == x.shape
        except RuntimeError as e:
            assert "Decoding failed" in str(e)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 20.0
    # the intact streams still decode afterwards
    tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (blob, st, nb))
    assert np.array_equal(fa.decode_flac_device(tb, ts, tn, n, is_int64=stereo).cpu().numpy(), x)


def test_concurrent_host_calls(fa, oracle):
    """The reference's entry points are re-entrant (compress.c has no globals).  Here calls
    share cached device scratch, so the library serialises them: concurrent callers get correct results."""
    import threading

    xs = [sinusoid_noise_i32(3, 20000 + 1000 * i, seed=40 + i) for i in range(4)]
    want = [oracle.encode_i32(x, 5)[0] for x in xs]
    got, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                c, s, n = fa.encode_flac(xs[i], 5)
                y = fa.decode_flac(np.asarray(c), s, n, xs[i].shape[1])
                assert np.array_equal(y, xs[i])
            got[i] = np.asarray(c).copy()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def _reference_fake_data(shape, sigma, dtype, seed=123456789):
    """create_fake_data of the reference (demo.py:12-110) without MPI: uniform full-range integers with
    the extremes at positions 0 / 1 when sigma is None, else the sinusoid + noise field."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    if sigma is None:
        lo, hi = np.iinfo(dtype).min, np.iinfo(dtype).max
        flat = rng.integers(low=lo, high=hi, size=n, dtype=np.int64).astype(dtype)
        flat[0], flat[1] = lo, hi
        return flat.reshape(shape)
    lead = shape[:-1] + (1,)
    ss = shape[-1]
    t = np.arange(ss)
    minf = 5 / ss
    wave = np.zeros(ss, dtype=dtype)
    for freq, amp in zip([3 * minf, minf], [2 * sigma, 6 * sigma]):
        wave[:] += amp * np.sin(2 * np.pi * freq * t)
    scale = rng.random(size=lead)
    data = np.empty(shape, dtype=dtype)
    data[...] = scale * wave if len(shape) > 1 else (scale * wave).reshape(shape)
    data[...] += rng.normal(0.0, sigma, n).reshape(shape)
    return data


@pytest.mark.parametrize("shape", [(4, 3, 1000), (10000,)])
@pytest.mark.parametrize("dt,sigma,quant", [(np.int32, None, None), (np.int64, None, None), (np.float32, 1.0, 1.0e-6), (np.float64, 1.0, 1.0e-7)])
def test_reference_helpers_recipe(fa, shape, dt, sigma, quant):
    """tests/array.py:26-146 (test_helpers) as written there: every dtype, both shapes, full decode and
    the 10-sample slice around the middle; ints exact, floats within 10 quanta."""
    dt = np.dtype(dt)
    x = _reference_fake_data(shape, sigma, dt)
    is64 = dt in (np.dtype(np.int64), np.dtype(np.float64))
    ftol = 1.0e-5 if quant is None else 10.0 * quant
    first, last = shape[-1] // 2 - 5, shape[-1] // 2 + 5
    comp, starts, nbytes, off, gain = fa.array_compress(x, level=5, quanta=quant)
    if quant is None:
        assert off is None and gain is None
    full = fa.array_decompress(comp, shape[-1], starts, nbytes, stream_offsets=off, stream_gains=gain, is_int64=is64)
    part = fa.array_decompress(comp, shape[-1], starts, nbytes, stream_offsets=off, stream_gains=gain,
                               first_stream_sample=first, last_stream_sample=last, is_int64=is64)
    assert full.dtype == dt and full.shape == x.shape and part.shape == x[..., first:last].shape
    if quant is None:
        assert np.array_equal(full, x) and np.array_equal(part, x[..., first:last])
    else:
        assert np.allclose(full, x, atol=ftol) and np.allclose(part, x[..., first:last], atol=ftol)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_reference_float_conversion_recipes(fa, dt):
    """tests/utils.py:22-108 (test_float64 / test_float32): float_to_int -> int_to_float with a scalar
    quanta, a scalar and a per-stream precision, and a per-stream quanta array."""
    shape = (4, 3, 1000)
    data = _reference_fake_data(shape, 1.0, np.dtype(np.float64)).astype(dt)
    if dt == np.float64:
        i, o, g = fa.float_to_int(data, quanta=1.0e-16, precision=None)
        assert i.dtype == np.int64 and np.allclose(fa.int_to_float(i, o, g), data, rtol=1e-15, atol=1e-15)
    else:
        i, o, g = fa.float_to_int(data, quanta=1e-6, precision=None)
        assert i.dtype == np.int32 and np.allclose(fa.int_to_float(i, o, g), data, rtol=1e-5, atol=1e-5)
    for kw in (dict(quanta=None, precision=5), dict(quanta=None, precision=5 * np.ones(shape[:-1])),
               dict(quanta=1e-5, precision=None), dict(quanta=1e-5 * np.ones(shape[:-1]), precision=None)):
        i, o, g = fa.float_to_int(data, **kw)
        assert o.shape == shape[:-1] and g.shape == shape[:-1]
        assert np.allclose(fa.int_to_float(i, o, g), data, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("group_kind", ["h5", "zarr3"])
@pytest.mark.parametrize("shape", [(4, 3, 1000), (10000,)])
def test_reference_io_recipes(fa, group_kind, shape):
    """tests/hdf5.py:29-160 and tests/zarr.py:28-160: write_array / read_array and FlacArray.write_* /
    read_* for every dtype (ints exact, floats atol 1e-6), on in-memory stand-ins for the group objects."""
    from flacarray_amd import hdf5 as H
    from flacarray_amd import hdf5 as Z  # (the Zarr layout is the HDF5 one: same functions)
    from tests.conftest import FakeH5Group, FakeZarr3Group

    mod, new = (H, FakeH5Group) if group_kind == "h5" else (Z, FakeZarr3Group)
    for dt, sigma, quant in [(np.int32, None, None), (np.int64, None, None), (np.float32, 1.0, 1.0e-7), (np.float64, 1.0, 1.0e-15)]:
        dt = np.dtype(dt)
        x = _reference_fake_data(shape, sigma, dt)
        g = new()
        mod.write_array(x, g, level=5, quanta=quant, precision=None, mpi_comm=None, use_threads=True)
        check = mod.read_array(g, keep=None, stream_slice=None, keep_indices=False, use_threads=True)
        assert check.dtype == dt and check.shape == x.shape
        assert np.array_equal(check, x) if quant is None else np.allclose(check, x, atol=1e-6)
        fl = fa.FlacArray.from_array(x, quanta=quant, use_threads=True)
        g2 = new()
        (fl.write_hdf5 if group_kind == "h5" else fl.write_zarr)(g2)
        back = (fa.FlacArray.read_hdf5 if group_kind == "h5" else fa.FlacArray.read_zarr)(g2)
        assert back == fl


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sweep_bytes_equal_oracle(fa, oracle, seed):
    """Randomised sweep (tools/fuzz_parity.py): shapes, lengths around every block boundary, levels 0-8, six signal
    classes incl. spikes / wasted bits / extreme values, int32 and int64 -- the HIP encoder's bytes equal the
    oracle's and decode(encode(x)) == x in every case.  (8000 cases / 204 Msamples of the same generator pass.)"""
    import importlib.util

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, tot = mod.run(150, seed, verbose=False)
    assert bad == 0 and tot > 0
    bad, tot = mod.run(100, seed + 100, verbose=False, single_pass_only=True)  # geometries of the single-pass kernel only
    assert bad == 0 and tot > 0


def test_host_abi_pipeline_many_chunks(fa, oracle, monkeypatch):
    """The host-pointer entry points cut the array into chunks, upload chunk c+1 on a feeder thread while chunk c is
    encoded / decoded and copied back, and place every chunk's bytes at its final offset of one reserved blob
    (flacarray_hip.hip: encode_host / decode_host).  With the chunk size forced down to a few streams the result must
    be what one chunk gives: the oracle's bytes, and the input back -- also for a scattered, reordered subset of the
    streams (only their byte ranges are uploaded) and through the fused float32 entry."""
    monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(3 * 20000 * 4))  # 3 streams per chunk -> 13 chunks, the last one short
    x = sinusoid_noise_i32(37, 20000, seed=77)
    comp, st, nb = fa.encode_flac(x, 5)
    blob_o, st_o, nb_o = oracle.encode_i32(x, 5)
    assert np.array_equal(comp, blob_o) and np.array_equal(st, st_o) and np.array_equal(nb, nb_o)
    assert np.array_equal(fa.decode_flac(np.asarray(comp), st, nb, 20000), x)
    assert np.array_equal(fa.decode_flac(np.asarray(comp), st, nb, 20000, first_sample=4000, last_sample=9001), x[:, 4000:9001])
    # a short sample range of every stream: rows of 400 B, so the chunks are cut by the COMPRESSED bytes of their
    # streams (about five streams of ~46 KB per chunk here), not by the decoded bytes (which would make one chunk)
    assert np.array_equal(fa.decode_flac(np.asarray(comp), st, nb, 20000, first_sample=4090, last_sample=4190), x[:, 4090:4190])
    pick = np.array([30, 2, 17, 3, 36, 0, 18])  # scattered and out of order: arbitrary starts / nbytes (decompress.c:194-313)
    assert np.array_equal(fa.decode_flac(np.asarray(comp), st[pick].copy(), nb[pick].copy(), 20000), x[pick])
    # more scattered byte ranges than one copy each is worth (> 2048 pieces): everything between the first and the last
    xs = sinusoid_noise_i32(5000, 64, seed=78)
    cs, ss, ns = fa.encode_flac(xs, 5)
    odd = np.arange(1, 5000, 2)
    monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(1 << 30))  # one chunk: 2500 separate ranges in it
    assert np.array_equal(fa.decode_flac(np.asarray(cs), ss[odd].copy(), ns[odd].copy(), 64), xs[odd])
    monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(3 * 20000 * 4))
    # int64 (two channels) through the same pipeline
    x64 = (x.astype(np.int64) << 20) + 12345
    c64, s64, n64 = fa.encode_flac(x64, 5)
    bo, so, no = oracle.encode_i64(x64, 5)
    assert np.array_equal(c64, bo) and np.array_equal(s64, so)
    assert np.array_equal(fa.decode_flac(np.asarray(c64), s64, n64, 20000, is_int64=True), x64)
    # float32 in one trip (fa_encode_f32_host) == float_to_int followed by encode_flac, for a geometry the single-pass kernel
    # takes (8192 = two full frames) and for one it does not (20000)
    from flacarray_amd.libflacarray import encode_flac_f32

    for n in (8192, 20000):
        xf = sinusoid_noise_f32(11, n, seed=5)
        for q in (None, np.full(11, 2.0**-12, np.float32) * (1 + np.arange(11) % 3)):
            monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(4 * n * 4))
            got = encode_flac_f32(xf, q, 5)
            ints, off, gain = fa.float_to_int(xf, quanta=None if q is None else q)
            c2, s2, n2 = fa.encode_flac(ints, 5)
            assert np.array_equal(got[0], c2) and np.array_equal(got[1], s2) and np.array_equal(got[2], n2)
            assert np.array_equal(got[3], off) and np.array_equal(got[4], gain)
    with pytest.raises(RuntimeError, match="NaNs"):
        bad = xf.copy()
        bad[3, 17] = np.nan
        fa.array_compress(bad, quanta=1e-3)
    # float64 in one trip (fa_encode_f64_host) == float_to_int followed by encode_flac (two-channel streams), several chunks
    from flacarray_amd.libflacarray import encode_flac_f64

    xd = (sinusoid_noise_f32(9, 12000, seed=8).astype(np.float64) * 1.0e3) + np.arange(9)[:, None] * 0.125
    for q in (None, np.full(9, 2.0**-20) * (1 + np.arange(9) % 3)):
        monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(2 * 12000 * 8))
        got = encode_flac_f64(xd, q, 5)
        ints, off, gain = fa.float_to_int(xd, quanta=None if q is None else q)
        c2, s2, n2 = fa.encode_flac(ints, 5)
        assert np.array_equal(got[0], c2) and np.array_equal(got[1], s2) and np.array_equal(got[2], n2)
        assert np.array_equal(got[3], off) and np.array_equal(got[4], gain)
    comp64 = fa.array_compress(xd, precision=9)
    back = fa.array_decompress(comp64[0], xd.shape[1], comp64[1], comp64[2], stream_offsets=comp64[3], stream_gains=comp64[4], is_int64=True)
    assert np.all(np.abs(back - xd) <= 0.5 * (np.std(xd, axis=-1, keepdims=True) / 1e9) * (1 + 1e-9) + 4 * np.finfo(np.float64).eps * np.abs(xd).max())
    with pytest.raises(RuntimeError, match="NaNs"):
        bad = xd.copy()
        bad[2, 5] = np.nan
        fa.array_compress(bad, precision=5)
    # decode + restore in one trip (fa_decode_f32_host / fa_decode_f64_host) == decode_flac followed by int_to_float, bit for
    # bit: whole streams, a sample range, a scattered subset, several chunks
    from flacarray_amd.libflacarray import decode_flac_restore

    for arr, is64 in ((xf, False), (xd, True)):
        monkeypatch.setenv("FLACARRAY_HIP_HOST_CHUNK_BYTES", str(3 * arr.shape[1] * arr.itemsize))
        comp, st, nb, off, gain = fa.array_compress(arr, quanta=2.0**-14)
        for first, last in ((-1, -1), (4000, 9001)):
            ints = fa.decode_flac(comp, st, nb, arr.shape[1], first_sample=first, last_sample=last, is_int64=is64)
            want = fa.int_to_float(ints, off, gain)
            got = decode_flac_restore(comp, st, nb, arr.shape[1], off, gain, first_sample=first, last_sample=last, is_int64=is64)
            assert got is not None and got.dtype == arr.dtype and np.array_equal(got, want)
        pick = np.array([7, 0, 5])
        got = decode_flac_restore(comp, st[pick].copy(), nb[pick].copy(), arr.shape[1], off[pick].copy(), gain[pick].copy(), is_int64=is64)
        assert np.array_equal(got, fa.int_to_float(fa.decode_flac(comp, st[pick].copy(), nb[pick].copy(), arr.shape[1], is_int64=is64), off[pick], gain[pick]))
        keep = np.zeros(arr.shape[0], dtype=bool)
        keep[[1, 4]] = True
        sub, idx = fa.array_decompress_slice(comp, arr.shape[1], st, nb, stream_offsets=off, stream_gains=gain, keep=keep,
                                             first_stream_sample=100, last_stream_sample=350, is_int64=is64)
        assert idx == [(1,), (4,)] and np.array_equal(sub, fa.int_to_float(fa.decode_flac(comp, st[keep].copy(), nb[keep].copy(), arr.shape[1], 100, 350, is_int64=is64), off[keep], gain[keep]))


def test_latency_decoder_matches_throughput_decoder(fa, oracle, monkeypatch):
    """K7L (one wavefront per frame, csrc/decode_latency.hpp) serves launches with few frames -- the reference's usage
    pattern, one small read per call (array.py:409-449 -> decompress.c:281-298).  For every kind of frame the encoder
    writes (LPC / FIXED / VERBATIM / CONSTANT, wasted bits, 1152- and 4096-sample blocks, partition orders 0..6, a short
    last frame) a batch of random slices decoded by K7L must equal the same batch decoded by K7 and the source."""
    import torch

    rng = np.random.default_rng(99)
    n = 30000
    kinds = {
        "sinus": sinusoid_noise_i32(3, n, seed=5),
        "full": full_range_i32((3, n), seed=6),
        "small": rng.integers(-40, 40, (3, n)).astype(np.int32),
        "wasted": (sinusoid_noise_i32(3, n, seed=7) >> 8) << 8,
        "const": np.full((3, n), -77, np.int32),
        "ramp": np.broadcast_to(np.arange(n, dtype=np.int32) * 3 - 5000, (3, n)).copy(),
        "bursts": (rng.integers(-5, 5, (3, n)) * (1 + 4000 * (rng.random((3, n)) < 0.002))).astype(np.int32),
    }
    for name, x in kinds.items():
        for level in (0, 3, 5, 8):
            blob, st, nb = oracle.encode_i32(x, level)
            dev = (torch.from_numpy(blob).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(nb).cuda())
            ix = fa.DeviceDecodeIndex(*dev, n)
            try:
                ch = rng.integers(0, 3, 40)
                cnt = rng.integers(1, 9000, 40)
                first = np.array([rng.integers(0, n - c + 1) for c in cnt])
                outs = {}
                for mode in ("0", "1"):
                    monkeypatch.setenv("FLACARRAY_HIP_LATENCY", mode)
                    flat, off = ix.decode_slices(ch, first, cnt)
                    outs[mode] = flat.cpu().numpy()
                    one, _ = ix.decode_slices(ch[:1], first[:1], cnt[:1])  # a single read: the task table rides in the kernel arguments
                    assert np.array_equal(one.cpu().numpy(), x[ch[0], first[0] : first[0] + cnt[0]]), (name, level, mode)
                assert np.array_equal(outs["0"], outs["1"]), (name, level)
                want = np.concatenate([x[c, f : f + k] for c, f, k in zip(ch, first, cnt)])
                assert np.array_equal(outs["1"], want), (name, level)
                monkeypatch.setenv("FLACARRAY_HIP_LATENCY", "1")
                assert np.array_equal(ix.decode(100, 9100).cpu().numpy(), x[:, 100:9100])  # grid mode through K7L
            finally:
                ix.close()


def test_small_reads_landing_in_pinned_host_memory(fa, oracle):
    """A read of up to 512 KB that wants numpy output has its samples and the status word written by the latency decoder
    straight into pinned host memory (csrc/flacarray_hip.hip, decode_device_impl).  Same samples as a device read; a
    frame the latency decoder does not take (predictor order 20 in the hand-assembled vector g7) makes the call fall
    back to the ordinary path with the same result; larger results go by copy."""
    import os

    import torch

    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flac_vectors.npz"))
    s, st, n = v["g7_deep_samples"], v["g7_deep_stream"], int(v["g7_deep_size"])
    blob = np.concatenate([st, np.zeros((-st.size) % 16, np.uint8), st])
    second = st.size + (-st.size) % 16
    dev = (torch.from_numpy(blob).cuda(), torch.tensor([0, second], dtype=torch.int64).cuda(), torch.full((2,), st.size, dtype=torch.int64).cuda())
    ix = fa.DeviceDecodeIndex(*dev, n)
    try:
        rng = np.random.default_rng(4)
        for _ in range(20):
            k = int(rng.integers(1, 6))
            ch = rng.integers(0, 2, k)
            cnt = rng.integers(1, min(n, 3000), k)
            first = np.array([rng.integers(0, n - c + 1) for c in cnt])
            host, off = ix.decode_slices(ch, first, cnt, to_host=True)
            flat, _ = ix.decode_slices(ch, first, cnt)
            want = np.concatenate([s[f : f + c] for f, c in zip(first, cnt)])
            assert isinstance(host, np.ndarray) and np.array_equal(host, want) and np.array_equal(flat.cpu().numpy(), want)
    finally:
        ix.close()
    # own streams: small (pinned) and large (copied) host results
    x = sinusoid_noise_i32(6, 100000, seed=77)
    b2, st2, nb2 = oracle.encode_i32(x, 5)
    ix = fa.DeviceDecodeIndex(torch.from_numpy(b2).cuda(), torch.from_numpy(st2).cuda(), torch.from_numpy(nb2).cuda(), 100000)
    try:
        for cnt in (1, 4095, 4097, 70000):  # 70000 x 6 x 4 bytes > 512 KB
            ch = np.arange(6)
            first = np.full(6, 12345)
            host, off = ix.decode_slices(ch, first, np.full(6, cnt), to_host=True)
            assert np.array_equal(host.reshape(6, cnt), x[:, 12345 : 12345 + cnt])
    finally:
        ix.close()


def test_single_pass_capacity_smaller_than_worst_case(fa, oracle):
    """The single-pass encoder writes into a caller-provided buffer.  Sized below the worst case it must either hold
    the blob (same bytes) or refuse cleanly with ERROR_ALLOC -- no frame, tail frame or stream header may be written
    outside it (the words behind the buffer stay untouched)."""
    import torch

    from flacarray_amd import _lib
    from flacarray_amd.libflacarray import EncodeWorkspace

    for n in (8192 * 3, 8192 * 3 + 1000):  # whole frames / a short last frame per stream (slot + move)
        x = sinusoid_noise_i32(24, n, seed=31)
        blob_o, st_o, nb_o = oracle.encode_i32(x, 5)
        xd = torch.from_numpy(x).cuda()
        comp, st, nb = fa.encode_flac_device(xd, level=5, capacity_bytes=int(blob_o.size) + 64)
        assert np.array_equal(comp.cpu().numpy(), blob_o) and np.array_equal(st.cpu().numpy(), st_o)
        with pytest.raises(RuntimeError, match="return code = 1$"):
            fa.encode_flac_device(xd, level=5, capacity_bytes=int(blob_o.size) // 2)
        with pytest.raises(RuntimeError, match="return code = 1$"):
            fa.encode_flac_device(xd, level=5, capacity_bytes=int(blob_o.size) - 8)  # only the last (tail) frame does not fit
        # the C entry point with a guard zone behind the buffer
        L = _lib.lib()
        cap = int(blob_o.size) - 3000
        buf = torch.full((cap + 65536,), 0xA5, dtype=torch.uint8, device="cuda")
        ws = EncodeWorkspace().get(L.fa_encode_single_pass_workspace_bytes(24, n, 5), xd.device)
        d_st = torch.empty(24, dtype=torch.int64, device="cuda")
        d_nb = torch.empty(24, dtype=torch.int64, device="cuda")
        import ctypes

        total = ctypes.c_int64(0)
        rc = L.fa_encode_i32_device(ctypes.c_void_p(xd.data_ptr()), 24, n, 5, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                    ctypes.c_void_p(buf.data_ptr()), cap, ctypes.c_void_p(d_st.data_ptr()),
                                    ctypes.c_void_p(d_nb.data_ptr()), ctypes.byref(total), None, None)
        assert rc == 1
        assert bool((buf[cap:] == 0xA5).all()), "bytes behind the caller's buffer were written"


def test_int64_side_right_decision(fa, oracle):
    """The reference's two-channel path lets libFLAC choose a stereo assignment per frame (compress.c:482-540).  The
    encoder here codes the first channel as SIDE (low word - high word, assignment 0b1001) when the low word is small, the
    high word is not zero throughout and the analysis estimates fewer bits for the side -- the one assignment that pays
    on (low word, high word) pairs.  HIP bytes must equal the oracle's, frames of both assignments must occur, and the
    throughput decoder, the oracle's decoder and the independent pure-Python decoder must all return the input."""
    import torch

    from tests.golden import pyflac

    rng = np.random.default_rng(17)
    n = 13000
    x = np.empty((4, n), dtype=np.int64)
    x[0] = rng.integers(-3, 4, n)                      # both signs, tiny: side + right wins
    x[1] = rng.integers(0, 6, n)                       # high word zero throughout: no trial, independent channels
    x[2] = np.rint(rng.normal(0, 40, n))               # small enough for the trial
    x[3] = np.rint(rng.normal(0, 5000, n))             # low word too large: independent channels
    for level in (0, 5, 8):
        blob_o, st_o, nb_o = oracle.encode_i64(x, level)
        comp, st, nb = fa.encode_flac_device(torch.from_numpy(x).cuda(), level=level)
        assert np.array_equal(comp.cpu().numpy(), blob_o), level
        assert np.array_equal(st.cpu().numpy(), st_o) and np.array_equal(nb.cpu().numpy(), nb_o)
        y = fa.decode_flac_device(comp, st, nb, n, is_int64=True).cpu().numpy()
        assert np.array_equal(y, x)
        assert np.array_equal(oracle.decode_i64(blob_o, st_o, nb_o, n), x)
        kinds = set()
        for i in range(4):
            stream = bytes(blob_o[st_o[i] : st_o[i] + nb_o[i]])
            out, sinfo = pyflac.decode_stream(stream)  # (sample-interleaved low / high words)
            assigns = [fr["assignment"] for fr in sinfo["frames"]]
            lo, hi = np.asarray(out[0::2], dtype=np.int64), np.asarray(out[1::2], dtype=np.int64)
            assert np.array_equal((hi << 32) | (lo & 0xFFFFFFFF), x[i]), (level, i)
            kinds |= {(i, a) for a in assigns}
        assert {a for (i, a) in kinds if i == 0} == {9}, "tiny values of both signs: every frame side + right"
        assert {a for (i, a) in kinds if i in (1, 3)} == {1}, "no trial where the high word is zero or the low word large"



@pytest.mark.parametrize("block", [8192, 16384])
def test_few_slices_of_streams_with_large_blocks(fa, block):
    """Foreign streams with blocks above 4096 samples (libFLAC writes them when asked to; the decoder takes up to
    65535): K7L does not serve them, so a handful of slices -- whose task table would otherwise ride in K7L's kernel
    arguments -- must reach K7 with the table in memory.  One-off calls and through a decode index, 1 to 8 slices."""
    import torch

    from tests.golden.make_golden import frame, stream

    rng = np.random.default_rng(block)
    n = 2 * block + 777
    rows, blobs = [], []
    for s in range(3):
        x = (np.cumsum(rng.integers(-40, 41, n)) + 1000 * s).tolist()
        frs = [frame(x[f * block : (f + 1) * block], f, 32, {"type": "fixed", "order": 1, "porder": 1, "params": [6, 6]} if f < 2 else
                     {"type": "fixed", "order": 1, "porder": 0, "params": [6]}) for f in range(3)]
        rows.append(x)
        blobs.append(np.frombuffer(stream(frs, block, 32, n), dtype=np.uint8))
    x = np.array(rows, dtype=np.int32)
    nb = np.array([b.size for b in blobs], dtype=np.int64)
    st = np.concatenate([[0], np.cumsum(nb)[:-1]]).astype(np.int64)
    blob = np.concatenate(blobs)
    dev = torch.device("cuda", 0)
    tb, ts, tn = (torch.from_numpy(a).to(dev) for a in (blob, st, nb))
    assert np.array_equal(fa.decode_flac_device(tb, ts, tn, n).cpu().numpy(), x)
    idx = fa.DeviceDecodeIndex(tb, ts, tn, n)
    for k in range(1, 9):
        ss = rng.integers(0, 3, k)
        cnt = rng.integers(1, 300, k)
        first = np.array([rng.integers(0, n - c + 1) for c in cnt])
        if k == 2:
            first[0], cnt[0] = block - 5, 10  # across a frame boundary
        for got, offs in (fa.decode_slices_device(tb, ts, tn, n, ss, first, cnt), idx.decode_slices(ss, first, cnt)):
            flat = got.cpu().numpy()
            for o, s_i, f0, c in zip(offs, ss, first, cnt):
                assert np.array_equal(flat[o : o + c], x[s_i, f0 : f0 + c]), (k, s_i, f0, c)
    idx.close()


@pytest.mark.parametrize("kind", ["noise", "verbatim_mix"])
def test_many_frames_in_flight_bytes_equal_oracle(fa, oracle, kind):
    """Byte identity with thousands of frames in flight.  With a handful of streams every frame finds its byte offset
    waiting; at this size most frames fill their bit ring before the scanner has their offset and take the other path of
    the single-pass writer -- completed blocks parked in registers, stored once the offset arrives (encode_fused.hpp,
    flush_blocks) --, and frames longer than ring + park (VERBATIM: 65 blocks) also reach the spin behind it."""
    import torch

    n_ch, n = 384, 16 * 4096
    x = sinusoid_noise_i32(n_ch, n, seed=31)
    if kind == "verbatim_mix":
        full = full_range_i32((n_ch, n), seed=32)
        sel = (np.arange(n_ch)[:, None] + np.arange(n // 4096)[None, :]) % 3 == 0  # every third frame incompressible
        x = np.where(np.repeat(sel, 4096, axis=1), full, x).astype(np.int32)
    oracle.lib().oracle_set_threads(8)
    blob_o, st_o, nb_o = oracle.encode_i32(x, 5, use_threads=True)
    comp, st, nb = fa.encode_flac_device(torch.from_numpy(x).cuda(), level=5)
    assert np.array_equal(st.cpu().numpy(), st_o) and np.array_equal(nb.cpu().numpy(), nb_o)
    assert np.array_equal(comp.cpu().numpy(), blob_o)
    assert torch.equal(fa.decode_flac_device(comp, st, nb, n, verify=True).cpu(), torch.from_numpy(x))


def test_side_channel_that_outgrows_32_bits_under_an_lpc_predictor(fa):
    """A side / right frame whose side channel starts small (33-bit warm-up sample that fits 32 bits) and grows past
    2^31 under an LPC predictor: the wave-per-frame decoder keeps side channels modulo 2^32, which an LPC recurrence does
    not survive -- it has to notice (its 32-bit stores saturate) and leave the frame to K7, whose side channels are
    doubles.  The same frame with a FIXED predictor (exact modulo 2^32) is decoded in place."""
    import torch

    from tests.golden.make_golden import frame, stream

    n = 64
    side = [i * (1 << 26) for i in range(n)]          # 0 .. 63 * 2^26 = 4.2e9 > 2^31: needs the 33rd bit from sample 32 on
    right = [-(v >> 1) for v in side]                 # left = side + right = side - side / 2 stays inside int32
    left = [a + b for a, b in zip(side, right)]
    assert max(left) < 2**31 and min(right) >= -(2**31) and max(side) >= 2**31
    want = np.array([(r << 32) | (l & 0xFFFFFFFF) for l, r in zip(left, right)], dtype=np.int64)
    vb = {"type": "verbatim"}
    for sub in ({"type": "lpc", "order": 1, "coefs": [1], "shift": 0, "precision": 2, "porder": 0, "params": [27], "rice2": True},
                {"type": "fixed", "order": 1, "porder": 0, "params": [27], "rice2": True}):
        data = stream([frame([side, right], 0, 32, [sub, vb], assignment=9)], n, 32, n, channels=2)
        blob = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        st, nb = torch.zeros(1, dtype=torch.int64).cuda(), torch.tensor([len(data)], dtype=torch.int64).cuda()
        got = fa.decode_flac_device(blob, st, nb, n, is_int64=True)
        assert np.array_equal(got.cpu().numpy()[0], want), sub["type"]
        part = fa.decode_flac_device(blob, st, nb, n, 30, 50, is_int64=True)
        assert np.array_equal(part.cpu().numpy()[0], want[30:50]), sub["type"]
