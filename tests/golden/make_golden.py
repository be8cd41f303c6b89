#!/usr/bin/env python3
"""Generate tests/golden/flac_vectors.npz: hand-assembled native-FLAC streams (RFC 9639).

Independent of the oracle and of the HIP path: streams are assembled field by field as strings
of '0'/'1' with explicitly chosen coding parameters (subframe type, predictor order,
coefficients, partition order, Rice parameters, escapes, wasted bits, blocksize/sample-size
codes), so they pin the DECODERS against the published format rather than against our own
encoder.  tests/golden/pyflac.py holds the matching independent decoder that pins the oracle
ENCODER.  Known-answer CRCs: CRC-8/SMBUS("123456789") = 0xF4, CRC-16/UMTS("123456789") = 0xFEE8.

No reference source is involved: the reference tree holds no FLAC golden vectors at all
(SURVEY.md 8c); these are data made from the public specification.
"""
import os

import numpy as np


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def ubits(v, n):
    assert 0 <= v < (1 << n), (v, n)
    return format(v, "0%db" % n) if n else ""


def sbits(v, n):
    return ubits(v & ((1 << n) - 1), n)


def utf8(v):
    if v < 0x80:
        return ubits(v, 8)
    n = 2
    while v >= (1 << (5 * n + 1)):
        n += 1
    out = "1" * n + "0" + ubits(v >> (6 * (n - 1)), 7 - n)
    for i in range(n - 2, -1, -1):
        out += "10" + ubits((v >> (6 * i)) & 0x3F, 6)
    return out


def to_bytes(bits):
    assert len(bits) % 8 == 0
    return bytes(int(bits[i : i + 8], 2) for i in range(0, len(bits), 8))


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def rice(r, k):
    u = (r << 1) if r >= 0 else ((-r) << 1) - 1
    return "0" * (u >> k) + "1" + ubits(u & ((1 << k) - 1), k)


def residual(res, order, porder, params, bs, rice2=False, escapes=None):
    """params[p]: Rice parameter or ('esc', width)."""
    plen = 5 if rice2 else 4
    out = ("01" if rice2 else "00") + ubits(porder, 4)
    ps = bs >> porder
    i = 0
    for p in range(1 << porder):
        n = ps - order if p == 0 else ps
        par = params[p]
        if isinstance(par, tuple):
            w = par[1]
            out += "1" * plen + ubits(w, 5)
            for _ in range(n):
                assert -(1 << (w - 1)) <= res[i] < (1 << (w - 1)), "escape width too small"
                out += sbits(res[i], w)
                i += 1
        else:
            out += ubits(par, plen)
            for _ in range(n):
                out += rice(res[i], par)
                i += 1
    assert i == len(res)
    return out


def subframe_bits(samples, bps, sub):
    """sub: dict(type='const'|'verbatim'|'fixed'|'lpc', order, coefs, shift, precision, porder, params, rice2, wasted)."""
    bs = len(samples)
    wasted = sub.get("wasted", 0)
    x = [int(v) >> wasted for v in samples]
    b = bps - wasted
    t = sub["type"]
    order = sub.get("order", 0)
    tc = {"const": 0, "verbatim": 1}.get(t)
    if t == "fixed":
        tc = 8 + order
    if t == "lpc":
        tc = 32 + order - 1
    body = "0" + ubits(tc, 6) + ("1" + "0" * (wasted - 1) + "1" if wasted else "0")
    if t == "const":
        body += sbits(x[0], b)
    elif t == "verbatim":
        body += "".join(sbits(v, b) for v in x)
    else:
        body += "".join(sbits(v, b) for v in x[:order])
        if t == "fixed":
            coefs = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
            shift = 0
        else:
            coefs, shift, prec = sub["coefs"], sub["shift"], sub["precision"]
            body += ubits(prec - 1, 4) + sbits(shift, 5) + "".join(sbits(c, prec) for c in coefs)
        res = []
        for i in range(order, bs):
            pred = sum(c * x[i - 1 - j] for j, c in enumerate(coefs)) >> shift
            res.append(x[i] - pred)
        body += residual(res, order, sub["porder"], sub["params"], bs, sub.get("rice2", False))
    return body


def frame(samples, frame_no, bps, sub, ss_code=None, force_bs_code=None, assignment=0):
    """One frame.  Mono: `samples` is the channel and `sub` its description.  Two channels
    (assignment 1 left/right, 8 left/side, 9 side/right, 10 mid/side): `samples` is the pair of
    CODED channels (the caller has already formed side / mid) and `sub` the pair of descriptions;
    a side channel is coded with one extra bit per sample."""
    chans = [samples] if assignment == 0 else list(samples)
    subs = [sub] if assignment == 0 else list(sub)
    bs = len(chans[0])
    code = force_bs_code if force_bs_code is not None else BS_CODES.get(bs, 6 if bs <= 256 else 7)
    hdr = "11111111111110" + "0" + "0" + ubits(code, 4) + ubits(9, 4) + ubits(assignment, 4) + ubits(SS_CODES[bps] if ss_code is None else ss_code, 3) + "0"
    hdr += utf8(frame_no)
    if code == 6:
        hdr += ubits(bs - 1, 8)
    elif code == 7:
        hdr += ubits(bs - 1, 16)
    hdr += ubits(crc8(to_bytes(hdr)), 8)
    extra = {0: [0], 1: [0, 0], 8: [0, 1], 9: [1, 0], 10: [0, 1]}[assignment]
    bits = hdr + "".join(subframe_bits(c, bps + e, sb) for c, e, sb in zip(chans, extra, subs))
    bits += "0" * ((-len(bits)) % 8)
    raw = to_bytes(bits)
    return raw + crc16(raw).to_bytes(2, "big")


def stream(frames_bytes, blocksize, bps, total, extra_blocks=(), channels=1):
    si = ubits(blocksize, 16) * 2 + ubits(0, 24) * 2 + ubits(44100, 20) + ubits(channels - 1, 3) + ubits(bps - 1, 5) + ubits(total, 36) + "0" * 128
    blocks = [(0, to_bytes(si))] + list(extra_blocks)
    out = b"fLaC"
    for i, (typ, payload) in enumerate(blocks):
        last = 0x80 if i == len(blocks) - 1 else 0
        out += bytes([last | typ]) + len(payload).to_bytes(3, "big") + payload
    return out + b"".join(frames_bytes)


def build():
    rng = np.random.default_rng(20240229)
    vec = {}

    def add(name, samples, data, stream_size):
        vec[name + "_samples"] = np.asarray(samples, dtype=np.int32)
        vec[name + "_stream"] = np.frombuffer(data, dtype=np.uint8)
        vec[name + "_size"] = np.int64(stream_size)

    # g1: CONSTANT, blocksize 16 (8-bit blocksize code), 32 bps
    s = [-5] * 16
    add("g1_const", s, stream([frame(s, 0, 32, {"type": "const"})], 16, 32, 16), 16)
    # g2: VERBATIM with INT32_MIN / INT32_MAX
    s = [-(2**31), 2**31 - 1, 0, -1, 123456789]
    add("g2_verbatim", s, stream([frame(s, 0, 32, {"type": "verbatim"})], 5, 32, 5), 5)
    # g3: FIXED order 2, one Rice partition k=3, blocksize 192 (code 1), two frames + a short last frame
    s = (np.cumsum(np.cumsum(rng.integers(-6, 7, 192 * 2 + 50))) + 1000).tolist()
    frs = [frame(s[0:192], 0, 32, {"type": "fixed", "order": 2, "porder": 0, "params": [3]}),
           frame(s[192:384], 1, 32, {"type": "fixed", "order": 2, "porder": 0, "params": [3]}),
           frame(s[384:], 2, 32, {"type": "fixed", "order": 1, "porder": 0, "params": [6]})]
    add("g3_fixed", s, stream(frs, 192, 32, len(s)), len(s))
    # g4: LPC order 2 (coefs 1.988, -1.0 at shift 8, precision 11), partition order 2, Rice2 with a
    #     parameter above 14 and one escaped partition; blocksize 256 (code 8)
    t = np.arange(256)
    s = np.rint(50000 * np.sin(t / 9.0) + rng.normal(0, 40, 256)).astype(np.int64)
    s[200:] = s[200:] + rng.integers(-(2**20), 2**20, 56)  # noisy tail -> large parameter
    s = s.tolist()
    add("g4_lpc", s, stream([frame(s, 0, 32, {"type": "lpc", "order": 2, "coefs": [509, -256], "shift": 8, "precision": 11,
                                               "porder": 2, "params": [7, ("esc", 12), 8, 21], "rice2": True})], 256, 32, 256), 256)
    # g5: wasted bits (3), FIXED order 0, 16-bit blocksize code (1000 samples)
    s = (rng.integers(-3000, 3000, 1000) * 8).tolist()
    add("g5_wasted", s, stream([frame(s, 0, 32, {"type": "fixed", "order": 0, "porder": 0, "params": [11], "wasted": 3})], 1000, 32, 1000), 1000)
    # g6: 16-bit stream, sample size taken from STREAMINFO (code 0), PADDING + VORBIS_COMMENT blocks to skip,
    #     FIXED order 4 with 8 partitions
    s = np.rint(12000 * np.sin(np.arange(512) / 20.0) + rng.normal(0, 3, 512)).astype(np.int64).tolist()
    vorbis = (4).to_bytes(4, "little") + b"test" + (0).to_bytes(4, "little")
    add("g6_16bit", s, stream([frame(s, 0, 16, {"type": "fixed", "order": 4, "porder": 3, "params": [3, 3, 4, 3, 2, 3, 3, 5]}, ss_code=0)],
                              512, 16, 512, extra_blocks=[(1, bytes(10)), (4, vorbis)]), 512)
    # g7: LPC order 12 and order 20 frames (history deeper than 8 / 16), 24-bit samples, frame numbers 0..129
    s = np.rint(3.0e6 * np.sin(np.arange(130 * 64) / 5.0) + rng.normal(0, 1000, 130 * 64)).astype(np.int64).tolist()
    c12 = [900, -300, 100, -50, 25, -12, 6, -3, 2, -1, 1, -1]
    c20 = c12 + [1, -1, 1, -1, 1, -1, 1, -1]
    frs = []
    for f in range(130):
        blk = s[64 * f : 64 * f + 64]
        sub = {"type": "lpc", "order": 12 if f % 2 == 0 else 20, "coefs": c12 if f % 2 == 0 else c20, "shift": 10, "precision": 12,
               "porder": 0, "params": [20], "rice2": True}
        frs.append(frame(blk, f, 24, sub))
    add("g7_deep", s, stream(frs, 64, 24, len(s)), len(s))
    # ---- two-channel streams: int64 samples = (channel 1 << 32) | (channel 0 as unsigned), the
    #      reference's split of int64 data into low / high words (utils.c:96-123) ----
    def add64(name, left, right, data):
        s64 = [(int(r) << 32) | (int(l) & 0xFFFFFFFF) for l, r in zip(left, right)]
        vec[name + "_samples"] = np.asarray(s64, dtype=np.int64)
        vec[name + "_stream"] = np.frombuffer(data, dtype=np.uint8)
        vec[name + "_size"] = np.int64(len(s64))

    n2 = 192
    ext = [2**31 - 1, -(2**31), -(2**31), 2**31 - 1, 0, -1, 1, 2**31 - 1]  # pairs that need all 33 bits of a side channel
    left = ext + rng.integers(-(2**31), 2**31, n2 - len(ext)).tolist()
    right = ext[::-1] + (np.cumsum(rng.integers(-9, 10, n2 - len(ext))) - 40).tolist()
    vb = {"type": "verbatim"}
    r0 = {"type": "fixed", "order": 0, "porder": 2, "params": [30, 7, 7, 7], "rice2": True}  # extremes only in the first partition
    # g8: independent channels: low word VERBATIM, high word FIXED order 0 (the encoder's own layout)
    add64("g8_stereo_lr", left, right, stream([frame([left, right], 0, 32, [vb, r0], assignment=1)], n2, 32, n2, channels=2))
    side = [l - r for l, r in zip(left, right)]
    mid = [(l + r) >> 1 for l, r in zip(left, right)]
    # g9: left/side, g10: side/right (33-bit VERBATIM side: its extremes +-(2^32 - 1) fit no 32-bit residual),
    # g11: mid/side with a Rice2-coded mid
    add64("g9_stereo_ls", left, right, stream([frame([left, side], 0, 32, [vb, vb], assignment=8)], n2, 32, n2, channels=2))
    add64("g10_stereo_sr", left, right, stream([frame([side, right], 0, 32, [vb, r0], assignment=9)], n2, 32, n2, channels=2))
    add64("g11_stereo_ms", left, right, stream([frame([mid, side], 0, 32, [{"type": "fixed", "order": 0, "porder": 0, "params": [30], "rice2": True}, vb],
                                                      assignment=10)], n2, 32, n2, channels=2))
    # g13: predictive 33-bit side channels: left near +2^31, right near -2^31, both smooth, so the side
    #      (about 2^32) needs 33-bit warm-up samples and a 64-bit predictor, while its residual is tiny
    t2 = np.arange(n2)
    lft = (2**31 - 5000 + np.rint(900 * np.sin(t2 / 7.0))).astype(np.int64) + rng.integers(-3, 4, n2)
    rgt = (-(2**31) + 5000 + np.rint(700 * np.cos(t2 / 5.0))).astype(np.int64) + rng.integers(-3, 4, n2)
    lft, rgt = lft.tolist(), rgt.tolist()
    sd = [a - c for a, c in zip(lft, rgt)]
    md = [(a + c) >> 1 for a, c in zip(lft, rgt)]
    lp = {"type": "lpc", "order": 2, "coefs": [1024, -512], "shift": 9, "precision": 12, "porder": 1, "params": [8, 8]}
    fx = {"type": "fixed", "order": 2, "porder": 0, "params": [7]}
    frs = [frame([lft[:64], sd[:64]], 0, 32, [fx, lp], assignment=8),
           frame([sd[64:128], rgt[64:128]], 1, 32, [fx, fx], assignment=9),
           frame([md[128:], sd[128:]], 2, 32, [lp, fx], assignment=10)]
    add64("g13_stereo_pred", lft, rgt, stream(frs, 64, 32, n2, channels=2))
    # g12: two frames of 16-bit stereo (sample size from the frame header), mid/side then left/right, wasted bits on one channel
    l16 = (rng.integers(-2000, 2000, 32) * 4).tolist()
    r16 = rng.integers(-30000, 30000, 32).tolist()
    m16 = [(a + c) >> 1 for a, c in zip(l16[:16], r16[:16])]
    s16 = [a - c for a, c in zip(l16[:16], r16[:16])]
    f0 = frame([m16, s16], 0, 16, [{"type": "verbatim"}, {"type": "fixed", "order": 0, "porder": 0, "params": [14]}], assignment=10)
    f1 = frame([l16[16:], r16[16:]], 1, 16, [{"type": "fixed", "order": 0, "porder": 0, "params": [9], "wasted": 2}, {"type": "verbatim"}], assignment=1)
    add64("g12_stereo16", l16, r16, stream([f0, f1], 16, 16, 32, channels=2))
    vec["crc8_check"] = np.uint8(crc8(b"123456789"))
    vec["crc16_check"] = np.uint16(crc16(b"123456789"))
    return vec


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flac_vectors.npz")
    v = build()
    assert v["crc8_check"] == 0xF4 and v["crc16_check"] == 0xFEE8
    np.savez_compressed(out, **v)
    print("wrote", out, {k: (v[k].shape if hasattr(v[k], "shape") else v[k]) for k in v if k.endswith("_stream")})
