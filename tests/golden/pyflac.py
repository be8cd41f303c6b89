"""Independent pure-Python native-FLAC decoder (RFC 9639), used only by the tests to pin the
oracle ENCODER: it shares no code with oracle/ or flacarray_amd/ and works on a Python bit
string, one field at a time.  Mono and two-channel streams (left/right, left/side, side/right,
mid/side; sample-interleaved output), every subframe type, Rice/Rice2, escapes, wasted bits;
both CRCs are verified."""
from .make_golden import crc8, crc16


class Bits:
    def __init__(self, data, pos=0):
        self.data = data
        self.pos = pos  # bit position

    def u(self, n):
        v = 0
        for _ in range(n):
            byte = self.data[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def s(self, n):
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        q = 0
        while self.u(1) == 0:
            q += 1
        return q


def decode_stream(data):
    """bytes -> (list of int samples, info dict)"""
    data = bytes(data)
    assert data[:4] == b"fLaC"
    off = 4
    info = {"blocks": []}
    while True:
        last, typ = data[off] >> 7, data[off] & 0x7F
        ln = int.from_bytes(data[off + 1 : off + 4], "big")
        body = data[off + 4 : off + 4 + ln]
        info["blocks"].append((typ, ln))
        if typ == 0:
            b = Bits(body)
            info["min_bs"], info["max_bs"] = b.u(16), b.u(16)
            b.u(48)
            info["rate"] = b.u(20)
            info["channels"] = b.u(3) + 1
            info["bps"] = b.u(5) + 1
            info["total"] = b.u(36)
        if typ == 3:
            pts = []
            for i in range(ln // 18):
                pts.append((int.from_bytes(body[18 * i : 18 * i + 8], "big"), int.from_bytes(body[18 * i + 8 : 18 * i + 16], "big"),
                            int.from_bytes(body[18 * i + 16 : 18 * i + 18], "big")))
            info["seektable"] = pts
        off += 4 + ln
        if last:
            break
    info["first_frame"] = off
    out = []
    frames = []
    fno = 0
    while off < len(data):
        start = off
        b = Bits(data, off * 8)
        assert b.u(14) == 0x3FFE and b.u(1) == 0
        assert b.u(1) == 0, "fixed blocksize expected"
        bsc, src, ch, ssc = b.u(4), b.u(4), b.u(4), b.u(3)
        assert b.u(1) == 0
        first = b.u(8)
        num = first
        if first & 0x80:
            n = 0
            while first & (0x80 >> n):
                n += 1
            num = first & (0x7F >> n)
            for _ in range(n - 1):
                c = b.u(8)
                assert c >> 6 == 2
                num = (num << 6) | (c & 0x3F)
        assert num == fno, (num, fno)
        if bsc == 1:
            bs = 192
        elif 2 <= bsc <= 5:
            bs = 576 << (bsc - 2)
        elif bsc == 6:
            bs = b.u(8) + 1
        elif bsc == 7:
            bs = b.u(16) + 1
        else:
            bs = 256 << (bsc - 8)
        assert src <= 11  # codes 12-14 carry extra header bytes (not needed by any vector here), 15 is invalid
        hdr_end = b.pos // 8
        assert b.u(8) == crc8(data[start:hdr_end]), "CRC-8"
        bps0 = {0: info["bps"], 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}[ssc]

        def subframe(bps):
            assert b.u(1) == 0
            tc = b.u(6)
            wasted = b.unary() + 1 if b.u(1) else 0
            bps -= wasted
            desc = {"bs": bs, "wasted": wasted, "offset": start}
            if tc == 0:
                x = [b.s(bps)] * bs
                desc["type"] = "const"
            elif tc == 1:
                x = [b.s(bps) for _ in range(bs)]
                desc["type"] = "verbatim"
            else:
                if 8 <= tc <= 12:
                    order = tc - 8
                    desc["type"] = "fixed"
                    x = [b.s(bps) for _ in range(order)]
                    coefs, shift = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order], 0
                else:
                    assert tc >= 32
                    order = (tc & 31) + 1
                    desc["type"] = "lpc"
                    x = [b.s(bps) for _ in range(order)]
                    prec = b.u(4) + 1
                    assert prec != 16
                    shift = b.s(5)
                    assert shift >= 0
                    coefs = [b.s(prec) for _ in range(order)]
                    desc["precision"], desc["shift"] = prec, shift
                desc["order"] = order
                method = b.u(2)
                assert method < 2
                po = b.u(4)
                plen, esc = (5, 31) if method else (4, 15)
                desc["porder"], desc["rice2"], desc["params"] = po, bool(method), []
                res = []
                for p in range(1 << po):
                    n = (bs >> po) - (order if p == 0 else 0)
                    k = b.u(plen)
                    if k == esc:
                        w = b.u(5)
                        desc["params"].append(("esc", w))
                        res += [b.s(w) for _ in range(n)]
                    else:
                        desc["params"].append(k)
                        for _ in range(n):
                            q = b.unary()
                            u = (q << k) | b.u(k)
                            res.append((u >> 1) ^ -(u & 1))
                for i in range(order, bs):
                    pred = sum(c * x[i - 1 - j] for j, c in enumerate(coefs)) >> shift
                    x.append(res[i - order] + pred)
            return [v << wasted for v in x], desc

        if info["channels"] == 1:
            assert ch == 0
            x, desc = subframe(bps0)
        else:
            # two channels, sample-interleaved output: left/right, left/side, side/right, mid/side
            assert info["channels"] == 2 and ch in (1, 8, 9, 10)
            c0, d0 = subframe(bps0 + (1 if ch == 9 else 0))
            c1, d1 = subframe(bps0 + (1 if ch in (8, 10) else 0))
            x = []
            for u0, u1 in zip(c0, c1):
                if ch == 1:
                    left, right = u0, u1
                elif ch == 8:
                    left, right = u0, u0 - u1
                elif ch == 9:
                    left, right = u0 + u1, u1
                else:
                    mid = (u0 << 1) | (u1 & 1)
                    left, right = (mid + u1) >> 1, (mid - u1) >> 1
                assert -(1 << (bps0 - 1)) <= left < (1 << (bps0 - 1)) and -(1 << (bps0 - 1)) <= right < (1 << (bps0 - 1))
                x += [left, right]
            desc = {"bs": bs, "offset": start, "assignment": ch, "subs": [d0, d1]}
        b.pos = (b.pos + 7) & ~7
        end = b.pos // 8
        assert b.u(16) == crc16(data[start:end]), "CRC-16"
        off = end + 2
        out += x
        frames.append(desc)
        fno += 1
    info["frames"] = frames
    return out, info
