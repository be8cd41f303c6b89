// decode_kernels.hpp -- HIP kernels of the FLAC decode path for gfx950 (wave64).
//
// Replaces what the reference reaches through libFLAC in decode()
// (src/flacarray/libflacarray/decompress.c:194-313): stream/metadata parsing, frame location
// (libFLAC's seek_absolute binary search, decompress.c:283), Rice/escape decoding, predictor
// restoration, slice trimming (dec_write_callback, decompress.c:66-101) and the optional
// int32 -> float32 restore (int32_to_float32, utils.c:350-368) fused into the store.
//
//   K6  parse_streams_kernel      one thread per stream: metadata blocks, STREAMINFO, SEEKTABLE
//       build_frame_table_kernel  frame byte offsets from the SEEKTABLE (one thread per frame)
//       scan_sync_kernel          streams without a complete SEEKTABLE (e.g. written by libFLAC
//       check_scan_kernel         through the reference): parallel sync-code scan, frames claim
//                                 their table entry by frame number
//       walk_frames_kernel        fallback for streams the scan found ambiguous: one thread
//                                 walks the stream's frames
//   K7  decode_frames_kernel      one LANE per frame (Rice decoding and LPC restoration are
//                                 serial recurrences inside a frame; frames are independent),
//                                 64 frames per wavefront, samples transposed through LDS so
//                                 that HBM stores are 128-byte row segments
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "flac_math.hpp"

namespace fa {

constexpr int kErrDecodeInit = 1 << 13;     // flacarray.h:33 ERROR_DECODE_INIT
constexpr int kErrDecodeProcess = 1 << 14;  // flacarray.h:34 ERROR_DECODE_PROCESS
constexpr int kErrDecodeSeek = 1 << 18;     // flacarray.h:38 ERROR_DECODE_SEEK

constexpr int64_t kFrameUnset = -2;          // frame table entry not found yet (sync scan)
constexpr int kScanSpan = 256 * 16;          // bytes one workgroup of the sync scan covers per pass

struct StreamMeta {
    int64_t first_frame;  // absolute byte offset of the first frame in the blob
    int64_t seek_abs;     // absolute byte offset of the first seek point, or -1
    int64_t end_abs;      // absolute end of the stream's bytes
    int32_t npoints;
    int32_t B;    // fixed blocksize from STREAMINFO
    int32_t bps;  // bits per sample from STREAMINFO
    int32_t flags;  // 1 = complete seek table
    int32_t channels;  // 1 (int32 / float32 arrays) or 2 (int64 / float64: low and high words)
};

__device__ __forceinline__ uint64_t load_be64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
    return v;
}

__global__ __launch_bounds__(256) void parse_streams_kernel(const uint8_t* __restrict__ blob, const int64_t* __restrict__ starts,
                                                            const int64_t* __restrict__ nbytes, int64_t n_stream,
                                                            int64_t stream_size, int64_t blob_bytes,
                                                            StreamMeta* __restrict__ meta, int* __restrict__ err) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_stream) return;
    const int64_t st0 = starts[s];
    const int64_t nb = nbytes[s];
    StreamMeta m;
    m.first_frame = -1; m.seek_abs = -1; m.end_abs = 0; m.npoints = 0; m.B = 0; m.bps = 0; m.flags = 0; m.channels = 0;
    // a damaged index (negative or out-of-range starts / nbytes, e.g. from a corrupt stream_starts dataset) must not
    // become an out-of-range read: nothing of the stream is touched before this test
    if (st0 < 0 || nb < 0 || st0 > blob_bytes || nb > blob_bytes - st0) {
        atomicOr(err, kErrDecodeInit);
        meta[s] = m;
        return;
    }
    const uint8_t* p = blob + st0;
    m.end_abs = st0 + nb;
    bool ok = nb >= 42 && p[0] == 'f' && p[1] == 'L' && p[2] == 'a' && p[3] == 'C';
    int64_t off = 4;
    while (ok) {
        if (off + 4 > nb) { ok = false; break; }
        const int last = p[off] >> 7, type = p[off] & 0x7f;
        const int64_t len = ((int64_t)p[off + 1] << 16) | ((int64_t)p[off + 2] << 8) | p[off + 3];
        off += 4;
        if (off + len > nb) { ok = false; break; }
        if (type == 0 && len >= 34) {
            const int minb = (p[off] << 8) | p[off + 1], maxb = (p[off + 2] << 8) | p[off + 3];
            m.B = maxb;
            if (minb != maxb) ok = false;  // variable-blocksize streams are never produced on this path
            m.bps = (((p[off + 12] & 1) << 4) | (p[off + 13] >> 4)) + 1;
            m.channels = ((p[off + 12] >> 1) & 7) + 1;
        } else if (type == 3) {
            m.seek_abs = starts[s] + off;
            m.npoints = (int32_t)(len / 18);
        }
        off += len;
        if (last) break;
    }
    if (ok && m.B > 0) {
        m.first_frame = starts[s] + off;
        const int64_t nf = (stream_size + m.B - 1) / m.B;
        if (m.seek_abs >= 0 && m.npoints == nf) m.flags = 1;
        if (!(m.flags & 1)) {
            atomicAdd(err + 2, 1);
            const int64_t span = (nb - off + kScanSpan - 1) / kScanSpan + 1;
            atomicMax(err + 3, (int)(span > 0x7fffffff ? 0x7fffffff : span));
        }
    } else {
        atomicOr(err, kErrDecodeInit);
    }
    meta[s] = m;
}

__global__ __launch_bounds__(256) void build_frame_table_kernel(const uint8_t* __restrict__ blob, const StreamMeta* __restrict__ meta,
                                                                int64_t n_stream, int64_t nf, int32_t B, int32_t nch,
                                                                int64_t* __restrict__ ftab, int* __restrict__ err) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_stream * nf) return;
    const int64_t s = t / nf, f = t - s * nf;
    const StreamMeta m = meta[s];
    if (m.first_frame < 0) { ftab[t] = -1; return; }
    if (m.B != B || m.channels != nch) { atomicOr(err, kErrDecodeInit); ftab[t] = -1; return; }
    if (!(m.flags & 1)) { ftab[t] = kFrameUnset; return; }
    const uint8_t* sp = blob + m.seek_abs + 18 * f;
    const uint64_t sn = load_be64(sp), off = load_be64(sp + 8);
    if (sn != (uint64_t)f * (uint64_t)B) { atomicOr(err, kErrDecodeSeek); ftab[t] = -1; return; }
    ftab[t] = m.first_frame + (int64_t)off;
}

// ------------------------------------------------------------------------------------------
// Per-lane MSB-first bit reader over global memory with a 16-byte register prefetch queue
// ------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t* base;  // readable range [base, lim)
    const uint8_t* lim;
    const uint8_t* wp;    // next 16-byte chunk (16-byte aligned address, may be outside the range)
    uint4 cur, nxt;
    int idx;              // next dword of cur
    uint64_t win;         // next bits, MSB first
    int nbits;            // valid bits in win
    int64_t used;         // bits consumed since init

    __device__ __forceinline__ uint4 load16(const uint8_t* q) const {
        if (q >= base && q + 16 <= lim) return *reinterpret_cast<const uint4*>(q);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; ++i)
            if (q + i >= base && q + i < lim) w[i >> 2] |= (uint32_t)q[i] << (8 * (i & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ __forceinline__ uint32_t next_dword() {
        uint32_t w = (idx == 0) ? cur.x : (idx == 1) ? cur.y : (idx == 2) ? cur.z : cur.w;
        idx++;
        if (idx == 4) {
            cur = nxt;
            nxt = load16(wp);
            wp += 16;
            idx = 0;
        }
        return __builtin_bswap32(w);
    }
    __device__ __forceinline__ void refill() {
        if (nbits <= 32) {
            win |= (uint64_t)next_dword() << (32 - nbits);
            nbits += 32;
        }
    }
    __device__ __forceinline__ void init(const uint8_t* b, const uint8_t* l, const uint8_t* start) {
        base = b; lim = l;
        const uintptr_t a = reinterpret_cast<uintptr_t>(start);
        const uint8_t* q = reinterpret_cast<const uint8_t*>(a & ~(uintptr_t)15);
        cur = load16(q);
        nxt = load16(q + 16);
        wp = q + 32;
        idx = (int)((a >> 2) & 3);
        win = 0; nbits = 0; used = 0;
        refill();
        const int drop = (int)(a & 3) * 8;
        win <<= drop;
        nbits -= drop;
        refill();
    }
    // n in 1..32
    __device__ __forceinline__ uint32_t get(int n) {
        const uint32_t v = (uint32_t)(win >> (64 - n));
        win <<= n;
        nbits -= n;
        used += n;
        refill();
        return v;
    }
    __device__ __forceinline__ int32_t get_signed(int n) {
        if (n == 0) return 0;
        const uint32_t v = get(n);
        return (int32_t)(v << (32 - n)) >> (32 - n);
    }
    // number of 0 bits before the next 1 bit; consumes the 1
    __device__ __forceinline__ uint32_t unary() {
        uint32_t q = 0;
        for (;;) {
            const uint32_t hi = (uint32_t)(win >> 32);
            if (hi) {
                const int z = __clz((int)hi);
                win <<= (z + 1);
                nbits -= (z + 1);
                used += (z + 1);
                refill();
                return q + (uint32_t)z;
            }
            q += 32;
            win <<= 32;
            nbits -= 32;
            used += 32;
            refill();
            if (q > (1u << 24)) return q;  // corrupt stream guard: every wave reaches an exit
        }
    }
    __device__ __forceinline__ void skip(int64_t n) {
        while (n > 0) {
            const int c = n > 32 ? 32 : (int)n;
            win <<= c; nbits -= c; used += c; n -= c;
            refill();
        }
    }
};

__device__ __forceinline__ bool channel_code_ok(int ch, int nch);

struct FrameHeader {
    int bs;
    int bps;
    int ok;
    int ch;  // channel assignment code
};

// frame header (RFC 9639 9.1); CRC-8 verified
__device__ __forceinline__ FrameHeader read_frame_header(BitReader& br, int si_bps, int nch) {
    FrameHeader h;
    h.ok = 0; h.bs = 0; h.bps = 0; h.ch = 0;
    uint8_t c8 = 0;
    const uint32_t b01 = br.get(16);
    c8 = crc8_byte(c8, (uint8_t)(b01 >> 8));
    c8 = crc8_byte(c8, (uint8_t)b01);
    if ((b01 & 0xFFFE) != 0xFFF8) return h;
    if (b01 & 1) return h;  // variable blocksize
    const uint32_t b2 = br.get(8), b3 = br.get(8);
    c8 = crc8_byte(c8, (uint8_t)b2);
    c8 = crc8_byte(c8, (uint8_t)b3);
    const int bsc = (int)(b2 >> 4), src = (int)(b2 & 15), ch = (int)(b3 >> 4), ssc = (int)((b3 >> 1) & 7);
    if (!channel_code_ok(ch, nch) || (b3 & 1)) return h;
    h.ch = ch;
    const uint32_t u0 = br.get(8);
    c8 = crc8_byte(c8, (uint8_t)u0);
    int extra = 0;
    if (u0 & 0x80) {
        int mbit = 0x40;
        while ((u0 & mbit) && extra < 7) { extra++; mbit >>= 1; }
        if (extra == 0 || extra > 6) return h;
    }
    for (int i = 0; i < extra; ++i) c8 = crc8_byte(c8, (uint8_t)br.get(8));
    int bs;
    if (bsc == 0) return h;
    else if (bsc == 1) bs = 192;
    else if (bsc <= 5) bs = 576 << (bsc - 2);
    else if (bsc == 6) { const uint32_t v = br.get(8); c8 = crc8_byte(c8, (uint8_t)v); bs = (int)v + 1; }
    else if (bsc == 7) { const uint32_t v = br.get(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); bs = (int)v + 1; }
    else bs = 256 << (bsc - 8);
    if (src == 12) { c8 = crc8_byte(c8, (uint8_t)br.get(8)); }
    else if (src == 13 || src == 14) { const uint32_t v = br.get(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); }
    else if (src == 15) return h;
    const uint32_t got = br.get(8);
    if (got != c8) return h;
    int bps;
    switch (ssc) {
        case 0: bps = si_bps; break;
        case 1: bps = 8; break;
        case 2: bps = 12; break;
        case 4: bps = 16; break;
        case 5: bps = 20; break;
        case 6: bps = 24; break;
        case 7: bps = 32; break;
        default: return h;
    }
    h.bs = bs; h.bps = bps; h.ok = 1;
    return h;
}

// ------------------------------------------------------------------------------------------
// K6s: parallel frame location for streams without a seek table (what libFLAC writes through the
// reference: STREAMINFO + VORBIS_COMMENT, then frames).  Every byte position that holds the
// fixed-blocksize sync code 0xFF 0xF8 followed by a header that (a) parses, (b) has a valid CRC-8,
// (c) is mono with the stream's block size and (d) carries a frame number n < nf claims entry n of
// the stream's frame table with a compare-and-swap.  Every true frame claims its own entry, so a
// false candidate can only ever collide with a true one: a collision, a missing entry or a
// non-increasing table marks the stream "ambiguous" and only such streams are walked serially.
// ------------------------------------------------------------------------------------------
struct HeaderBytes {
    int ok;
    int bs;
    uint64_t num;
};

// frame header from a byte range [p, p + avail) (RFC 9639 9.1); fixed blocksize, mono or stereo
__device__ __forceinline__ bool channel_code_ok(int ch, int nch) {
    return (nch == 1) ? (ch == 0) : (ch == 1 || (ch >= 8 && ch <= 10));
}
__device__ __noinline__ HeaderBytes parse_header_bytes(const uint8_t* p, int64_t avail, int nch) {
    HeaderBytes h;
    h.ok = 0; h.bs = 0; h.num = 0;
    if (avail < 6) return h;
    if (p[0] != 0xFF || p[1] != 0xF8) return h;
    const int bsc = p[2] >> 4, src = p[2] & 15, ch = p[3] >> 4, ssc = (p[3] >> 1) & 7;
    if (bsc == 0 || src == 15 || !channel_code_ok(ch, nch) || (p[3] & 1) || ssc == 3) return h;
    int n = 4;
    const uint32_t u0 = p[n++];
    int extra = 0;
    uint64_t num = u0;
    if (u0 & 0x80) {
        int mbit = 0x40;
        while ((u0 & mbit) && extra < 7) { extra++; mbit >>= 1; }
        if (extra == 0 || extra > 5) return h;  // a frame number has at most 31 bits (6 bytes)
        num = u0 & (uint32_t)(mbit - 1);
    }
    const int need = n + extra + (bsc == 6 ? 1 : bsc == 7 ? 2 : 0) + (src == 12 ? 1 : (src == 13 || src == 14) ? 2 : 0) + 1;
    if (need > avail) return h;
    for (int i = 0; i < extra; ++i) {
        const uint32_t c = p[n++];
        if ((c & 0xC0) != 0x80) return h;
        num = (num << 6) | (c & 0x3F);
    }
    int bs;
    if (bsc == 1) bs = 192;
    else if (bsc <= 5) bs = 576 << (bsc - 2);
    else if (bsc == 6) { bs = (int)p[n] + 1; n += 1; }
    else if (bsc == 7) { bs = (((int)p[n] << 8) | (int)p[n + 1]) + 1; n += 2; }
    else bs = 256 << (bsc - 8);
    if (src == 12) n += 1;
    else if (src == 13 || src == 14) n += 2;
    uint8_t c8 = 0;
    for (int i = 0; i < n; ++i) c8 = crc8_byte(c8, p[i]);
    if (c8 != p[n]) return h;
    h.ok = 1; h.bs = bs; h.num = num;
    return h;
}

// grid: x = stream, y = pass over the stream's bytes in spans of kScanSpan (grid-stride in y)
__global__ __launch_bounds__(256) void scan_sync_kernel(const uint8_t* __restrict__ blob, int64_t blob_bytes,
                                                        const StreamMeta* __restrict__ meta, int64_t nf, int32_t B,
                                                        int64_t stream_size, int64_t* __restrict__ ftab,
                                                        int* __restrict__ sflag) {
    const int64_t s = blockIdx.x;
    const StreamMeta m = meta[s];
    if (m.first_frame < 0 || (m.flags & 1) || m.B != B || m.channels < 1 || m.channels > 2) return;
    const int64_t lo = m.first_frame, hi = m.end_abs;
    const int tail_bs = (int)(stream_size - (nf - 1) * (int64_t)B);
    // 16-byte groups on the blob's own 16-byte grid (the blob base is 16-byte aligned)
    const int64_t g_lo = lo & ~(int64_t)15;
    for (int64_t g = g_lo + 16 * ((int64_t)blockIdx.y * 256 + threadIdx.x); g < hi; g += (int64_t)kScanSpan * gridDim.y) {
        uint32_t w[4];
        if (g + 16 <= blob_bytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(blob + g);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            w[0] = w[1] = w[2] = w[3] = 0;
            for (int i = 0; i < 16; ++i)
                if (g + i < blob_bytes) w[i >> 2] |= (uint32_t)blob[g + i] << (8 * (i & 3));
        }
        // bytes equal to 0xFF (zero bytes of ~w; the test may flag a byte above a true hit, which
        // the exact comparison below discards)
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = ~w[j];
            any |= (t - 0x01010101u) & ~t & 0x80808080u;
        }
        if (!any) continue;
        for (int i = 0; i < 16; ++i) {
            if (((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) != 0xFFu) continue;
            const int64_t pos = g + i;
            if (pos < lo || pos + 6 > hi) continue;
            if (blob[pos + 1] != 0xF8) continue;
            const HeaderBytes h = parse_header_bytes(blob + pos, hi - pos, m.channels);
            if (!h.ok || h.num >= (uint64_t)nf) continue;
            if (h.bs != ((h.num == (uint64_t)(nf - 1)) ? tail_bs : B)) continue;
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&ftab[s * nf + (int64_t)h.num]),
                                                     (unsigned long long)kFrameUnset, (unsigned long long)pos);
            if (old != (unsigned long long)kFrameUnset) sflag[s] = 1;
        }
    }
}

// one thread per (stream, frame): the scanned table must be complete, start at the first frame
// and increase strictly
__global__ __launch_bounds__(256) void check_scan_kernel(const StreamMeta* __restrict__ meta, int64_t n_stream, int64_t nf, int32_t B,
                                                         const int64_t* __restrict__ ftab, int* __restrict__ sflag) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_stream * nf) return;
    const int64_t s = t / nf, f = t - s * nf;
    const StreamMeta m = meta[s];
    if (m.first_frame < 0 || (m.flags & 1) || m.B != B) return;
    const int64_t v = ftab[t];
    const bool bad = (v < 0) || ((f == 0) ? (v != m.first_frame) : (v <= ftab[t - 1]));
    if (bad) sflag[s] = 1;
}

// Walk every frame of streams that carry no complete seek table.  One thread per stream; the
// walk parses but does not reconstruct (no prediction).
__global__ __launch_bounds__(64) void walk_frames_kernel(const uint8_t* __restrict__ blob, int64_t blob_bytes,
                                                         const StreamMeta* __restrict__ meta, int64_t n_stream, int64_t nf,
                                                         int32_t B, int64_t stream_size, int64_t* __restrict__ ftab,
                                                         const int* __restrict__ sflag, int* __restrict__ err) {
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= n_stream) return;
    const StreamMeta m = meta[s];
    if (m.first_frame < 0 || (m.flags & 1) || m.B != B || m.channels < 1 || m.channels > 2) return;
    if (sflag && !sflag[s]) return;  // the sync scan located every frame
    int64_t at = m.first_frame;
    for (int64_t f = 0; f < nf; ++f) {
        ftab[s * nf + f] = at;
        if (at >= m.end_abs) { atomicOr(err, kErrDecodeProcess); for (int64_t r = f; r < nf; ++r) ftab[s * nf + r] = -1; return; }
        BitReader br;
        br.init(blob, blob + blob_bytes, blob + at);
        FrameHeader h = read_frame_header(br, m.bps, m.channels);
        bool ok = h.ok;
        for (int chn = 0; ok && chn < m.channels; ++chn) {
            // a side channel (left/side and mid/side: channel 1; side/right: channel 0) has one more bit
            const int fbps = h.bps + (((chn == 1 && (h.ch == 8 || h.ch == 10)) || (chn == 0 && h.ch == 9)) ? 1 : 0);
            const uint32_t sf = br.get(8);
            int tc = (int)((sf >> 1) & 0x3f);
            int wasted = 0;
            if (sf & 0x80) ok = false;
            if (sf & 1) wasted = (int)br.unary() + 1;
            const int bps = fbps - wasted;
            int order = 0;
            bool pred = false;
            if (bps <= 0) ok = false;
            else if (tc == 0) br.skip(bps);
            else if (tc == 1) br.skip((int64_t)bps * h.bs);
            else if (tc >= 8 && tc <= 12) { order = tc - 8; pred = true; br.skip((int64_t)bps * order); }
            else if (tc >= 32) {
                order = (tc & 31) + 1; pred = true;
                br.skip((int64_t)bps * order);
                const int prec = (int)br.get(4) + 1;
                br.get(5);
                br.skip((int64_t)prec * order);
            } else ok = false;
            if (ok && pred) {
                const int method = (int)br.get(2);
                const int po = (int)br.get(4);
                const int plen = method ? 5 : 4, esc = method ? 31 : 15;
                if (method > 1 || order > h.bs) ok = false;
                const int ps = h.bs >> po;
                for (int p = 0; ok && p < (1 << po); ++p) {
                    const int n = (p == 0) ? ps - order : ps;
                    if (n < 0) { ok = false; break; }
                    const int k = (int)br.get(plen);
                    if (k == esc) {
                        const int nbw = (int)br.get(5);
                        br.skip((int64_t)nbw * n);
                    } else {
                        for (int j = 0; j < n; ++j) {
                            const uint32_t q = br.unary();
                            if (q > (1u << 24)) { ok = false; break; }
                            if (k) br.get(k);
                        }
                    }
                }
            }
        }
        if (!ok) { atomicOr(err, kErrDecodeProcess); for (int64_t r = f + 1; r < nf; ++r) ftab[s * nf + r] = -1; return; }
        const int64_t bits = (br.used + 7) & ~(int64_t)7;
        at += (bits >> 3) + 2;
    }
    (void)stream_size;
}

// ------------------------------------------------------------------------------------------
// K7: decode.  One lane per task = (output row, frame).  Two task layouts:
//   grid mode  : task t -> stream t / nfr, frame f0 + t % nfr, one [first,last) range for all
//   list mode  : explicit arrays (scattered slices)
// ------------------------------------------------------------------------------------------
struct DecodeArgs {
    const uint8_t* blob;
    int64_t blob_bytes;
    const StreamMeta* meta;
    const int64_t* ftab;  // [n_stream][nf] absolute frame offsets
    int64_t nf;           // frames per stream
    int32_t B;
    int64_t stream_size;
    int64_t n_tasks;
    // grid mode
    int64_t nfr;       // frames per row in the requested range
    int64_t f0;        // first frame of the range
    int64_t first;     // first decoded sample
    int64_t n_decode;  // samples per output row
    // list mode (task_stream != null)
    const int64_t* task_stream;
    const int64_t* task_frame;
    const int64_t* task_first;    // slice first sample (stream coordinates)
    const int64_t* task_last;     // slice end (exclusive)
    const int64_t* task_out_off;  // output element offset of the slice's first sample
    // outputs
    int32_t* out_i32;
    float* out_f32;         // when non-null: dequantised output instead of out_i32
    const float* offsets;   // per stream
    const float* gains;
    int* err;
    // two-channel arrays (NCH == 2): out_i32 is then the task-local planar image
    uint32_t* hibits;   // [n_tasks][2][hib_words] bit 32 of every sample
    int32_t* assign;    // [n_tasks] channel assignment once both subframes are decoded, else -1
    int32_t hib_words;  // ceil(B / 32)
    int32_t verbatim_done;  // NCH == 2: VERBATIM first subframes (no wasted bits) were decoded by verbatim_channel0_kernel
};

// ------------------------------------------------------------------------------------------
// K7 (v2).  One wavefront = 64 frames, one LANE per frame, all lanes in lockstep on the sample
// index.  LDS per wave (64-thread workgroup, ~15.5 KB -> 10 waves per CU):
//   ring  64 x 36 words  per-lane input window: two 64-byte chunks of the frame's bytes stored
//                        as big-endian words (+2 mirror words so a 3-word read never wraps);
//                        topped up for ALL lanes together every 16 samples (a lane consumes at
//                        most 64 bytes in 16 fast-path samples), so the per-sample code has no
//                        refill branch: it re-reads its 96-bit window at `bitpos` from LDS
//   tile  64 x 20 words  decoded samples, transposed so that HBM stores are 64-byte row pieces
//   rows  (offset, 1/gain) of each lane's frame for the float32 instantiation; output offsets and valid
//                        ranges stay in registers and travel by ds_bpermute
// Rare paths (headers, warm-up, partition parameters, escapes, codes longer than 32 bits) go
// through two out-of-line helpers that take and return the reader state by value, so the hot
// loop keeps `bitpos` in a register.
// ------------------------------------------------------------------------------------------
template <int... U, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, U...>, F&& f) {
    (f(std::integral_constant<int, U>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#ifndef FA_TILE_W
#define FA_TILE_W 32
#endif
#ifndef FA_K7_SPEC
#define FA_K7_SPEC 0  // experiment (r04, profiles/r04_k7_notes.md): the window's ring words stay in registers and the one word a
                      // sample can newly need is requested a sample ahead -- the LDS round trip leaves the loop-carried chain,
                      // two LDS reads per sample go, five vector instructions come: 13 % SLOWER (4.07 against 3.59 ms)
#endif
#ifndef FA_K7_SEED
#define FA_K7_SEED 0
#endif
#ifndef FA_K7_NT
#define FA_K7_NT 1  // whole-tile stores bypass the caches (streaming output): -1.5 % in the same-box A/B (r03i)
#endif
#ifndef FA_K7_X
#define FA_K7_X 0  // timing experiments only (results are wrong): 1 = no prediction, 2 = no global stores, 4 = no tile write,
                   // 8 = no chunk loads in the sample loop
#endif
constexpr int kTileW = FA_TILE_W;                // samples per lane between two cooperative stores
constexpr int kTileG = kTileW / 4;               // 16-byte groups per row
constexpr int kTileSwz = 32 / kTileG;            // XOR swizzle step: 32 lanes of a store pass hit 32 banks
#ifndef FA_CHUNK_BYTES
#define FA_CHUNK_BYTES 64
#endif
constexpr int kChunkBytes = FA_CHUNK_BYTES;           // input is staged per lane in chunks of this size (two resident)
constexpr int kChunkW = kChunkBytes / 4;              // words per chunk
constexpr int kChunkShift = (kChunkBytes == 64) ? 9 : 8;  // log2(bits per chunk)
constexpr int kRingW = 2 * kChunkW;                   // ring words without the mirror
constexpr int kDecRingWords = kRingW + 2;             // + 2 mirror words, per lane
constexpr int kTopup = kChunkBytes / 8;               // samples between top-ups (<= 4 bytes per fast sample)
constexpr int kLaneStride = 64;  // word j of lane l lives at j*64 + l: the bank depends on the lane only

constexpr int kFlagNeed16 = 1, kFlagNeed32 = 2;
__device__ __forceinline__ int4 shl4(const int4& v, int n) {
    return make_int4((int)((uint32_t)v.x << n), (int)((uint32_t)v.y << n), (int)((uint32_t)v.z << n), (int)((uint32_t)v.w << n));
}

__device__ __forceinline__ void ring_load_chunk(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t ci) {
    const uint8_t* q = cbase + (size_t)ci * kChunkBytes;
    uint32_t* dst = ring + (ci & 1) * kChunkW * kLaneStride;
#pragma unroll
    for (int v = 0; v < kChunkBytes / 16; ++v) {
        uint4 d = make_uint4(0, 0, 0, 0);
        if (q + 16 * v + 16 <= lim16) d = *reinterpret_cast<const uint4*>(q + 16 * v);
        d.x = __builtin_bswap32(d.x); d.y = __builtin_bswap32(d.y); d.z = __builtin_bswap32(d.z); d.w = __builtin_bswap32(d.w);
        dst[(4 * v + 0) * kLaneStride] = d.x; dst[(4 * v + 1) * kLaneStride] = d.y;
        dst[(4 * v + 2) * kLaneStride] = d.z; dst[(4 * v + 3) * kLaneStride] = d.w;
        if (v == 0 && (ci & 1) == 0) { ring[kRingW * kLaneStride] = d.x; ring[(kRingW + 1) * kLaneStride] = d.y; }  // mirror of words 0,1
    }
}
// the same in two halves, so that the global loads can be issued one chunk ahead of their use
struct Chunk {
    uint4 d[kChunkBytes / 16];
};
// (The four loads are conditional loads into zero-initialised registers, which makes the compiler wait for the
// previous load before each of them.  That serialisation is worth keeping: the four pieces share a cache line, the
// first load brings it into L1 and the other three hit; issued back to back -- unconditional loads from a clamped
// address -- all four miss and K7 is 12 % slower, profiles/r03_k7_experiments.md.)
__device__ __forceinline__ Chunk chunk_fetch(const uint8_t* cbase, const uint8_t* lim16, uint32_t ci) {
    const uint8_t* q = cbase + (size_t)ci * kChunkBytes;
    Chunk c;
#pragma unroll
    for (int v = 0; v < kChunkBytes / 16; ++v) {
        c.d[v] = make_uint4(0, 0, 0, 0);
#if (FA_K7_X & 8)  // timing experiment: no chunk loads in the sample loop (a bit pattern of short codes instead)
        c.d[v] = make_uint4(0x55555555u ^ ci, 0x55555555u, 0x55555555u, 0x55555555u);
#else
        if (q + 16 * v + 16 <= lim16) c.d[v] = *reinterpret_cast<const uint4*>(q + 16 * v);
#endif
    }
    return c;
}
__device__ __forceinline__ void chunk_store(uint32_t* ring, uint32_t ci, const Chunk& c) {
    uint32_t* dst = ring + (ci & 1) * kChunkW * kLaneStride;
#pragma unroll
    for (int v = 0; v < kChunkBytes / 16; ++v) {
        uint4 d = c.d[v];
        d.x = __builtin_bswap32(d.x); d.y = __builtin_bswap32(d.y); d.z = __builtin_bswap32(d.z); d.w = __builtin_bswap32(d.w);
        dst[(4 * v + 0) * kLaneStride] = d.x; dst[(4 * v + 1) * kLaneStride] = d.y;
        dst[(4 * v + 2) * kLaneStride] = d.z; dst[(4 * v + 3) * kLaneStride] = d.w;
        if (v == 0 && (ci & 1) == 0) { ring[kRingW * kLaneStride] = d.x; ring[(kRingW + 1) * kLaneStride] = d.y; }
    }
}
// the three ring words starting at the word that holds bit (bitpos - 1): with off' = ((bitpos-1) & 31) + 1
// in 1..32 the window is ((w0:w1:w2) << off'), i.e. v_alignbit_b32 with a shift of 32 - off' in 0..31
// (bitpos >= 8 always: the frame starts at least one byte after its chunk base)
// `lane4` is the LDS byte address of the lane's ring word 0 (the ring image starts at a multiple of 64 x kRingW x 4
// bytes, checked by the kernel): the address of word wi is lane4 | (wi << 8), two instructions from bitpos -- the
// compiler's own sequence needs three, because a VOP3 instruction takes no literal on gfx9 and it does not think of
// keeping the mask in a scalar register.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
constexpr uint32_t kRingAddrMask = (uint32_t)(kRingW - 1) << 8;
__device__ __forceinline__ void ring_words(uint32_t lane4, uint32_t bitpos, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    static_assert(kLaneStride == 64, "word wi of a lane lies 256 wi bytes behind its word 0");
    uint32_t t, ad;
    asm("v_add_lshl_u32 %0, %2, -1, 3\n\tv_and_or_b32 %1, %0, %3, %4" : "=&v"(t), "=v"(ad) : "v"(bitpos), "s"(kRingAddrMask), "v"(lane4));
    const lds_u32* p = (const lds_u32*)(uintptr_t)ad;
    w0 = p[0]; w1 = p[kLaneStride]; w2 = p[2 * kLaneStride];
}
// the ring word behind the three that ring_words returns for `bitpos` (its own wrap: no mirror word needed)
__device__ __forceinline__ uint32_t ring_word3(uint32_t lane4, uint32_t bitpos, uint32_t k95) {
    uint32_t t, ad;
    asm("v_add_lshl_u32 %0, %2, %5, 3\n\tv_and_or_b32 %1, %0, %3, %4" : "=&v"(t), "=v"(ad) : "v"(bitpos), "s"(kRingAddrMask), "v"(lane4), "s"(k95));
    return *(const lds_u32*)(uintptr_t)ad;
}
// bits [bitpos, bitpos+32) -> A and [bitpos+32, bitpos+64) -> B
__device__ __forceinline__ void ring_window(const uint32_t* ring, uint32_t bitpos, uint32_t& A, uint32_t& B) {
    const uint32_t wi = (bitpos >> 5) & (kRingW - 1);
    const uint32_t off = bitpos & 31;
    uint32_t w0 = ring[wi * kLaneStride], w1 = ring[(wi + 1) * kLaneStride], w2 = ring[(wi + 2) * kLaneStride];
    // keep the three reads together (one LDS round trip): without this the compiler sinks the
    // third read into the fast-path block, behind a second wait
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
    A = (uint32_t)((((((uint64_t)w0) << 32) | w1) << off) >> 32);
    B = (uint32_t)((((((uint64_t)w1) << 32) | w2) << off) >> 32);
}

struct BitsRet {
    uint32_t val, bitpos, next_chunk;
    uint32_t bad;  // a unary run longer than any sane code: corrupt or truncated stream
};
constexpr uint32_t kUnaryLimit = 1u << 20;
// read n (0..32) bits
__device__ __noinline__ BitsRet slow_get(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t bitpos,
                                         uint32_t next_chunk, int n) {
    BitsRet r;
    r.val = 0;
    r.bad = 0;
    if (n > 0) {
        while (((bitpos + 64) >> kChunkShift) >= next_chunk) { ring_load_chunk(cbase, lim16, ring, next_chunk); next_chunk++; }
        uint32_t A, B;
        ring_window(ring, bitpos, A, B);
        r.val = A >> (32 - n);
        bitpos += (uint32_t)n;
    }
    r.bitpos = bitpos;
    r.next_chunk = next_chunk;
    return r;
}
// count 0 bits up to and including the next 1 bit; val = number of zeros
__device__ __noinline__ BitsRet slow_unary(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t bitpos,
                                           uint32_t next_chunk) {
    BitsRet r;
    uint32_t q = 0;
    for (;;) {
        while (((bitpos + 64) >> kChunkShift) >= next_chunk) { ring_load_chunk(cbase, lim16, ring, next_chunk); next_chunk++; }
        uint32_t A, B;
        ring_window(ring, bitpos, A, B);
        if (A) {
            const int z = __clz((int)A);
            bitpos += (uint32_t)z + 1;
            q += (uint32_t)z;
            break;
        }
        q += 32;
        bitpos += 32;
        if (q > kUnaryLimit) break;  // corrupt stream guard: every lane reaches an exit
    }
    r.bad = (q > kUnaryLimit) ? 1u : 0u;
    r.val = q;
    r.bitpos = bitpos;
    r.next_chunk = next_chunk;
    return r;
}

// make chunks [chunk(bitpos), chunk(bitpos)+1] resident (synchronous loads; rare)
__device__ __forceinline__ uint32_t ring_ensure(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t bitpos,
                                                uint32_t next_chunk) {
    while ((bitpos >> kChunkShift) + 1 >= next_chunk) { ring_load_chunk(cbase, lim16, ring, next_chunk); next_chunk++; }
    return next_chunk;
}
struct ParamRet {
    uint32_t k, escw, bitpos, next_chunk;
};
// partition header: Rice parameter (plen bits), escape width if k == esc (escw = 0xffffffff: none)
__device__ __noinline__ ParamRet slow_param(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t bitpos,
                                            uint32_t next_chunk, int plen, int esc) {
    ParamRet o;
    BitsRet r = slow_get(cbase, lim16, ring, bitpos, next_chunk, plen);
    o.k = r.val;
    o.escw = 0xffffffffu;
    if ((int)r.val == esc) {
        r = slow_get(cbase, lim16, ring, r.bitpos, r.next_chunk, 5);
        o.escw = r.val;
    }
    o.bitpos = r.bitpos;
    o.next_chunk = ring_ensure(cbase, lim16, ring, r.bitpos, r.next_chunk);
    return o;
}
// one residual that the fast path cannot take: escaped (escw > 0) or a Rice code longer than 32 bits
__device__ __noinline__ BitsRet slow_sample(const uint8_t* cbase, const uint8_t* lim16, uint32_t* ring, uint32_t bitpos,
                                            uint32_t next_chunk, int k, int escw) {
    BitsRet r;
    if (escw > 0) {
        r = slow_get(cbase, lim16, ring, bitpos, next_chunk, escw);
        r.val = (uint32_t)((int32_t)(r.val << (32 - escw)) >> (32 - escw));
    } else {
        const BitsRet q = slow_unary(cbase, lim16, ring, bitpos, next_chunk);
        r = slow_get(cbase, lim16, ring, q.bitpos, q.next_chunk, k);
        const uint32_t uu = (q.val << k) | r.val;
        r.val = (uint32_t)((int32_t)(uu >> 1) ^ -(int32_t)(uu & 1));
        r.bad = q.bad;
    }
    r.next_chunk = ring_ensure(cbase, lim16, ring, r.bitpos, r.next_chunk);
    return r;
}

#define FA_GET(n) ({ const BitsRet r_ = slow_get(cbase, lim16, ring, bitpos, next_chunk, (n)); bitpos = r_.bitpos; next_chunk = r_.next_chunk; r_.val; })
#define FA_GETS(n) ({ const int n_ = (n); const uint32_t v_ = FA_GET(n_); (n_ == 0) ? 0 : ((int32_t)(v_ << (32 - n_)) >> (32 - n_)); })
#define FA_UNARY() ({ const BitsRet r_ = slow_unary(cbase, lim16, ring, bitpos, next_chunk); bitpos = r_.bitpos; next_chunk = r_.next_chunk; if (r_.bad) bad = true; r_.val; })

// MO = history depth of this variant (8, 16 or 32).  Tasks whose predictor order exceeds MO are
// left for a later pass (flag word); tasks with order <= MO_DONE were done by an earlier one.
// F32: dequantised float32 output (needs per-row offset / 1/gain descriptors in LDS); the int32
// variant leaves them out, which is what lets a ninth wave fit on a CU.
// NCH = 2: two-channel frames (the reference's int64 / float64 arrays: low and high words).  A lane
// decodes its frame's two subframes one after the other into a task-local planar image
// tmp[(task * 2 + channel) * B + sample] (low 32 bits) plus one bit per sample for bit 32 (side
// channels carry 33 bits); combine_channels_kernel undoes the stereo decorrelation and writes int64.
template <int MO, int MO_DONE, bool F32, int NCH>
#ifndef FA_K7_WAVES_ATTR
#define FA_K7_WAVES_ATTR
#endif
__global__ __launch_bounds__(64) FA_K7_WAVES_ATTR void decode_frames_kernel(DecodeArgs a, int* flags) {
    // tile width of this instantiation (shadows the namespace defaults): FA_TILE_W applies to the one-channel
    // variants with a history of at most 16; the others need 32-sample tiles
    constexpr int kTileW = (NCH == 1 && MO <= 16) ? FA_TILE_W : 32;
    constexpr int kTileG = kTileW / 4;
    constexpr int kTileSwz = 32 / kTileG;
    static_assert(NCH == 1 || (kTileW == 32 && !F32), "two-channel variant: 32-sample tiles, integer output");
    // one LDS object, so that the ring image starts at LDS address 0 (ring_words builds its addresses with an OR)
    __shared__ __attribute__((aligned(16))) uint32_t lds_k7[kDecRingWords * kLaneStride + kTileW * kLaneStride + (F32 ? 128 : 0)];
    uint32_t* const rings = lds_k7;
    int32_t* const tile = reinterpret_cast<int32_t*>(lds_k7 + kDecRingWords * kLaneStride);  // sample t of lane l at t*64 + (l ^ 8*(t>>2))
    float2* const row_fg = reinterpret_cast<float2*>(lds_k7 + kDecRingWords * kLaneStride + kTileW * kLaneStride);
    const int lane = threadIdx.x;
    const int64_t task = (int64_t)blockIdx.x * 64 + lane;
    const bool has_task = task < a.n_tasks;
    uint32_t* const ring = rings + lane;
    const uint32_t lane4 = (uint32_t)(uintptr_t)(lds_u32*)ring;
    if ((uint32_t)(uintptr_t)(lds_u32*)rings & (kRingAddrMask | 255u)) {  // (never: `rings` is the first LDS object of the kernel)
        if (lane == 0) atomicOr(a.err, kErrDecodeInit);
        return;
    }

    // ---- per-lane task setup ----
    int64_t s = 0, f = 0, sl_first = 0, sl_last = 0, out_off = 0;
    if (has_task) {
        if (a.task_stream) {
            s = a.task_stream[task]; f = a.task_frame[task];
            sl_first = a.task_first[task]; sl_last = a.task_last[task];
            out_off = a.task_out_off[task];
        } else {
            s = task / a.nfr; f = a.f0 + (task - s * a.nfr);
            sl_first = a.first; sl_last = a.first + a.n_decode;
            out_off = s * a.n_decode;
        }
    }
    const int64_t fstart = f * (int64_t)a.B;
    // reader state
    const uint8_t* cbase = a.blob;
    const uint8_t* const lim16 = reinterpret_cast<const uint8_t*>((reinterpret_cast<uintptr_t>(a.blob + a.blob_bytes) + 15) & ~(uintptr_t)15);
    uint32_t bitpos = 0, next_chunk = 0x00400000u;  // idle lanes never refill

    // ---- frame header (RFC 9639 9.1), CRC-8 verified ----
    bool bad = false;
    bool task_live = has_task;  // false: nothing (more) to decode for this lane
    int fbs = 0, fbps = 0, ch_assign = 0, si_bps = 0;
    if (has_task) {
        const StreamMeta m = a.meta[s];
        si_bps = m.bps;
        const int64_t at = a.ftab[s * a.nf + f];
        if (m.first_frame < 0 || at < 0) bad = true;
        else {
            const uint8_t* start = a.blob + at;
            cbase = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(start - 1) & ~(uintptr_t)(kChunkBytes - 1));  // bitpos >= 8
            if (cbase < a.blob) cbase = a.blob;  // the blob base is 16-byte aligned (host side guarantees it)
            bitpos = (uint32_t)(start - cbase) * 8;
            ring_load_chunk(cbase, lim16, ring, 0);
            ring_load_chunk(cbase, lim16, ring, 1);
            next_chunk = 2;
            {
                uint8_t c8 = 0;
                const uint32_t w = FA_GET(32);
                c8 = crc8_byte(c8, (uint8_t)(w >> 24));
                c8 = crc8_byte(c8, (uint8_t)(w >> 16));
                c8 = crc8_byte(c8, (uint8_t)(w >> 8));
                c8 = crc8_byte(c8, (uint8_t)w);
                if ((w >> 16) != 0xFFF8) bad = true;  // sync, reserved 0, fixed blocksize
                const uint32_t b2 = (w >> 8) & 0xff, b3 = w & 0xff;
                const int bsc = (int)(b2 >> 4), src = (int)(b2 & 15), ch = (int)(b3 >> 4), ssc = (int)((b3 >> 1) & 7);
                if (!channel_code_ok(ch, NCH) || (b3 & 1)) bad = true;
                ch_assign = ch;
                const uint32_t u0 = FA_GET(8);
                c8 = crc8_byte(c8, (uint8_t)u0);
                int extra = 0;
                if (u0 & 0x80) {
                    int mbit = 0x40;
                    while ((u0 & mbit) && extra < 7) { extra++; mbit >>= 1; }
                    if (extra == 0 || extra > 6) bad = true;
                }
                for (int i = 0; i < extra && !bad; ++i) c8 = crc8_byte(c8, (uint8_t)FA_GET(8));
                if (bsc == 0) bad = true;
                else if (bsc == 1) fbs = 192;
                else if (bsc <= 5) fbs = 576 << (bsc - 2);
                else if (bsc == 6) { const uint32_t v = FA_GET(8); c8 = crc8_byte(c8, (uint8_t)v); fbs = (int)v + 1; }
                else if (bsc == 7) { const uint32_t v = FA_GET(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); fbs = (int)v + 1; }
                else fbs = 256 << (bsc - 8);
                if (src == 12) { c8 = crc8_byte(c8, (uint8_t)FA_GET(8)); }
                else if (src == 13 || src == 14) { const uint32_t v = FA_GET(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); }
                else if (src == 15) bad = true;
                if (FA_GET(8) != c8) bad = true;
                switch (ssc) {
                    case 0: fbps = si_bps; break;
                    case 1: fbps = 8; break;
                    case 2: fbps = 12; break;
                    case 4: fbps = 16; break;
                    case 5: fbps = 20; break;
                    case 6: fbps = 24; break;
                    case 7: fbps = 32; break;
                    default: bad = true; break;
                }
            }
            int64_t expect = a.stream_size - fstart;
            if (expect > a.B) expect = a.B;
            if (fbs != (int)expect) bad = true;
        }
        if constexpr (NCH == 2) {
            if (a.assign[task] >= 0) task_live = false;  // completed by an earlier (shallower) pass
        }
        if (bad) { task_live = false; atomicOr(a.err, kErrDecodeProcess); }
    }
    const bool out_aligned = ((reinterpret_cast<uintptr_t>(a.out_f32 ? (const void*)a.out_f32 : (const void*)a.out_i32) & 15) == 0);

    // read `n` (1..33) bits as a signed value; 33-bit values (side channels) only exist for NCH == 2
    auto get_wide = [&](int n) __attribute__((always_inline)) -> double {
        if constexpr (NCH == 2) {
            if (n > 32) {
                const uint32_t top = FA_GET(1);
                const uint32_t low = FA_GET(32);
                return (double)low - (top ? 4294967296.0 : 0.0);
            }
        }
        return (double)FA_GETS(n);
    };
    // low 32 bits of an exact integer in (-2^32, 2^32), as int32
    auto wrap32 = [&](double v) __attribute__((always_inline)) -> int32_t {
        if constexpr (NCH == 2) {
            if (v >= 2147483648.0) v -= 4294967296.0;
            else if (v < -2147483648.0) v += 4294967296.0;
        }
        return (int32_t)v;
    };

#pragma unroll 1
    for (int chn = 0; chn < NCH; ++chn) {
    int lo = 0, hi = 0;  // valid sample range inside this frame
    int bs = 0;
    int mode = 3;  // 0 const, 1 verbatim, 2 predictive, 3 idle
    int order = 0, bps = 0, wasted = 0;
    double cval = 0.0;
    double scale = 1.0;
    double c[MO], h[MO];
#pragma unroll
    for (int j = 0; j < MO; ++j) { c[j] = 0.0; h[j] = 0.0; }
    int plen = 4, esc = 15, ps = 0, pleft = 0, k = 0, escw = -1;
    uint32_t kp1 = 1;  // k + 1: the bits a code takes beyond its zeros (0 for a lane that consumes nothing, see below)
    uint32_t hbw = 0;  // NCH == 2: bit 32 (the sign) of the samples of the current tile

    if (task_live) {
        bool is_lpc = false;
        {
            bs = fbs;
            const uint32_t sf = FA_GET(8);
            const int tc = (int)((sf >> 1) & 0x3f);
            if (sf & 0x80) bad = true;
            if (sf & 1) wasted = (int)FA_UNARY() + 1;
            // a side channel (left/side, mid/side: channel 1; side/right: channel 0) has one more bit
            const int side = (NCH == 2 && ((chn == 1 && (ch_assign == 8 || ch_assign == 10)) || (chn == 0 && ch_assign == 9))) ? 1 : 0;
            bps = fbps + side - wasted;
            if (bps <= 0 || bps > 32 + side) bad = true;
            else if (tc == 0) { mode = 0; cval = get_wide(bps); }
            else if (tc == 1) { mode = 1; }
            else if (tc >= 8 && tc <= 12) { mode = 2; order = tc - 8; }
            else if (tc >= 32) { mode = 2; order = (tc & 31) + 1; is_lpc = true; }
            else bad = true;
            if (order > bs) bad = true;
        }
        if (!bad) {
            if (order > MO) {  // a later pass with a deeper history decodes this frame
                atomicOr(flags, order > 16 ? kFlagNeed32 : kFlagNeed16);
                mode = 3;
                task_live = false;
            } else if (NCH == 1 && order <= MO_DONE) {
                mode = 3;  // decoded by an earlier pass
            }
        }
        if (!bad && mode == 2) {
            // ---- warm-up samples, predictor description, residual header (serial per lane) ----
            for (int i = 0; i < order; ++i) {
                const double x = get_wide(bps);
                if (i < kTileW) {
                    tile[i * kLaneStride + (lane ^ ((i >> 2) * kTileSwz))] = wrap32(x);  // (wasted bits are restored in flush_tile)
                    if constexpr (NCH == 2) hbw |= (x < 0.0) ? (1u << i) : 0u;
                }
#pragma unroll
                for (int jj = 0; jj < MO; ++jj)
                    if (jj == (i % MO)) h[jj] = x;
            }
            if (is_lpc) {
                const int prec = (int)FA_GET(4) + 1;
                const int sh = FA_GETS(5);
                if (prec == 16 || sh < 0) bad = true;
                for (int j = 0; j < order; ++j) {
                    const double v = (double)FA_GETS(prec);
#pragma unroll
                    for (int jj = 0; jj < MO; ++jj)
                        if (jj == j) c[jj] = v;
                }
                scale = bitsd((uint64_t)(1023 - (sh < 0 ? 0 : sh)) << 52);
            } else {
                if (order == 1) { c[0] = 1.0; }
                else if (order == 2) { c[0] = 2.0; c[1] = -1.0; }
                else if (order == 3) { c[0] = 3.0; c[1] = -3.0; c[2] = 1.0; }
                else if (order == 4) { c[0] = 4.0; c[1] = -6.0; c[2] = 4.0; c[3] = -1.0; }
            }
            const int method = (int)FA_GET(2);
            const int po = (int)FA_GET(4);
            plen = method ? 5 : 4;
            esc = method ? 31 : 15;
            ps = bs >> po;
            if (method > 1 || (po > 0 && (ps << po) != bs) || ps < order) bad = true;
            pleft = -order;  // partition 0 is short by `order`
        }
        if (bad) { mode = 3; task_live = false; atomicOr(a.err, kErrDecodeProcess); }
        if (mode != 3) {
            int64_t l = sl_first - fstart, h2 = sl_last - fstart;
            if (l < 0) l = 0;
            if (h2 > bs) h2 = bs;
            lo = (int)l;
            hi = (int)(h2 > l ? h2 : l);
        }
    }
    // A VERBATIM first subframe (no wasted bits) is a run of fixed-width fields: verbatim_channel0_kernel has decoded it
    // with a lane per sample.  This lane steps over it -- bs x bps bits -- and sits out the channel; a wave whose 64 frames
    // all do (the low words of fine-grained float64 data are incompressible: every frame) skips the channel's loop.
    uint32_t resume_bitpos = 0;
    bool skipped = false;
    if constexpr (NCH == 2) {
        if (chn == 0 && task_live && mode == 1 && wasted == 0 && a.verbatim_done) {
            skipped = true;
            resume_bitpos = bitpos + (uint32_t)bps * (uint32_t)bs;
            mode = 3;
            lo = hi = 0;
        }
    }
    // ---- unify every lane as a "predictive" lane so the sample loop has one code path ----
    //   CONSTANT : order-1 predictor with c0 = 1 on h0 = value, zero-width escape residuals
    //   VERBATIM : no predictor, escape residuals of width bps in one endless partition
    //   idle     : zero-width escapes, nothing stored (empty row range)
    // A lane that consumes no bits in this channel -- CONSTANT subframe, idle lane, subframe decoded elsewhere -- must not
    // drag the wave into the rare branch at every sample (a zero-width escape fails the fast-path test; the high words of
    // an int64 array are CONSTANT frames more often than not: 3.3x the instructions per wave, measured).  It becomes a
    // plain Rice lane with k = 0 and a code that takes ZERO bits beyond its zeros (kp1 = 0), reading a ring column filled
    // with ones: no zeros, stop bit at once, value 0, position unchanged -- on the fast path.  Its real position is kept
    // aside and restored (ring re-read) when the channel is over.
    bool zero_width = false;
    if (mode == 3) { bs = 0x7fffffff; order = 0; ps = 0; zero_width = true; }
    if (mode == 0) {
#pragma unroll
        for (int j = 0; j < MO; ++j) { c[j] = 0.0; h[j] = cval; }
        c[0] = 1.0; scale = 1.0; order = 0;
        if constexpr (NCH == 2) {
            if (chn == 0) { resume_bitpos = bitpos; skipped = true; }  // the second subframe starts behind the constant
        }
        zero_width = true;
    }
    if (zero_width) { next_chunk = 0x00400000u; bitpos = 8; escw = -1; k = 0; kp1 = 0; pleft = 0x7fffffff; }
    if (__any(zero_width)) {
        if (zero_width) {
#pragma unroll 1
            for (int j = 0; j < kRingW + 2; ++j) ring[j * kLaneStride] = 0xffffffffu;  // (ring words and their two mirror words)
        }
    }
    if (mode == 1) {
#pragma unroll
        for (int j = 0; j < MO; ++j) c[j] = 0.0;
        scale = 1.0; order = 0; escw = bps; pleft = 0x7fffffff;
    }
    // coefficients pre-scaled by 2^-shift: every product and partial sum keeps its significand (|sum| < 2^53),
    // so floor(sum of (c 2^-shift) x) equals floor((sum of c x) 2^-shift) and the hot loop has no multiply by the scale
#pragma unroll
    for (int j = 0; j < MO; ++j) c[j] *= scale;
    // smallest blocksize in the wave decides where the guarded tail starts (idle lanes: huge)
    int bs_min = bs;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_xor(bs_min, off, 64);
        bs_min = o < bs_min ? o : bs_min;
    }
    bs_min = __builtin_amdgcn_readfirstlane(bs_min);
    // row descriptors for the cooperative store
    __builtin_amdgcn_wave_barrier();
    int64_t row0;  // output element index of frame sample 0
    if constexpr (NCH == 2) row0 = (task * 2 + chn) * (int64_t)a.B;
    else row0 = out_off + (fstart - sl_first);
    // The store pass `it` of a tile has this lane write a piece of row it * (64 / kTileG) + lane / kTileG: that
    // row's output offset is fetched from its owner lane once per frame and kept in registers (no LDS table:
    // the LDS saved is what lets more waves share a CU); the valid range is fetched in the rare clipped path.
    int64_t rout[kTileG];
#pragma unroll
    for (int it = 0; it < kTileG; ++it) {
        const int r = it * (64 / kTileG) + (lane / kTileG);
        const uint32_t rl = (uint32_t)__builtin_amdgcn_ds_bpermute(r << 2, (int)(uint32_t)(uint64_t)row0);
        const uint32_t rh = (uint32_t)__builtin_amdgcn_ds_bpermute(r << 2, (int)(uint32_t)((uint64_t)row0 >> 32));
        rout[it] = (int64_t)(((uint64_t)rh << 32) | rl);
    }
    if constexpr (F32) {
        float og = 0.0f, cf = 1.0f;
        if (mode != 3) {
            og = a.offsets[s];
            cf = (float)(1.0 / (double)a.gains[s]);  // utils.c:361
        }
        row_fg[lane] = make_float2(og, cf);
    }
    // a tile that lies inside every row's valid range and whose rows are 16-byte aligned in the
    // output is stored without per-element tests (the common case: whole frames of whole streams)
    int lo_max = lo, hi_min = hi, hi_max = hi;
    bool row_al = out_aligned && ((row0 & 3) == 0);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o1 = __shfl_xor(lo_max, off, 64), o2 = __shfl_xor(hi_min, off, 64), o3 = __shfl_xor(hi_max, off, 64);
        lo_max = o1 > lo_max ? o1 : lo_max;
        hi_min = o2 < hi_min ? o2 : hi_min;
        hi_max = o3 > hi_max ? o3 : hi_max;
    }
    lo_max = __builtin_amdgcn_readfirstlane(lo_max);
    hi_min = __builtin_amdgcn_readfirstlane(hi_min);
    hi_max = __builtin_amdgcn_readfirstlane(hi_max);
    const bool all_al = __all(row_al);
    // wasted bits are rare: the sample loop stores unshifted values and the tile flush restores the shift of each
    // row's frame only when some lane of the wave has one (wave-uniform branch)
    const bool any_wasted = __any(wasted != 0);
    // chunk `pend_ci` is requested one chunk-time before it is stored into the ring
    uint32_t pend_ci = next_chunk;
    Chunk pend = chunk_fetch(cbase, lim16, pend_ci);
    auto topup = [&]() __attribute__((always_inline)) {
        if ((bitpos >> kChunkShift) + 1 >= next_chunk) {
            if (pend_ci == next_chunk) chunk_store(ring, next_chunk, pend);
            else ring_load_chunk(cbase, lim16, ring, next_chunk);  // a slow read overtook the prefetch
            next_chunk++;
            pend_ci = next_chunk;
            pend = chunk_fetch(cbase, lim16, pend_ci);
        }
    };
    uint32_t pw0, pw1, pw2;  // ring words of the current bit position
    ring_words(lane4, bitpos, pw0, pw1, pw2);
    // fast-path key: a plain Rice code of at most kFastBits bits (stop bit inside the window)  <=>  zr < zlim (unsigned; v_ffbh_u32 gives 0xffffffff for an
    // empty window, which no bound admits; escaped partitions, finished frames and idle lanes carry the bound 0)
    // (The window arithmetic below is right for any code whose stop bit lies in the first 32 bits -- up to 64 bits long; what
    // limits the fast path is the top-up budget: at least 513 bits are resident after a top-up, 8 samples of at most
    // kFastBits = 40 bits + 8 partition parameters + the 96 bits of the last window stay inside.  At 32 bits -- the limit of
    // rounds 1-3 -- data of nearly full range, Rice parameters of 29-30 and codes of 31-34 bits, left the fast path at every
    // sample: the low words of the int64 benchmark array decoded at a third of the speed.)
    constexpr int kFastBits = (kChunkBytes == 64) ? 40 : 32;
    auto zlim_for = [&](int kk) __attribute__((always_inline)) -> uint32_t {
        const int lim = kFastBits - kk;
        return (uint32_t)(lim < 32 ? (lim > 0 ? lim : 0) : 32);
    };
    uint32_t zlim = (escw < 0) ? zlim_for(k) : 0u;

    // one sample of every lane.  GUARD: lanes may be in warm-up or past their frame's end.
    // PART: a partition boundary may fall inside this macro step (decided once per step for the wave)
    auto sample = [&](auto guard_tag, auto part_tag, auto u_tag, int i) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        constexpr bool PART = decltype(part_tag)::value;
        constexpr int u = decltype(u_tag)::value;
        bool live = true;
        if constexpr (GUARD) {
            live = (i < bs) && (i >= order);
            if (i >= bs) { escw = 0; zlim = 0u; pleft = 0x7fffffff; }  // frame finished: consume nothing more
        }
        if (live) {
            // rare events are tested wave-wide first, so the common case carries no exec-mask code
            if (PART && __builtin_expect(__any(pleft <= 0), 0)) {
                // A partition starts with its Rice parameter, plen bits at the lane's position -- the top of the window
                // the sample would read.  Taken from the window in place (the top-up budget covers it: 8 samples x 32 bits
                // + 8 parameters x 5 bits of the 512 resident); only an ESCAPED partition (parameter all ones, a raw
                // width follows) goes through the general reader.  (Round 4: through slow_param for every partition, a
                // frame of 16-32 partitions spent more time there -- one lane at a time -- than in its samples: 2x on
                // the reference's demo data, 5x on the quiet high words of int64 arrays.)
                const bool np = pleft <= 0;
                const uint32_t A0 = __builtin_amdgcn_alignbit(pw0, pw1, (~(bitpos - 1)) & 31);
                const uint32_t kk = A0 >> ((32 - plen) & 31);
                if (__builtin_expect(__any(np && (int)kk == esc), 0)) {
                    if (np) {
                        const ParamRet pr = slow_param(cbase, lim16, ring, bitpos, next_chunk, plen, esc);
                        k = (int)pr.k;
                        kp1 = pr.k + 1u;
                        escw = (int)pr.escw;  // -1: plain Rice partition
                        zlim = (escw < 0) ? zlim_for(k) : 0u;
                        bitpos = pr.bitpos;
                        next_chunk = pr.next_chunk;
                        pleft += ps;
                    }
                } else {
                    k = np ? (int)kk : k;
                    kp1 = np ? kk + 1u : kp1;
                    escw = np ? -1 : escw;
                    zlim = np ? zlim_for((int)kk) : zlim;
                    bitpos += np ? (uint32_t)plen : 0u;
                    pleft += np ? ps : 0;
                }
                ring_words(lane4, bitpos, pw0, pw1, pw2);
            }
#if FA_K7_SPEC
            // The three ring words of the window stay in registers from sample to sample: a fast-path code is at most 32 bits
            // long, so the next window starts in the same word or in the next one, and the only word it can newly need --
            // the one behind the three -- is requested HERE, from the position the sample starts at.  The LDS round trip
            // is then off the loop-carried chain (position -> address -> LDS -> window -> code length -> position): the
            // chain closes with a compare and three selects.
            const uint32_t pw3 = ring_word3(lane4, bitpos, 95u);
#endif
            // window from the words prefetched at the end of the previous sample
            const uint32_t sh = (~(bitpos - 1)) & 31;  // 32 - off'
            const uint32_t A = __builtin_amdgcn_alignbit(pw0, pw1, sh);
            const uint32_t Bw = __builtin_amdgcn_alignbit(pw1, pw2, sh);
            uint32_t zr;  // zeros before the stop bit; 0xffffffff when the window holds none (asm: __clz adds a v_min)
            asm("v_ffbh_u32 %0, %1" : "=v"(zr) : "v"(A));
            const int z = (int)zr;
            const bool fastok = (zr < zlim);
            // fast path for every lane (harmless where it does not apply): z zeros, stop bit, k low bits
            const uint32_t X = __builtin_amdgcn_alignbit(A, Bw, (uint32_t)(31 - z) & 31);  // bits after the stop bit
            const uint32_t low = __builtin_amdgcn_ubfe(X, (uint32_t)((32 - k) & 31), (uint32_t)k);
            const uint32_t uu = ((uint32_t)z << k) | low;
            int32_t r = (int32_t)(uu >> 1) ^ -(int32_t)(uu & 1);
            uint32_t nbp = bitpos + (uint32_t)z + kp1;
            double radd = 0.0;  // NCH == 2: what a 33-bit VERBATIM sample adds to its low word
            // one compare for both the wave vote and (inside the rare branch only) the lane's own answer
            const uint64_t okm = __ballot(fastok);
            if (__builtin_expect(okm != __builtin_amdgcn_read_exec(), 0)) {
                if (!((okm >> lane) & 1ull)) {
                    if (escw < 0 && zr < 32u && ((nbp + 128u) >> kChunkShift) < next_chunk) {
                        // a Rice code of 33..63 bits whose stop bit lies inside the window and whose
                        // successor's window is resident: r and nbp above are already right (the
                        // 32-bit limit of `fastok` only budgets the top-up, it is not a decoding limit)
                    } else if (escw == 0) {
                        r = 0;
                        nbp = bitpos;
                    } else if (escw > 0 && escw <= 32) {
                        // fixed-width sample (escaped partition, VERBATIM subframe): at most 32 bits,
                        // which the window already holds and the top-up budget covers
                        r = (int32_t)A >> (32 - escw);
                        nbp = bitpos + (uint32_t)escw;
                    } else if (NCH == 2 && escw > 32) {
                        const BitsRet t1 = slow_get(cbase, lim16, ring, bitpos, next_chunk, 1);
                        const BitsRet t2 = slow_get(cbase, lim16, ring, t1.bitpos, t1.next_chunk, 32);
                        r = (int32_t)t2.val;
                        radd = ((t2.val >> 31) ? 4294967296.0 : 0.0) - (t1.val ? 4294967296.0 : 0.0);
                        nbp = t2.bitpos;
                        next_chunk = ring_ensure(cbase, lim16, ring, t2.bitpos, t2.next_chunk);
                    } else {
                        const BitsRet sr = slow_sample(cbase, lim16, ring, bitpos, next_chunk, k, escw);
                        r = (int32_t)sr.val;
                        nbp = sr.bitpos;
                        next_chunk = sr.next_chunk;
                        if (sr.bad) {  // stop consuming: the rest of this lane's frame is zero-width
                            atomicOr(a.err, kErrDecodeProcess);
                            escw = 0; zlim = 0u; pleft = 0x7fffffff; r = 0; nbp = bitpos;
                        }
                    }
                }
#if FA_K7_SPEC
                // (some lane left the fast path: its position may have moved by any amount -- every lane reads its window again)
                bitpos = nbp;
                ring_words(lane4, bitpos, pw0, pw1, pw2);
#endif
            }
#if FA_K7_SPEC
            {
                const bool cross = (((nbp - 1u) ^ (bitpos - 1u)) >> 5) != 0u;  // (a lane re-read in the rare branch has bitpos == nbp)
                pw0 = cross ? pw1 : pw0;
                pw1 = cross ? pw2 : pw1;
                pw2 = cross ? pw3 : pw2;
                bitpos = nbp;
            }
#else
            bitpos = nbp;
            ring_words(lane4, bitpos, pw0, pw1, pw2);  // LDS latency hides behind the prediction below
#endif
            if constexpr (PART) pleft--;
            // every term is an exact integer in double, so the order is free: the newest sample enters last and the
            // loop-carried chain is one fma + floor + add
#if FA_K7_SEED  // experiment: the residual seeds the chain (floor(r + sum) = r + floor(sum), every partial sum exact): one add
                // less, but the eight dependent FMAs then start behind the bit decode instead of beside it
            double sum = (double)r;
            if constexpr (NCH == 2) sum += radd;
#pragma unroll
            for (int j = MO - 1; j >= 0; --j) sum = __builtin_fma(c[j], h[(u + MO - 1 - j) % MO], sum);
            const double xd = fa_floor(sum);
            if constexpr (NCH == 2) hbw |= (xd < 0.0) ? (1u << (i & 31)) : 0u;
#else
            double sum = 0.0;
#if !(FA_K7_X & 1)
#pragma unroll
            for (int j = MO - 1; j >= 0; --j) sum = __builtin_fma(c[j], h[(u + MO - 1 - j) % MO], sum);
#else
            sum = h[(u + MO - 1) % MO];
#endif
            double xd = (double)r + fa_floor(sum);
            if constexpr (NCH == 2) {
                xd += radd;
                hbw |= (xd < 0.0) ? (1u << (i & 31)) : 0u;
            }
#endif
            h[u % MO] = xd;
#if (FA_K7_X & 4)
            if (u == 0)
#endif
            tile[u * kLaneStride + (lane ^ ((u >> 2) * kTileSwz))] = wrap32(xd);
        } else if constexpr (GUARD) {
            // warm-up sample 16..31 (orders above 16): its value sits in the history
            if (i < bs && i >= kTileW) tile[u * kLaneStride + (lane ^ ((u >> 2) * kTileSwz))] = wrap32(h[u % MO]);
        }
    };

    // cooperative store of the tile: (64 / kTileG) rows x kTileW samples per pass, 16 bytes per lane
    auto flush_tile = [&](int tbase) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        if constexpr (NCH == 2) {
            // bit 32 of this lane's 32 samples (tiles are 32 samples wide)
            if (tbase < hi && tbase + kTileW > lo) a.hibits[(task * 2 + chn) * (int64_t)a.hib_words + (tbase >> 5)] = hbw;
            hbw = 0;
        }
        constexpr int kRowsPerPass = 64 / kTileG;
        if (all_al && tbase >= lo_max && tbase + kTileW <= hi_min) {
            // the whole tile lies inside every row's range and every row is 16-byte aligned: plain vector stores (for the
            // float32 output too: the restore of utils.c:364 with the row's offset and 1 / gain)
#pragma unroll
            for (int it = 0; it < kTileG; ++it) {
                const int r = it * kRowsPerPass + (lane / kTileG);
                const int cg = lane % kTileG;
                const int cb = 4 * cg;
                const int rsw = r ^ (cg * kTileSwz);
                int4 v = make_int4(tile[(cb + 0) * kLaneStride + rsw], tile[(cb + 1) * kLaneStride + rsw],
                                   tile[(cb + 2) * kLaneStride + rsw], tile[(cb + 3) * kLaneStride + rsw]);
                if (__builtin_expect(any_wasted, 0)) v = shl4(v, __builtin_amdgcn_ds_bpermute(r << 2, wasted));
                if constexpr (F32) {
                    const float2 fg = row_fg[r];
                    float4 o;
                    o.x = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.x));
                    o.y = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.y));
                    o.z = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.z));
                    o.w = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.w));
#if FA_K7_NT
                    {
                        typedef float f32x4_t __attribute__((ext_vector_type(4)));
                        f32x4_t vv; vv.x = o.x; vv.y = o.y; vv.z = o.z; vv.w = o.w;
                        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_t*>(a.out_f32 + rout[it] + tbase + cb));
                    }
#else
                    *reinterpret_cast<float4*>(a.out_f32 + rout[it] + tbase + cb) = o;
#endif
                } else {
#if (FA_K7_X & 2)
                    if (v.x == 0x12345678)
#endif
#if FA_K7_NT
                    {
                        typedef int i32x4_t __attribute__((ext_vector_type(4)));
                        i32x4_t vv; vv.x = v.x; vv.y = v.y; vv.z = v.z; vv.w = v.w;
                        __builtin_nontemporal_store(vv, reinterpret_cast<i32x4_t*>(a.out_i32 + rout[it] + tbase + cb));
                    }
#else
                    *reinterpret_cast<int4*>(a.out_i32 + rout[it] + tbase + cb) = v;
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
#pragma unroll
        for (int it = 0; it < kTileG; ++it) {
            const int r = it * kRowsPerPass + (lane / kTileG);
            const int cg = lane % kTileG;
            const int cb = 4 * cg;
            const int rsw = r ^ (cg * kTileSwz);  // the writer's swizzle
            int4 v = make_int4(tile[(cb + 0) * kLaneStride + rsw], tile[(cb + 1) * kLaneStride + rsw], tile[(cb + 2) * kLaneStride + rsw],
                               tile[(cb + 3) * kLaneStride + rsw]);
            if (__builtin_expect(any_wasted, 0)) v = shl4(v, __builtin_amdgcn_ds_bpermute(r << 2, wasted));
            const int2 rg = make_int2(__builtin_amdgcn_ds_bpermute(r << 2, lo), __builtin_amdgcn_ds_bpermute(r << 2, hi));
            const int si = tbase + cb;
            if (si + 3 >= rg.x && si < rg.y) {
                const int64_t ob = rout[it] + si;
                const bool fullv = (si >= rg.x) && (si + 3 < rg.y) && out_aligned && ((ob & 3) == 0);
                if constexpr (F32) {
                    const float2 fg = row_fg[r];
                    float4 o;
                    o.x = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.x));  // utils.c:364
                    o.y = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.y));
                    o.z = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.z));
                    o.w = __fadd_rn(fg.x, __fmul_rn(fg.y, (float)v.w));
                    if (fullv) *reinterpret_cast<float4*>(a.out_f32 + ob) = o;
                    else {
                        if (si + 0 >= rg.x && si + 0 < rg.y) a.out_f32[ob + 0] = o.x;
                        if (si + 1 >= rg.x && si + 1 < rg.y) a.out_f32[ob + 1] = o.y;
                        if (si + 2 >= rg.x && si + 2 < rg.y) a.out_f32[ob + 2] = o.z;
                        if (si + 3 >= rg.x && si + 3 < rg.y) a.out_f32[ob + 3] = o.w;
                    }
                } else {
                    if (fullv) *reinterpret_cast<int4*>(a.out_i32 + ob) = v;
                    else {
                        if (si + 0 >= rg.x && si + 0 < rg.y) a.out_i32[ob + 0] = v.x;
                        if (si + 1 >= rg.x && si + 1 < rg.y) a.out_i32[ob + 1] = v.y;
                        if (si + 2 >= rg.x && si + 2 < rg.y) a.out_i32[ob + 2] = v.z;
                        if (si + 3 >= rg.x && si + 3 < rg.y) a.out_i32[ob + 3] = v.w;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    const int bs_max = a.B;  // uniform loop bound (B >= every frame's blocksize)
    // a macro step is one output tile (32 samples): the tile index of every sample is then a compile-time
    // constant, so the swizzled LDS address of its store costs no instructions inside the loop
    constexpr int MACRO = kTileW;
    static_assert(MACRO % MO == 0 && MACRO % 16 == 0, "history rotation and tile must divide the macro step");
    auto macro_step = [&](auto guard_tag, auto part_tag, int i0) __attribute__((always_inline)) {
        static_for<MACRO>([&](auto ut) __attribute__((always_inline)) {
            constexpr int u = decltype(ut)::value;
            const int i = i0 + u;
            if constexpr ((u % kTopup) == 0) {
                // all lanes together: the next 8 fast-path samples need at most 32 bytes, and at
                // least 64 are resident after this top-up -> the sample code needs no residency test
                topup();
            }
            sample(guard_tag, part_tag, ut, i);
            if constexpr (u == MACRO - 1) flush_tile(i0);
        });
    };
    // guarded head (warm-up zone), unguarded main part, guarded tail (frames shorter than B).
    // The two guarded ranges share one loop so that the guarded body is instantiated once.
    constexpr int kStep = (MACRO > kTileW) ? MACRO : kTileW;
    int end = (bs_max + kStep - 1) / kStep * kStep;  // the last tile is stored whole
    if (chn == NCH - 1) {
        // nothing behind the last sample any lane of this wave has to deliver is decoded (a short read from the
        // head of a frame costs what it reads, not the frame); a first channel is always walked to its end,
        // because that is where the second sub-frame starts
        const int need = (hi_max + kStep - 1) / kStep * kStep;
        if (need < end) end = need;
    }
    if constexpr (NCH == 2) {
        if (!__any(mode != 3)) end = 0;  // nothing to decode in this channel for any lane of the wave
    }
    int main_lo = (32 + MACRO - 1) / MACRO * MACRO;
    if (main_lo > end) main_lo = end;
    int main_hi = (bs_min < bs_max ? bs_min : bs_max) / MACRO * MACRO;
    if (main_hi > end) main_hi = end;
    if (main_hi < main_lo) main_hi = main_lo;
    for (int pass = 0; pass < 2; ++pass) {
        const int g_lo = pass == 0 ? 0 : main_hi;
        const int g_hi = pass == 0 ? main_lo : end;
        for (int i0 = g_lo; i0 < g_hi; i0 += MACRO) macro_step(std::true_type{}, std::true_type{}, i0);
        if (pass == 0) {
            for (int i0 = main_lo; i0 < main_hi; i0 += MACRO) {
                // no lane's partition ends inside this step: the per-sample boundary test and count are dropped
                if (__any(pleft < MACRO)) {
                    macro_step(std::false_type{}, std::true_type{}, i0);
                } else {
                    macro_step(std::false_type{}, std::false_type{}, i0);
                    pleft -= MACRO;
                }
            }
        }
    }
    if constexpr (NCH == 2) {
        if (skipped) {  // on to the second subframe: the reader state at its first bit
            bitpos = resume_bitpos;
            next_chunk = ring_ensure(cbase, lim16, ring, bitpos, bitpos >> kChunkShift);
        }
    }
    }  // channel loop
    if constexpr (NCH == 2) {
        if (has_task && task_live) a.assign[task] = ch_assign;  // both subframes decoded
    }
}

// ------------------------------------------------------------------------------------------
// K7v (two-channel arrays): VERBATIM first subframes, one lane per sample.
// A lane of K7 walks its frame sample by sample whatever the subframe holds, and a VERBATIM sample goes through the escape
// branch of its loop: ~3x the cost of a Rice-coded one.  But a VERBATIM subframe is bs fields of bps bits at known
// positions -- and the low word of an int64 sample (channel 0, utils.c:96-123) is VERBATIM whenever the data is finer than
// 32 bits, i.e. for most float64 arrays.  One workgroup per task reads the frame header (its length only: K7 validates it),
// and if the first subframe is VERBATIM without wasted bits writes the wanted samples into K7's planar image and their
// bit 32 (33-bit side channels; the sign otherwise) into the bit image; K7 steps over such a subframe (`verbatim_done`).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void verbatim_channel0_kernel(DecodeArgs a) {
    const int64_t task = blockIdx.x;
    int64_t s, f, sl_first, sl_last;
    if (a.task_stream) {
        s = a.task_stream[task]; f = a.task_frame[task];
        sl_first = a.task_first[task]; sl_last = a.task_last[task];
    } else {
        s = task / a.nfr; f = a.f0 + (task - s * a.nfr);
        sl_first = a.first; sl_last = a.first + a.n_decode;
    }
    const StreamMeta m = a.meta[s];
    const int64_t at = a.ftab[s * a.nf + f];
    // (Every early return below is a frame that K7 either rejects or does not treat as "VERBATIM first subframe" itself: K7
    // skips the subframe only if it read the byte 0x02 at this same position from inside the blob.)
    if (m.first_frame < 0 || at < 0 || at + 5 > a.blob_bytes) return;
    const uint8_t* const p = a.blob + at;
    const uint32_t b2 = p[2], b3 = p[3], u0 = p[4];
    const int bsc = (int)(b2 >> 4), src = (int)(b2 & 15), ch = (int)(b3 >> 4), ssc = (int)((b3 >> 1) & 7);
    if (p[0] != 0xFF || p[1] != 0xF8 || (ch != 1 && ch != 8 && ch != 9 && ch != 10)) return;
    int extra = 0;
    if (u0 & 0x80) {
        int mbit = 0x40;
        while ((u0 & mbit) && extra < 7) { extra++; mbit >>= 1; }
        if (extra == 0 || extra > 6) return;
    }
    const int hl = 4 + 1 + extra + (bsc == 6 ? 1 : bsc == 7 ? 2 : 0) + (src == 12 ? 1 : (src == 13 || src == 14) ? 2 : 0) + 1;
    if (at + hl + 1 > a.blob_bytes || p[hl] != 0x02) return;  // the first subframe: VERBATIM (type 000001), no wasted bits
    int fbps;
    switch (ssc) {
        case 0: fbps = m.bps; break;
        case 1: fbps = 8; break;
        case 2: fbps = 12; break;
        case 4: fbps = 16; break;
        case 5: fbps = 20; break;
        case 6: fbps = 24; break;
        case 7: fbps = 32; break;
        default: return;
    }
    const int bps = fbps + (ch == 9 ? 1 : 0);  // side / right: channel 0 is the side channel, one bit wider
    if (bps <= 0 || bps > 33) return;
    const int64_t fstart = f * (int64_t)a.B;
    int64_t bs = a.stream_size - fstart;
    if (bs > a.B) bs = a.B;
    int64_t lo = sl_first - fstart, hi = sl_last - fstart;
    if (lo < 0) lo = 0;
    if (hi > bs) hi = bs;
    if (hi <= lo) return;
    int32_t* const t0 = a.out_i32 + (task * 2 + 0) * (int64_t)a.B;
    uint32_t* const h0 = a.hibits + (task * 2 + 0) * (int64_t)a.hib_words;
    const uint64_t bit0 = 8ull * (uint64_t)(hl + 1);
    const int lane = threadIdx.x & 63;
    // whole words of the bit image: samples [lo rounded down to 32, hi rounded up to 32), clipped to the block
    const int64_t i0 = lo & ~(int64_t)31;
    const int64_t i1 = ((hi + 31) & ~(int64_t)31) < ((bs + 31) & ~(int64_t)31) ? ((hi + 31) & ~(int64_t)31) : ((bs + 31) & ~(int64_t)31);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {  // (i1 - i0 is a multiple of 32: a wave's lanes 0-31 / 32-63 stay together)
        uint32_t low = 0;
        bool neg = false;
        if (i < bs) {
            const uint64_t bp = bit0 + (uint64_t)i * (uint64_t)bps;
            const int64_t q = at + (int64_t)(bp >> 3);
            uint64_t w = 0;  // eight bytes from q, big endian (the field starts up to 7 bits in and is at most 33 bits long)
            if (q + 8 <= a.blob_bytes) {
                uint32_t r0, r1;
                __builtin_memcpy(&r0, a.blob + q, 4);
                __builtin_memcpy(&r1, a.blob + q + 4, 4);
                w = ((uint64_t)__builtin_bswap32(r0) << 32) | __builtin_bswap32(r1);
            } else {
                for (int k = 0; k < 8; ++k) w = (w << 8) | (q + k < a.blob_bytes ? a.blob[q + k] : 0u);
            }
            const uint64_t v = (w << (bp & 7)) >> (64 - bps);  // the field, right-aligned
            neg = ((v >> (bps - 1)) & 1) != 0;
            // low 32 bits of the sign-extended value
            low = (bps >= 32) ? (uint32_t)v : (uint32_t)((int32_t)((uint32_t)v << (32 - bps)) >> (32 - bps));
            if (i >= lo && i < hi) t0[i] = (int32_t)low;
        }
        const uint64_t bal = __ballot(neg);
        if (lane == 0) h0[i >> 5] = (uint32_t)bal;
        if (lane == 32) h0[i >> 5] = (uint32_t)(bal >> 32);
    }
}

// ------------------------------------------------------------------------------------------
// K8 (two-channel arrays): planar (low words + bit 32) -> int64 or float64.  One workgroup per
// task (stream row x frame).  Undoes left/side, side/right and mid/side (RFC 9639 9.1.3); the
// reference's int64 value is (channel 1 << 32) | (channel 0 as unsigned) (utils.c:96-123), and the
// optional float64 restore is offsets + (1 / gains) * value (int64_to_float64, utils.c:329-348).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void combine_channels_kernel(DecodeArgs a, int64_t* __restrict__ out_i64,
                                                               double* __restrict__ out_f64, const double* __restrict__ offsets64,
                                                               const double* __restrict__ gains64) {
    const int64_t task = blockIdx.x;
    int64_t s, f, sl_first, sl_last, out_off;
    if (a.task_stream) {
        s = a.task_stream[task]; f = a.task_frame[task];
        sl_first = a.task_first[task]; sl_last = a.task_last[task];
        out_off = a.task_out_off[task];
    } else {
        s = task / a.nfr; f = a.f0 + (task - s * a.nfr);
        sl_first = a.first; sl_last = a.first + a.n_decode;
        out_off = s * a.n_decode;
    }
    const int asg = a.assign[task];
    if (asg < 0) {
        if (threadIdx.x == 0) atomicOr(a.err, kErrDecodeProcess);
        return;
    }
    const int64_t fstart = f * (int64_t)a.B;
    int64_t bs = a.stream_size - fstart;
    if (bs > a.B) bs = a.B;
    int64_t lo = sl_first - fstart, hi = sl_last - fstart;
    if (lo < 0) lo = 0;
    if (hi > bs) hi = bs;
    const int32_t* t0 = a.out_i32 + (task * 2 + 0) * (int64_t)a.B;
    const int32_t* t1 = a.out_i32 + (task * 2 + 1) * (int64_t)a.B;
    const uint32_t* h0 = a.hibits + (task * 2 + 0) * (int64_t)a.hib_words;
    const uint32_t* h1 = a.hibits + (task * 2 + 1) * (int64_t)a.hib_words;
    double off = 0.0, coeff = 1.0;
    if (out_f64) { off = offsets64[s]; coeff = 1.0 / gains64[s]; }
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        // 33-bit two's complement: low word plus bit 32, which is also the sign
        const int64_t c0 = (int64_t)(uint64_t)(uint32_t)t0[i] | (((h0[i >> 5] >> (i & 31)) & 1u) ? (int64_t)0xFFFFFFFF00000000LL : 0);
        const int64_t c1 = (int64_t)(uint64_t)(uint32_t)t1[i] | (((h1[i >> 5] >> (i & 31)) & 1u) ? (int64_t)0xFFFFFFFF00000000LL : 0);
        int64_t L, R;
        if (asg == 1) { L = c0; R = c1; }
        else if (asg == 8) { L = c0; R = c0 - c1; }
        else if (asg == 9) { R = c1; L = c0 + c1; }
        else {
            const int64_t mid = (int64_t)((uint64_t)c0 << 1) | (c1 & 1);
            L = (mid + c1) >> 1;
            R = (mid - c1) >> 1;
        }
        const int64_t v = (int64_t)(((uint64_t)R << 32) | (uint64_t)(uint32_t)L);
        const int64_t o = out_off + (fstart - sl_first) + i;
        if (out_f64) out_f64[o] = off + coeff * (double)v;
        else out_i64[o] = v;
    }
}

}  // namespace fa
