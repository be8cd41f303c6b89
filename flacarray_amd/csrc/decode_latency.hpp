// decode_latency.hpp -- K7L: one WAVEFRONT decodes one frame, for launches with few frames.
//
// The throughput decoder (decode_frames_kernel, K7) gives every lane a frame of its own: a launch with fewer frames
// than lanes leaves the chip idle and takes as long as one lane needs for one frame, ~1 ms -- the reference's usage
// pattern, one small read per call (src/flacarray/array.py:409-449 -> decompress.c:281-298: seek_absolute +
// process_single), is exactly that launch.  Here the 64 lanes of a wave share ONE frame:
//
//   * the frame's bytes are fetched with coalesced loads into an LDS image (big-endian words);
//   * headers, warm-up samples and predictor description are read by all lanes alike (uniform control flow);
//   * the Rice codes of a partition are located in parallel: the partition's bit range is cut into 64 segments, every
//     lane parses code LENGTHS from its segment's first bit -- usually the middle of a code -- through its own segment
//     and the two after it.  Two parses that ever stand on the same bit stay together, and lane 0, which starts on a
//     code boundary, is the true one: lane l's parse is known to be true from the entry of segment l + 1 on if it
//     left its own segment where the true parse of lane l - 1 or l - 2 did.  A segment whose entry is reached by a true
//     parse has a known entry bit and code count; the leading run of such segments gets sample indices from a prefix
//     sum of the counts, a second pass decodes the values, and what is left (rarely anything) is taken up by the next
//     round, which starts where the last resolved segment ended;
//   * the predictor: FIXED on all lanes alike in wrapping 32-bit integers; LPC in exact doubles as in K7, in transposed
//     form with the taps spread over the 16 lanes of a row (lat_restore_lpc), only as far as the last sample the read
//     asks for;
//   * the requested range is stored with coalesced stores (optionally dequantised, utils.c:350-368).
//
// What this kernel does not take (mid / side frames and LPC side channels that need their 33rd bit, blocks above 4096 samples,
// predictor orders above 12, frames that do
// not fit the image, anything that does not parse) raises a flag and the launch is repeated by K7, which also owns
// all error reporting.  Results are bit-identical to K7's.
#pragma once
#include "decode_kernels.hpp"

namespace fa {

#ifndef FA_LAT_X
#define FA_LAT_X 0  // timing experiments only (wrong results): 1 = no predictor, 2 = one parse round only, 4 = no value pass
#endif
#ifndef FA_LAT_SEG_BASE
#define FA_LAT_SEG_BASE 4  // shortest segment of the speculative parse: FA_LAT_SEG_BASE + (k + 1)^2 / FA_LAT_SEG_DIV codes
#endif
#ifndef FA_LAT_SEG_DIV
#define FA_LAT_SEG_DIV 36
#endif
#ifndef FA_LAT_STAMPS
#define FA_LAT_STAMPS 0  // diagnostic build: lane 0 of task 0 prints where its time went (100 MHz ticks), tools/lat_time.py
#endif
#if FA_LAT_STAMPS
#define FA_LAT_STAMP(i) do { lat_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FA_LAT_STAMP(i) do { } while (0)
#endif
constexpr int kLatMaxBlock = 4096;
constexpr int kLatImgWords = 4416;  // 17664 bytes: a 4096-sample VERBATIM frame at 32 bits per sample is 16384 + headers
constexpr int kLatPadWords = 8;     // zero words behind the image: speculative parses may look past the frame
constexpr uint32_t kLatUnaryMax = 1u << 16;

// tasks of a tiny launch travel as a kernel argument (no upload, no synchronisation before the launch)
struct LatInline {
    int64_t stream[8], frame[8], first[8], last[8], out_off[8];
    int32_t n;
};

// The image is kept in REVERSE word order (word w of the frame at img[kLatTop - w]): the two words of a window then
// come out of one ds_read2_b32 as the low and high half of a 64-bit register pair in the order the shift wants them
// (forward order costs a swap and its wait state per window, and a window is read for every code of the parse).
constexpr uint32_t kLatTop = (uint32_t)(kLatImgWords + kLatPadWords - 1);
// (two-channel frames: a VERBATIM low word and whatever the high word takes -- the image is twice as long; TOP is the index
// of the image's last word, i.e. of the frame's word 0)
constexpr int kLatImgWords2 = 2 * kLatImgWords;
constexpr uint32_t kLatTop2 = (uint32_t)(kLatImgWords2 + kLatPadWords - 1);
template <uint32_t TOP>
__device__ __forceinline__ uint32_t lat_win(const uint32_t* img, uint32_t pos) {  // bits [pos, pos + 32)
    uint32_t wi = pos >> 5;
    wi = wi < TOP - 1u ? wi : TOP - 1u;  // (zero words behind the frame)
    const uint32_t* const q = img + (TOP - 1u - wi);  // q[0] = word wi + 1, q[1] = word wi
    const uint64_t v = ((uint64_t)q[1] << 32) | q[0];
    return (uint32_t)((v << (pos & 31)) >> 32);
}
// the same for positions known to lie inside the image (the walks of the parse stop at the frame's end)
template <uint32_t TOP>
__device__ __forceinline__ uint32_t lat_win_in(const uint32_t* img, uint32_t pos) {
    const uint32_t* const q = img + (TOP - 1u - (pos >> 5));
    const uint64_t v = ((uint64_t)q[1] << 32) | q[0];
    return (uint32_t)((v << (pos & 31)) >> 32);
}

// length of the Rice code that starts at bit p (0 = no stop bit within reach: not a code)
template <uint32_t TOP>
__device__ __forceinline__ uint32_t lat_code_len(const uint32_t* img, uint32_t p, uint32_t k, uint32_t lim) {
    uint32_t q = 0, A;
    for (;;) {
        A = lat_win<TOP>(img, p + q);
        if (A != 0 || p + q >= lim || q > kLatUnaryMax) break;
        q += 32;
    }
    if (A == 0) return 0;
    return q + (uint32_t)__clz((int)A) + 1u + k;
}

constexpr int kLatResPad = 80;  // the predictor loops work in whole blocks (LPC: 16 and one block of look-ahead; FIXED: 64) behind the last sample of a 4096 block

// The LPC recurrence of one frame -- serial by nature -- with the TAPS spread over the lanes of a row.
//
// Transposed form: a new sample x_n is added, times c_j, into the running sum of sample n + 1 + j (j = 0 .. order - 1);
// a sample's sum starts as its residual and is complete when sample n - 1 has been added, x = floor(sum): the residual
// seeds the sum, x = floor(r + sum c_j x_j 2^-shift) = r + floor(sum ...), every partial sum being an exact multiple of
// 2^-shift below 2^53 (coefficients < 2^15, samples < 2^32, at most 12 terms) -- so the order of the terms is free and
// the result is bit-identical to K7's.  Lane m of a 16-lane row owns the sums of samples base + m, base + 16 + m, ...;
// at step k the row's lane k holds a complete sum, `v_floor_f64_dpp row_newbcast:k` floors it and hands it to all 16
// lanes in one instruction, and ONE `v_fmac_f64` adds it to the 16 sums with the lane's own coefficient for that
// distance (sixteen rotated copies of the coefficient vector, zero beyond the order, are kept in registers: lane m,
// step k: c[(m - k - 1) mod 16]).  Two vector instructions per sample instead of order + 2; the chain from sample to
// sample is those two.  Every four steps the four lanes that have completed hand over their samples and take the
// residuals of the samples 16 further on -- early enough for orders up to 12, whose first term for a sample arrives
// 12 steps before it, and late enough that the lane's value has been broadcast; in between such a lane only adds
// zeros.  The four rows of the wave do the same work on the same data.  (A lone wave issues one vector instruction per
// ~6 cycles, profiles/r01k_valu_rates.txt: 116 us of an order-8 frame's 182 were the predictor with order + 4
// instructions per sample on every lane alike.)
// The FIXED recurrences (RFC 9639 9.2.5) are repeated prefix sums: with e_j the j-th backward difference of the samples,
// e_order is the residual, e_j[i] = e_j[i - 1] + e_(j+1)[i], e_0 the samples -- `order` passes of a running sum over the
// block, each started from the difference of the warm-up samples.  A pass takes the block 64 samples at a time: lane l
// holds sample base + l, an inclusive scan across the wave (DPP, as the encoder's row writer uses), the carry of the tile
// before it added as a scalar.  Every lane reads and writes only its own positions in every pass, so the passes need no
// barrier between them.  Wrapping 32-bit arithmetic is exact (every sample fits its 32 bits, and a 33-bit side channel is
// wanted modulo 2^32 only).  ~10 instructions per 64 samples and pass; the serial form this replaces (all lanes alike,
// four samples per trip through LDS) took 75-180 cycles per SAMPLE: 130-310 us for a whole FIXED frame, what the high
// words of an int64 array and quiet, small-valued data are made of.
__device__ __forceinline__ uint32_t lat_scan_incl_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ void lat_restore_fixed(int32_t* res, int order, int hi) {
    const int lane = threadIdx.x & 63;
    uint32_t w[4] = {0u, 0u, 0u, 0u};  // warm-up samples, newest first: w[t] = x[order - 1 - t]
    for (int t = 0; t < order; ++t) w[t] = (uint32_t)res[order - 1 - t];
    for (int j = order - 1; j >= 0; --j) {
        // e_j[order - 1]: the j-th backward difference of the warm-up samples
        uint32_t carry;
        if (j == 0) carry = w[0];
        else if (j == 1) carry = w[0] - w[1];
        else if (j == 2) carry = w[0] - 2u * w[1] + w[2];
        else carry = w[0] - 3u * w[1] + 3u * w[2] - w[3];
        for (int base = order; base < hi; base += 64) {  // (the last tile may run past `hi` into values nobody reads)
            const uint32_t s = lat_scan_incl_u32((uint32_t)res[base + lane]) + carry;
            res[base + lane] = (int32_t)s;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
        }
    }
}

template <int K>
__device__ __forceinline__ void lat_lpc_step(double& acc, double c) {
    double x;
    // (two wait states between the vector instruction that wrote `acc` and a DPP read of it; floor and FMA sit in one
    // statement so that the compiler does not pad the DPP's result with a wait state of its own)
    asm volatile("s_nop 1\n\tv_floor_f64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64 %1, %0, %2"
                 : "=&v"(x), "+v"(acc) : "v"(c), "n"(K));
}

__device__ __forceinline__ void lat_restore_lpc(int32_t* res, const double* coef16, int order, int hi) {
    const int m = threadIdx.x & 15;
    double cr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cr[k] = coef16[(m - k - 1) & 15];  // (zero from `order` on)
    int base = order;
    // the first 16 sums: residual + the terms of the warm-up samples
    double acc = (double)res[base + m];
    for (int j = 0; j < order; ++j) {
        const int xi = base + m - 1 - j;  // >= 0: j < order = base
        if (xi < base) acc = __builtin_fma(coef16[j], (double)res[xi], acc);
    }
    const bool g0 = (m >> 2) == 0, g1 = (m >> 2) == 1, g2 = (m >> 2) == 2, g3 = (m >> 2) == 3;
    for (; base < hi; base += 16) {  // (the last block and the look-ahead may run past `hi` into values nobody reads)
        const double rn = (double)res[base + 16 + m];
        int32_t out = 0;
        auto hand_over = [&](bool mine) __attribute__((always_inline)) {
            const int32_t t = (int32_t)fa_floor(acc);  // (v_cvt_i32_f64 saturates: see the side-channel check of the caller)
            out = mine ? t : out;
            acc = mine ? rn : acc;
        };
#define FA_LAT_STEP(K) lat_lpc_step<K>(acc, cr[K]);
        FA_LAT_STEP(0) FA_LAT_STEP(1) FA_LAT_STEP(2) FA_LAT_STEP(3)
        hand_over(g0);
        FA_LAT_STEP(4) FA_LAT_STEP(5) FA_LAT_STEP(6) FA_LAT_STEP(7)
        hand_over(g1);
        FA_LAT_STEP(8) FA_LAT_STEP(9) FA_LAT_STEP(10) FA_LAT_STEP(11)
        hand_over(g2);
        FA_LAT_STEP(12) FA_LAT_STEP(13) FA_LAT_STEP(14) FA_LAT_STEP(15)
        hand_over(g3);
#undef FA_LAT_STEP
        res[base + m] = out;  // (the four rows store the same 16 values)
    }
}

// outputs of the two-channel variant (int64 / float64 arrays: low and high word of a sample are the two channels,
// utils.c:96-123; the restore is utils.c:329-348)
struct LatWide {
    int64_t* out_i64;
    double* out_f64;
    const double* offsets;
    const double* gains;
};

// NCH == 2: frames with two independent channels (what the int64 encoder writes for all but small-valued frames), left /
// side and side / right (the latter is what it writes for small values of both signs).  A side channel has 33 bits and
// this decoder keeps 32: the differences are undone modulo 2^32 -- left = side + right and right = left - side are exact
// in their low 32 bits whatever bit 32 of the side was, and so are CONSTANT, VERBATIM and the FIXED recurrences (integer
// coefficients); an LPC-coded side channel needs the true values and hands the frame to K7 if one of them leaves int32.
// mid / side (a shift of a 33-bit sum) stays with K7.  F32 then means "float64 output".
template <bool F32, int NCH>
__global__ __launch_bounds__(64) void decode_latency_kernel(DecodeArgs a, LatInline inl, LatWide wd, int* fallback) {
    constexpr int kImgWords = (NCH == 2) ? kLatImgWords2 : kLatImgWords;
    constexpr uint32_t kTop = (NCH == 2) ? kLatTop2 : kLatTop;
    __shared__ __attribute__((aligned(16))) uint32_t img[kImgWords + kLatPadWords];
    __shared__ __attribute__((aligned(16))) int32_t res_all[NCH][kLatMaxBlock + kLatResPad];
    __shared__ double coef_s[16];  // pre-scaled coefficients, zero from the order on
    const int lane = threadIdx.x;
    const int64_t task = blockIdx.x;
    if (task >= a.n_tasks) return;
    // reason bits (diagnostics only; any non-zero value makes the host repeat the launch with K7):
    // 1 geometry / frame does not fit, 2 header, 4 subframe kind or order, 8 residual layout, 16 no parse, 32 overrun
    auto give_up = [&](int why) __attribute__((always_inline)) {
        // (a plain store: the word may live in pinned host memory, and all the host asks is "non-zero?"; with several
        // frames giving up the diagnostic shows one of their reasons)
        if (lane == 0) *reinterpret_cast<volatile int*>(fallback) = why;
    };

#if FA_LAT_STAMPS
    uint64_t lat_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t lat_rounds = 0, lat_parts = 0, lat_segs = 0, lat_codes = 0, lat_k = 0, lat_S = 0;
#endif
    FA_LAT_STAMP(0);
    // ---- task (wave-uniform) ----
    int64_t s, f, sl_first, sl_last, out_off;
    if (inl.n > 0) {
        s = inl.stream[task]; f = inl.frame[task]; sl_first = inl.first[task]; sl_last = inl.last[task]; out_off = inl.out_off[task];
    } else if (a.task_stream) {
        s = a.task_stream[task]; f = a.task_frame[task];
        sl_first = a.task_first[task]; sl_last = a.task_last[task];
        out_off = a.task_out_off[task];
    } else {
        s = task / a.nfr; f = a.f0 + (task - s * a.nfr);
        sl_first = a.first; sl_last = a.first + a.n_decode;
        out_off = s * a.n_decode;
    }
    const StreamMeta m = a.meta[s];
    const int64_t at = a.ftab[s * a.nf + f];
    const int64_t nxt = (f + 1 < a.nf) ? a.ftab[s * a.nf + f + 1] : m.end_abs;
    if (m.first_frame < 0 || at < 0 || nxt <= at + 6 || nxt > a.blob_bytes || a.B > kLatMaxBlock || m.channels != NCH) { give_up(1); return; }
    const int64_t base = at & ~(int64_t)15;
    const uint32_t skip = (uint32_t)(at - base);
    const uint32_t nbytes = skip + (uint32_t)(nxt - at);
    if (nbytes > (uint32_t)kImgWords * 4u) { give_up(1); return; }
    const uint32_t frame_end_bits = nbytes * 8u;  // (includes the CRC-16; an upper bound is all that is needed)

    // ---- frame image: coalesced 16-byte loads, big-endian words ----
    {
        // (the frame lies inside the blob and the blob starts on a 16-byte boundary: every piece can be read whole.)
        // Eight loads in flight per lane: a loop of load / wait / store pays the memory latency once per 1 KB, ~10 times
        // per frame of the benchmark data.
        const uint32_t pieces = (nbytes + 15u) >> 4;
        const uint8_t* const q0 = a.blob + base;
        for (uint32_t i0 = 0; i0 < pieces; i0 += 512) {
            uint4 d[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                uint32_t i = i0 + 64u * (uint32_t)t + (uint32_t)lane;
                i = i < pieces ? i : pieces - 1u;  // (unconditional loads: a predicated one would be waited for on its own)
                d[t] = *reinterpret_cast<const uint4*>(q0 + 16 * (int64_t)i);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t i = i0 + 64u * (uint32_t)t + (uint32_t)lane;
                if (i < pieces) {
                    uint4 o;
                    // words 4 i .. 4 i + 3 go to img[kTop - 4 i - 3 .. kTop - 4 i], highest word first
                    o.w = __builtin_bswap32(d[t].x); o.z = __builtin_bswap32(d[t].y); o.y = __builtin_bswap32(d[t].z); o.x = __builtin_bswap32(d[t].w);
                    *reinterpret_cast<uint4*>(&img[kTop - 3u - 4u * i]) = o;
                }
            }
        }
        for (uint32_t i = 4 * pieces + lane; i < (uint32_t)(kImgWords + kLatPadWords); i += 64) img[kTop - i] = 0;
    }
    __syncthreads();
    FA_LAT_STAMP(1);

    // ---- uniform bit reader ----
    uint32_t pos = skip * 8u;
    bool bad = false;
    auto get = [&](int n) __attribute__((always_inline)) -> uint32_t {  // n in 1..32
        if (pos + (uint32_t)n > frame_end_bits) { bad = true; return 0u; }
        const uint32_t v = lat_win<kTop>(img, pos) >> (32 - n);
        pos += (uint32_t)n;
        return v;
    };
    auto gets = [&](int n) __attribute__((always_inline)) -> int32_t {
        if (n == 0) return 0;
        const uint32_t v = get(n);
        return (int32_t)(v << (32 - n)) >> (32 - n);
    };

    // ---- frame header (RFC 9639 9.1), CRC-8 verified ----
    int bs = 0, fbps = 0, assign = 0;
    {
        uint8_t c8 = 0;
        const uint32_t w = get(32);
        c8 = crc8_byte(c8, (uint8_t)(w >> 24)); c8 = crc8_byte(c8, (uint8_t)(w >> 16));
        c8 = crc8_byte(c8, (uint8_t)(w >> 8)); c8 = crc8_byte(c8, (uint8_t)w);
        if ((w >> 16) != 0xFFF8) bad = true;
        const uint32_t b2 = (w >> 8) & 0xff, b3 = w & 0xff;
        const int bsc = (int)(b2 >> 4), src = (int)(b2 & 15), ch = (int)(b3 >> 4), ssc = (int)((b3 >> 1) & 7);
        if constexpr (NCH == 2) {
            if ((ch != 1 && ch != 8 && ch != 9) || (b3 & 1)) bad = true;  // independent, left / side, side / right
            assign = ch;
        } else {
            if (ch != 0 || (b3 & 1)) bad = true;
        }
        const uint32_t u0 = get(8);
        c8 = crc8_byte(c8, (uint8_t)u0);
        int extra = 0;
        if (u0 & 0x80) {
            int mbit = 0x40;
            while ((u0 & mbit) && extra < 7) { extra++; mbit >>= 1; }
            if (extra == 0 || extra > 6) bad = true;
        }
        for (int i = 0; i < extra && !bad; ++i) c8 = crc8_byte(c8, (uint8_t)get(8));
        if (bsc == 0) bad = true;
        else if (bsc == 1) bs = 192;
        else if (bsc <= 5) bs = 576 << (bsc - 2);
        else if (bsc == 6) { const uint32_t v = get(8); c8 = crc8_byte(c8, (uint8_t)v); bs = (int)v + 1; }
        else if (bsc == 7) { const uint32_t v = get(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); bs = (int)v + 1; }
        else bs = 256 << (bsc - 8);
        if (src == 12) { c8 = crc8_byte(c8, (uint8_t)get(8)); }
        else if (src == 13 || src == 14) { const uint32_t v = get(16); c8 = crc8_byte(c8, (uint8_t)(v >> 8)); c8 = crc8_byte(c8, (uint8_t)v); }
        else if (src == 15) bad = true;
        if (get(8) != c8) bad = true;
        switch (ssc) {
            case 0: fbps = m.bps; break;
            case 1: fbps = 8; break;
            case 2: fbps = 12; break;
            case 4: fbps = 16; break;
            case 5: fbps = 20; break;
            case 6: fbps = 24; break;
            case 7: fbps = 32; break;
            default: bad = true; break;
        }
    }
    const int64_t fstart = f * (int64_t)a.B;
    {
        int64_t expect = a.stream_size - fstart;
        if (expect > a.B) expect = a.B;
        if (bs != (int)expect || bs > kLatMaxBlock) bad = true;
    }
    if (bad) { give_up(2); return; }
    int64_t l0 = sl_first - fstart, h0 = sl_last - fstart;
    if (l0 < 0) l0 = 0;
    if (h0 > bs) h0 = bs;
    const int lo = (int)l0, hi = (int)(h0 > l0 ? h0 : l0);  // samples [lo, hi) of this frame are wanted
    if (hi <= lo) return;

    int wasted_lo = 0, wasted_hi = 0;  // wasted bits of channel 0 / the last channel
#pragma unroll 1
    for (int chn = 0; chn < NCH; ++chn) {
    int32_t* const res = res_all[chn];
    // (the first of two channels is parsed to its end whatever the read wants: the second begins where it stops)
    const int hi_parse = (NCH == 2 && chn == 0) ? bs : hi;
    // ---- subframe ----
    const uint32_t sf = get(8);
    const int tc = (int)((sf >> 1) & 0x3f);
    int wasted = 0;
    if (sf & 0x80) bad = true;
    if (sf & 1) {
        uint32_t z = 0;
        while (!bad && get(1) == 0) { if (++z > 32) bad = true; }
        wasted = (int)z + 1;
    }
    const int side = (NCH == 2 && ((assign == 8 && chn == 1) || (assign == 9 && chn == 0))) ? 1 : 0;
    const int bps = fbps + side - wasted;
    if (bps <= 0 || bps > 32 + side) bad = true;
    if (bad) { give_up(4); return; }
    // a field of `bps` bits, low 32 bits of its value; `fits` = false when the value needs the 33rd
    bool fits = true;
    auto gets_wide = [&](int n) __attribute__((always_inline)) -> int32_t {
        if (n <= 32) return gets(n);
        const uint32_t top = get(1), low = get(32);
        fits = fits && (top == (low >> 31));
        return (int32_t)low;
    };
    int order = 0;
    bool is_lpc = false;
    if (tc == 0) {  // CONSTANT
        const int32_t v = gets_wide(bps);
        if (bad) { give_up(4); return; }
        for (int i = lo + lane; i < hi; i += 64) res[i] = v;
    } else if (tc == 1) {  // VERBATIM: fixed-width fields, one lane per sample
        if (pos + (uint32_t)bps * (uint32_t)bs > frame_end_bits) { give_up(4); return; }
        if (bps > 32) {  // a 33-bit side channel: the low 32 bits of every field
            for (int i = lo + lane; i < hi; i += 64) res[i] = (int32_t)lat_win<kTop>(img, pos + (uint32_t)i * 33u + 1u);
        } else {
            for (int i = lo + lane; i < hi; i += 64) {
                const uint32_t v = lat_win<kTop>(img, pos + (uint32_t)i * (uint32_t)bps) >> (32 - bps);
                res[i] = (int32_t)(v << (32 - bps)) >> (32 - bps);
            }
        }
        pos += (uint32_t)bps * (uint32_t)bs;
    } else if ((tc >= 8 && tc <= 12) || tc >= 32) {
        if (tc >= 32) { order = (tc & 31) + 1; is_lpc = true; }
        else order = tc - 8;
        if (order > bs || order > 12) { give_up(4); return; }  // (orders 13..32 exist in foreign streams: K7 takes them)
        for (int i = 0; i < order; ++i) {
            const int32_t v = gets_wide(bps);
            if (lane == 0) res[i] = v;
        }
        if (tc >= 32 && !fits) { give_up(4); return; }  // an LPC side channel with a 33-bit warm-up sample: K7 (doubles)
        int shift = 0;
        if (is_lpc) {
            const int prec = (int)get(4) + 1;
            shift = gets(5);
            if (prec == 16 || shift < 0) bad = true;
            const double scale = bitsd((uint64_t)(1023 - (shift < 0 ? 0 : shift)) << 52);
            if (lane < 16) coef_s[lane] = 0.0;
            for (int j = 0; j < order; ++j) {
                const double v = (double)gets(prec) * scale;  // pre-scaled by 2^-shift (exact, as in K7)
                if (lane == 0) coef_s[j] = v;
            }
        }
        const int method = (int)get(2);
        const int po = (int)get(4);
        const int plen = method ? 5 : 4, esc = method ? 31 : 15;
        const int ps = bs >> po;
        if (bad || method > 1 || (po > 0 && (ps << po) != bs) || ps < order) { give_up(8); return; }
        __syncthreads();

        FA_LAT_STAMP(2);
        // ---- residual: partitions in turn, the codes of a partition in parallel ----
        uint32_t idx0 = (uint32_t)order;           // sample index of the next residual
        uint32_t left_in_frame = (uint32_t)(bs - order);
        for (int p = 0; p < (1 << po) && idx0 < (uint32_t)hi_parse; ++p) {
            uint32_t n = (uint32_t)(p == 0 ? ps - order : ps);
            const uint32_t k = get(plen);
            if (bad) { give_up(8); return; }
            if ((int)k == esc) {
                const int wbits = (int)get(5);
                if (bad || (uint64_t)pos + (uint64_t)wbits * n > frame_end_bits) { give_up(8); return; }
                for (uint32_t i = lane; i < n; i += 64) {
                    int32_t v = 0;
                    if (wbits) {
                        const uint32_t u = lat_win<kTop>(img, pos + i * (uint32_t)wbits) >> (32 - wbits);
                        v = (int32_t)(u << (32 - wbits)) >> (32 - wbits);
                    }
                    res[idx0 + i] = v;
                }
                pos += (uint32_t)wbits * n;
                idx0 += n;
                left_in_frame -= n;
                continue;
            }
            // nothing behind the last wanted sample is decoded
            uint32_t want = n;
            if (idx0 + want > (uint32_t)hi_parse) want = (uint32_t)hi_parse - idx0;
            const bool last_partition_needed = (want < n);
            uint32_t b = pos, todo = want;
#if FA_LAT_STAMPS
            lat_parts++; lat_codes += want;
#endif
            while (todo > 0) {
                // segment length: this partition's share of the frame's remaining bits, spread over 64 lanes; at least
                // a few codes long so that a parse started in the middle of a code has room to fall in step
                const uint32_t rem_bits = frame_end_bits > b ? frame_end_bits - b : 64u;
                // (a heuristic: single precision will do, and a 64-bit integer division is ~100 instructions per round)
                const float bits_per_code = (float)rem_bits / (float)(left_in_frame ? left_in_frame : 1u);
                uint32_t S = (uint32_t)(bits_per_code * (float)todo * (1.0f / 64.0f)) + 1u;
                // A parse that starts inside a code falls in step with the true one when the two land on the same bit;
                // their distance does a random walk whose steps are the differences of the unary parts: a few codes at
                // k = 0, around a hundred at k = 16 (noise-like data).  With ONE segment of continuation per lane a lane
                // that had not fallen in step inside its own segment ended the round, and segments had to be that long
                // (32 + 4 k codes: 64 us for a FIXED-0 frame of the benchmark data, k = 16..17; 32 codes: 125 us, 32 + 14 k:
                // 154 us, profiles/r03_reads.md).  With TWO segments of continuation (below) the neighbour's parse carries
                // the truth across such a lane and the frame's own spread over the 64 lanes (64 codes each) resolves in
                // one round: 53 us (32 + 4 k with the third walk: 75 us).
                // How long that takes grows with the square of the code length (the walk has to cover the code's length
                // in steps of a bit or two): ~50 codes at k = 16, a handful at k = 4 -- and a frame with many small
                // partitions, which are parsed one after the other, needs its lanes busy in each of them: the shortest
                // segment is FA_LAT_SEG_BASE + (k + 1)^2 / FA_LAT_SEG_DIV codes (tools/lat_sweep.py, frames of 32
                // partitions at noise amplitudes 2^1 .. 2^20: 32 + k codes whatever k: 0.92 - 1.08 ms; 4 + (k+1)^2 / 6:
                // 0.37 - 1.58; / 12: 0.37 - 0.93; / 24: 0.37 - 0.61; / 48: 0.36 - 0.52; 4 codes: 0.35 - 1.04).
                constexpr uint32_t kSegBase = FA_LAT_SEG_BASE, kSegDiv = FA_LAT_SEG_DIV;
                const uint32_t avg_bits = (uint32_t)bits_per_code + 1u;
                // ... for SMALL partitions (at most 512 codes).  A big partition -- the benchmark's frames are one
                // partition of 4088 codes -- is best served by segments long enough to verify at once, 32 + k codes, also
                // when only its beginning is wanted (segments doubling from round to round, an eighth of the run as the
                // minimum, a cap at a third of what is left: all measured, all slower on one of the two kinds of frame).
                const uint32_t cmin = (n <= 512u) ? kSegBase + (k + 1u) * (k + 1u) / kSegDiv : 32u + k;
                const uint32_t smin = cmin * avg_bits;
                if (S < smin) S = smin;
                const uint32_t lim = frame_end_bits + 64u;
                // phase A: code lengths through the lane's own segment and the two after it.  (Both phases read the image
                // through lat_win, two LDS words per code.  A register window of four words per lane, refilled one word
                // per 32 bits of progress and two words ahead, was built and measured: 93 us instead of 64 for a FIXED-0
                // frame -- the refill's divergent shifts cost a lone wave more issue slots than the LDS round trips it saves.)
                uint32_t q0 = b + (uint32_t)lane * S;
                const uint32_t lim1 = q0 + S;
                // (a parse that finds no further stop bit has run past the frame's last code: it ends there, with the
                // sentinel as its exit -- whatever lies behind the last code is cut off by the code count below)
                uint32_t xs[3], cs[3];  // exit from the lane's own segment and the two after it; codes counted in each
                bool ended = false;
                // (the walks are written with wave-uniform control: every lane takes every step, those that are done just do
                // not advance.  Per-lane loops cost ~50 instructions per code, two thirds of them exec-mask bookkeeping --
                // and a lone wave pays ~6 cycles for each; this form takes ~20.)
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const uint32_t limw = lim1 + (uint32_t)w * S;
                    uint32_t c = 0;
                    for (;;) {  // four codes per trip: the trip's bookkeeping is as long as a code's own work (1 / 2 / 4 per trip: 53 / 49 / 46 us)
                        bool slow = false;
                        auto step = [&]() __attribute__((always_inline)) -> bool {
                            ended = ended || q0 >= lim;  // (behind the frame: zeros, no code)
                            const bool go = !ended && q0 < limw;
                            const uint32_t A = lat_win_in<kTop>(img, ended ? 0u : q0);
                            const bool adv = go && A != 0;
                            slow = slow || (go && A == 0);  // 32 zeros or more: the general reader below
                            q0 += adv ? (uint32_t)__clz((int)A) + 1u + k : 0u;
                            c += adv ? 1u : 0u;
                            return go;
                        };
                        const bool any_go = __builtin_amdgcn_ballot_w64(step()) != 0;
                        if (!any_go) break;
                        (void)step();
                        (void)step();
                        (void)step();
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(slow) != 0, 0)) {
                            if (slow) {
                                const uint32_t len = lat_code_len<kTop>(img, q0, k, lim);
                                if (len == 0) ended = true;
                                else { q0 += len; c++; }
                            }
                        }
                    }
                    xs[w] = ended ? 0xffffffffu : q0;
                    cs[w] = c;
                }
                // Verification.  Two parses that ever stand on the same bit stay together, and lane 0's is the true one.
                // Lane l's parse is TRUE at the entry of segment l + 1 if it left its own segment where a true parse of
                // lane l - 1 or l - 2 did (t1), at the entry of segment l + 2 if it was true before or left segment l + 1
                // where lane l - 1's true parse did (t2).  With two segments of continuation a lane that has not fallen in
                // step within its own segment does not end the round: its neighbour's parse carries the truth across.
                const uint32_t x1p1 = (uint32_t)__shfl_up((int)xs[0], 1, 64), c2p1 = (uint32_t)__shfl_up((int)cs[1], 1, 64);
                const uint32_t x2p1 = (uint32_t)__shfl_up((int)xs[1], 1, 64), x3p1 = (uint32_t)__shfl_up((int)xs[2], 1, 64);
                const uint32_t x2p2 = (uint32_t)__shfl_up((int)xs[1], 2, 64), x3p2 = (uint32_t)__shfl_up((int)xs[2], 2, 64);
                const uint32_t c3p2 = (uint32_t)__shfl_up((int)cs[2], 2, 64);
                const uint64_t mA = __ballot(lane >= 1 && xs[0] == x2p1);  // joins lane l-1 at the entry of segment l+1
                const uint64_t mB = __ballot(lane >= 2 && xs[0] == x3p2);  // joins lane l-2 there
                const uint64_t mC = __ballot(lane >= 1 && xs[1] == x3p1);  // joins lane l-1 at the entry of segment l+2
                if (__builtin_amdgcn_readfirstlane((int)cs[0]) == 0) { give_up(16); return; }  // not one code where one must be: damaged (K7 reports it)
                uint64_t t1 = 1, t2 = 1;
                // (only as many segments as the codes still wanted can fill -- lanes behind them parse what follows the
                // partition and would keep this scalar loop going for nothing; an underestimate costs another round)
                const int need = (int)fminf(64.0f, bits_per_code * (float)todo / (float)S + 3.0f);
                for (int l = 1; l < need; ++l) {
                    const uint64_t p1 = (t2 >> (l - 1)) & 1u, p2 = (l >= 2) ? ((t2 >> (l - 2)) & 1u) : 0u;
                    if (!(p1 | p2)) break;  // no true parse in the two lanes before: nothing reaches further
                    const uint64_t b1 = ((p1 & (mA >> l)) | (p2 & (mB >> l))) & 1u;
                    const uint64_t b2 = (b1 | (p1 & (mC >> l))) & 1u;
                    t1 |= b1 << l;
                    t2 |= b2 << l;
                }
                // segment m is resolved (true entry and code count known) by lane m-1 true at its entry, or by lane m-2
                const uint64_t by1 = t1 << 1, by2 = t2 << 2;
                const uint64_t resolved = 1ull | by1 | by2;
                const int nseg = (resolved == ~0ull) ? 64 : __builtin_ctzll(~resolved);  // segments 0 .. nseg-1
#if FA_LAT_STAMPS
                lat_rounds++; lat_segs += (uint32_t)nseg; lat_k = k; lat_S = S;
#endif
                const bool active = lane < nseg;
                const bool mine1 = (by1 >> lane) & 1u;
                const uint32_t entry = (lane == 0) ? b : (mine1 ? x1p1 : x2p2);
                const uint32_t cnt = active ? ((lane == 0) ? cs[0] : (mine1 ? c2p1 : c3p2)) : 0u;
                const uint32_t incl = wave_incl_scan_u32(cnt);
                const uint32_t first_idx = incl - cnt;
                // phase B: values
                uint32_t pe = entry;
                const uint32_t seg_end = b + ((uint32_t)lane + 1u) * S;
                bool bad_lane = false;
                {
                    const bool mine = !(FA_LAT_X & 4) && active && first_idx < todo;
                    uint32_t j = first_idx;
                    for (;;) {
                        const bool go = mine && !bad_lane && pe < seg_end && j < todo;
                        if (__builtin_amdgcn_ballot_w64(go) == 0) break;
                        const uint32_t pin = (go && pe < lim) ? pe : 0u;  // (a true code starts inside the frame)
                        uint32_t A = lat_win_in<kTop>(img, pin);
                        uint32_t z = (uint32_t)__clz((int)A);
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(go && A == 0) != 0, 0)) {  // 32 zeros or more somewhere: the general reader
                            if (go) {
                                uint32_t q = 0;
                                for (;;) {
                                    A = lat_win<kTop>(img, pe + q);
                                    if (A != 0 || pe + q >= lim || q > kLatUnaryMax) break;
                                    q += 32;
                                }
                                if (A == 0) bad_lane = true;
                                z = q + (uint32_t)__clz((int)A);
                            }
                        }
                        const uint32_t lowpos = pe + z + 1u;
                        const uint32_t low = k ? (lat_win<kTop>(img, (go && !bad_lane) ? lowpos : 0u) >> (32 - k)) : 0u;
                        const uint32_t uu = (z << k) | low;
                        const bool put = go && !bad_lane;
                        // (lanes that are done store into the last pad word, which nobody reads)
                        res[put ? idx0 + j : (uint32_t)(kLatMaxBlock + kLatResPad - 1)] = (int32_t)(uu >> 1) ^ -(int32_t)(uu & 1);
                        pe = put ? lowpos + k : pe;
                        j += put ? 1u : 0u;
                    }
                }
                if (__any(bad_lane)) { give_up(16); return; }
                const uint64_t took = __ballot(active && first_idx < todo);
                const int last = 63 - __builtin_clzll(took);  // (lane 0 always takes part)
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t done = total < todo ? total : todo;
                b = (uint32_t)__shfl((int)pe, last, 64);
                idx0 += done;
                todo -= done;
                left_in_frame -= done;
                if (b > frame_end_bits) { give_up(32); return; }
                if (FA_LAT_X & 2) todo = 0;
            }
            pos = b;
            if (last_partition_needed) break;
        }
        __syncthreads();
        FA_LAT_STAMP(3);
        // ---- predictor ----
        if ((FA_LAT_X & 1) != 0) {
        } else if (is_lpc) {
            lat_restore_lpc(res, coef_s, order, hi);
            if (side) {
                // A side channel has 33 bits and this decoder keeps 32; an LPC recurrence (unlike the FIXED ones, exact
                // modulo 2^32) needs the true values.  The recurrence itself runs in exact doubles, only its stores are
                // 32 bits wide and saturate: a sample at either end of the int32 range means "possibly more than 32
                // bits" and hands the frame to K7 (a true INT32_MIN / MAX does so too: slower, never wrong).
                __syncthreads();
                bool edge = false;
                for (int i = order + lane; i < hi; i += 64) edge = edge || res[i] == INT32_MAX || res[i] == INT32_MIN;
                if (__builtin_amdgcn_ballot_w64(edge) != 0) { give_up(4); return; }
            }
        } else if (order > 0) {
            lat_restore_fixed(res, order, hi);
        }
    } else {
        give_up(4);
        return;
    }
    if (chn == 0) wasted_lo = wasted;
    if (chn == NCH - 1) wasted_hi = wasted;
    __syncthreads();
    }  // (channels)

    FA_LAT_STAMP(4);
    // ---- store [lo, hi) ----
    const int64_t row0 = out_off + (fstart - sl_first);
    auto sample = [&](int i) __attribute__((always_inline)) -> int32_t { return (int32_t)((uint32_t)res_all[0][i] << wasted_lo); };
    if constexpr (NCH == 2) {
        // channel 0 is the low word, channel 1 the high word (utils.c:96-123)
        auto wide = [&](int i) __attribute__((always_inline)) -> int64_t {
            const uint32_t c0 = (uint32_t)res_all[0][i] << wasted_lo, c1 = (uint32_t)res_all[NCH - 1][i] << wasted_hi;
            // left / side: right = left - side; side / right: left = side + right (modulo 2^32: see the kernel's header)
            const uint32_t lw = (assign == 9) ? c0 + c1 : c0;
            const uint32_t hw = (assign == 8) ? c0 - c1 : c1;
            return (int64_t)(((uint64_t)hw << 32) | lw);
        };
        if constexpr (F32) {
            const double og = wd.offsets[s];
            const double cf = 1.0 / wd.gains[s];  // utils.c:342
            for (int i = lo + lane; i < hi; i += 64) wd.out_f64[row0 + i] = og + cf * (double)wide(i);  // utils.c:345 (as K8)
        } else {
            for (int i = lo + lane; i < hi; i += 64) wd.out_i64[row0 + i] = wide(i);
        }
    } else if constexpr (F32) {
        const float og = a.offsets[s];
        const float cf = (float)(1.0 / (double)a.gains[s]);  // utils.c:361
        for (int i = lo + lane; i < hi; i += 64) a.out_f32[row0 + i] = __fadd_rn(og, __fmul_rn(cf, (float)sample(i)));  // utils.c:364
    } else {
        for (int i = lo + lane; i < hi; i += 64) a.out_i32[row0 + i] = sample(i);
    }
#if FA_LAT_STAMPS
    FA_LAT_STAMP(5);
    if (task == 0 && lane == 0)
        printf("K7L stamps (us): image %.2f headers %.2f parse %.2f predictor %.2f store %.2f | partitions %u rounds %u segments resolved %u codes %u last k %u last S %u bits\n", (lat_t[1] - lat_t[0]) * 0.01,
               (lat_t[2] - lat_t[1]) * 0.01, (lat_t[3] - lat_t[2]) * 0.01, (lat_t[4] - lat_t[3]) * 0.01, (lat_t[5] - lat_t[4]) * 0.01,
               lat_parts, lat_rounds, lat_segs, lat_codes, lat_k, lat_S);
#endif
}

}  // namespace fa
