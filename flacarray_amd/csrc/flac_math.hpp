// flac_math.hpp -- wave-uniform scalar pieces of the FLAC frame analysis, usable from host and
// device (FA_HD).  Everything here is either integer arithmetic or IEEE double arithmetic in a
// fixed operation order with contraction disabled (-ffp-contract=off), so that the HIP kernels
// reproduce the CPU oracle bit for bit.
//
// Replaces the analysis hidden behind the reference's libFLAC calls
// (FLAC__stream_encoder_process_interleaved, src/flacarray/libflacarray/compress.c:374-378).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FA_HD __host__ __device__ __forceinline__
#else
#define FA_HD inline
#endif

// Non-template kernels defined in headers that more than one translation unit includes get internal linkage in the
// secondary units (the unit that launches them keeps the external definition).
#if defined(FA_UNIT_FUSED) || defined(FA_UNIT_PLACED)
#define FA_GLOBAL static
#else
#define FA_GLOBAL
#endif

namespace fa {

constexpr int kMaxBlock = 4096;        // largest frame the kernels stage in LDS
constexpr int kChunk = 64;             // samples per lane in the chunked phases
constexpr int kRow = 256;              // samples per row in the bit-writing phase
constexpr int kRowCapBits = 12288;     // longer residual rows force a VERBATIM subframe
constexpr int kRiceLimit = 31;         // Rice2 escape; largest usable parameter is 30
constexpr int kMaxLpcOrder = 12;       // levels 7, 8
constexpr int kSlotBytes = 16640;      // per-frame scratch slot (verbatim 4096x32 bit + headers), 65 * 256

struct LevelParams {
    int blocksize;
    int max_lpc_order;
    int max_porder;
    int qlp_precision;
};

FA_HD LevelParams level_params(uint32_t level) {
    const int lpc[9] = {0, 0, 0, 6, 8, 8, 8, 12, 12};
    const int po[9] = {3, 3, 3, 4, 4, 5, 6, 6, 6};
    LevelParams p;
    p.blocksize = (level <= 2) ? 1152 : 4096;
    p.max_lpc_order = lpc[level];
    p.max_porder = po[level];
    p.qlp_precision = (p.blocksize <= 384) ? 13 : (p.blocksize <= 1152 ? 14 : 15);
    return p;
}

FA_HD uint64_t dbits(double x) {
    union { double d; uint64_t u; } c;
    c.d = x;
    return c.u;
}
FA_HD double bitsd(uint64_t u) {
    union { double d; uint64_t u; } c;
    c.u = u;
    return c.d;
}

// floor(log2(v)) for v > 0
FA_HD int ilog2_u64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)v);
#else
    return 63 - __builtin_clzll(v);
#endif
}

// log2 from +,-,*,/ only: identical results on host and device
FA_HD double det_log2(double x) {
    uint64_t b = dbits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = bitsd((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e + (2.8853900817779268 * s) * p;
}

FA_HD double fa_floor(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_floor(x);
#else
    return __builtin_floor(x);
#endif
}
FA_HD double fa_ceil(double x) { return __builtin_ceil(x); }
FA_HD double fa_fabs(double x) { return __builtin_fabs(x); }

// Levinson-Durbin on autoc[0..MLO].  coef[(o-1)*MLO + j], j<o: order-o predictor (as float);
// err[o-1]: residual energy.  Returns the usable maximum order.
template <int MLO>
FA_HD int levinson(const double* autoc, int max_order, float* coef, double* err_out) {
    double lpc[MLO];
    double err = autoc[0];
#pragma unroll
    for (int i = 0; i < MLO; ++i) {
        if (i < max_order) {
            double r = -autoc[i + 1];
#pragma unroll
            for (int j = 0; j < MLO; ++j)
                if (j < i) r = r - lpc[j] * autoc[i - j];
            r = r / err;
            lpc[i] = r;
#pragma unroll
            for (int j = 0; j < MLO / 2; ++j) {
                if (j < (i >> 1)) {
                    double tmp = lpc[j];
                    lpc[j] = lpc[j] + r * lpc[i - 1 - j];
                    lpc[i - 1 - j] = lpc[i - 1 - j] + r * tmp;
                }
            }
            if (i & 1) lpc[i >> 1] = lpc[i >> 1] + lpc[i >> 1] * r;
            err = err * (1.0 - r * r);
#pragma unroll
            for (int j = 0; j < MLO; ++j)
                if (j <= i) coef[i * MLO + j] = (float)(-lpc[j]);
            err_out[i] = err;
            if (err == 0.0) return i + 1;
        }
    }
    return max_order;
}

FA_HD int best_lpc_order(const double* err, int max_order, int total_samples, int overhead_bits) {
    double error_scale = 0.5 / (double)total_samples;
    int best_index = 0;
    double best_bits = 4294967295.0;
    for (int indx = 0; indx < max_order; ++indx) {
        int order = indx + 1;
        double bps;
        if (err[indx] > 0.0) {
            bps = 0.5 * det_log2(error_scale * err[indx]);
            if (!(bps >= 0.0)) bps = 0.0;
        } else if (err[indx] < 0.0) {
            bps = 1e32;
        } else {
            bps = 0.0;
        }
        double bits = bps * (double)(total_samples - order) + (double)(order * overhead_bits);
        if (bits < best_bits) { best_index = indx; best_bits = bits; }
    }
    return best_index + 1;
}

// Quantise `order` float coefficients to `precision`-bit integers; returns 0 on success.
FA_HD int quantize_coefs(const float* c, int order, int precision, int32_t* q, int* shift) {
    precision--;
    int32_t qmax = (1 << precision) - 1, qmin = -(1 << precision);
    double cmax = 0.0;
    for (int i = 0; i < order; ++i) {
        double d = fa_fabs((double)c[i]);
        if (d > cmax) cmax = d;
    }
    if (cmax <= 0.0) return 2;
    int log2cmax = (int)((dbits(cmax) >> 52) & 0x7ff) - 1023;
    int sh = precision - log2cmax - 1;
    if (sh > 15) sh = 15;
    else if (sh < -16) return 1;
    double error = 0.0;
    for (int i = 0; i < order; ++i) {
        if (sh >= 0) error = error + (double)c[i] * (double)(1 << sh);
        else error = error + (double)c[i] / (double)(1 << (-sh));
        double rq = (error >= 0.0) ? fa_floor(error + 0.5) : fa_ceil(error - 0.5);
        if (rq > (double)qmax) rq = (double)qmax;
        else if (rq < (double)qmin) rq = (double)qmin;
        error = error - rq;
        q[i] = (int32_t)rq;
    }
    *shift = sh < 0 ? 0 : sh;
    return 0;
}

// the same with the coefficient index unrolled (q stays in registers on the device); q[j] = 0 for j >= order
template <int MLO>
FA_HD int quantize_coefs_t(const float* c, int order, int precision, int32_t (&q)[MLO], int* shift) {
    precision--;
    const int32_t qmax = (1 << precision) - 1, qmin = -(1 << precision);
    double cmax = 0.0;
#pragma unroll
    for (int i = 0; i < MLO; ++i) {
        q[i] = 0;
        if (i < order) {
            const double d = fa_fabs((double)c[i]);
            if (d > cmax) cmax = d;
        }
    }
    if (cmax <= 0.0) return 2;
    const int log2cmax = (int)((dbits(cmax) >> 52) & 0x7ff) - 1023;
    int sh = precision - log2cmax - 1;
    if (sh > 15) sh = 15;
    else if (sh < -16) return 1;
    // c * 2^sh for either sign of sh: scaling by a power of two is exact, so this equals the reference
    // form (multiply by 1 << sh, or divide by 1 << -sh) bit for bit, without eight divisions in a serial chain
    const double mul = bitsd((uint64_t)(1023 + sh) << 52);
    double error = 0.0;
#pragma unroll
    for (int i = 0; i < MLO; ++i) {
        if (i < order) {
            error = error + (double)c[i] * mul;
            double rq = (error >= 0.0) ? fa_floor(error + 0.5) : fa_ceil(error - 0.5);
            if (rq > (double)qmax) rq = (double)qmax;
            else if (rq < (double)qmin) rq = (double)qmin;
            error = error - rq;
            q[i] = (int32_t)rq;
        }
    }
    *shift = sh < 0 ? 0 : sh;
    return 0;
}

// largest Rice partition order for a frame: level limit, divisibility, 64-sample chunk
// geometry, predictor order
FA_HD int max_porder_for(int bs, int level_max, int pred_order) {
    int p = 0, b = bs;
    while (!(b & 1) && p < 15) { p++; b >>= 1; }
    if (p > level_max) p = level_max;
    while (p > 0 && ((bs >> p) % kChunk) != 0) p--;
    while (p > 0 && (bs >> p) <= pred_order) p--;
    return p;
}

// Rice parameter estimate and estimated bits for one partition with n samples and sum of
// magnitudes `mean`
FA_HD int rice_param(uint64_t mean, uint32_t n) {
    uint64_t fpd = 0x40000u / n;
    int k;
    if (mean < 2 || (((mean - 1) * fpd) >> 18) == 0) k = 0;
    else k = ilog2_u64(((mean - 1) * fpd) >> 18) + 1;
    if (k >= kRiceLimit) k = kRiceLimit - 1;
    return k;
}
FA_HD uint64_t rice_part_bits(uint64_t mean, uint32_t n, int k) {
    uint64_t pb = 4 + (uint64_t)(1 + k) * (uint64_t)n + (k ? (mean >> (k - 1)) : (mean << 1)) - (uint64_t)(n >> 1);
    if (pb > 0xffffffffULL) pb = 0xffffffffULL;
    return pb;
}

FA_HD int blocksize_code(int bs) {
    switch (bs) {
        case 192: return 1;
        case 576: return 2;
        case 1152: return 3;
        case 2304: return 4;
        case 4608: return 5;
        case 256: return 8;
        case 512: return 9;
        case 1024: return 10;
        case 2048: return 11;
        case 4096: return 12;
        case 8192: return 13;
        case 16384: return 14;
        case 32768: return 15;
        default: return (bs <= 256) ? 6 : 7;
    }
}

// CRC-8 (poly 0x07) one byte at a time through a 16-entry nibble table held in two constants
FA_HD uint8_t crc8_byte(uint8_t crc, uint8_t v) {
    uint32_t c = (uint32_t)(crc ^ v);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t i = c >> 4;
        const uint64_t tab = (i < 8) ? 0x15121b1c090e0700ULL : 0x2d2a232431363f38ULL;
        c = ((c << 4) & 0xFFu) ^ (uint32_t)((tab >> (8 * (i & 7))) & 0xFFu);
    }
    return (uint8_t)c;
}
FA_HD uint16_t crc16_byte(uint16_t crc, uint8_t v) {
    crc ^= (uint16_t)((uint16_t)v << 8);
    for (int b = 0; b < 8; ++b) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x8005) : (uint16_t)(crc << 1);
    return crc;
}

// bytes of "fLaC" + STREAMINFO + SEEKTABLE(nframes points)
FA_HD int64_t stream_header_bytes(int64_t nframes) { return 4 + 4 + 34 + 4 + 18 * nframes; }

}  // namespace fa
