/*
 * flac_oracle.c -- CPU ORACLE for the flacarray hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * file's shared object.  The product (flacarray_amd/) never links, imports or calls it.
 *
 * What it restates (reference = /root/reference, hpc4cmb/flacarray v0.3.4):
 *   - plumbing of encode()/encode_threaded()   src/flacarray/libflacarray/compress.c:133-435
 *     (one complete native FLAC stream per row, starts = exclusive scan, error bitmask)
 *   - plumbing of decode()                     src/flacarray/libflacarray/decompress.c:194-313
 *     (arbitrary starts/nbytes, [first,last) slice semantics, row stride n_decode)
 *   - float32_to_int32 / int32_to_float32      src/flacarray/libflacarray/utils.c:160-243,350-368
 *
 * The FLAC arithmetic itself lives in the reference's un-vendored dependency libFLAC
 * (meson.build:13 ">= 1.4.0", wheels pin 1.5.0: packaging/wheels/install_deps_linux.sh:53),
 * which is ABSENT from this image.  The codec below is a restatement of the published
 * format (RFC 9639) and of libFLAC's published level-0..8 encoder heuristics
 * (fixed-predictor search, tukey(0.5) window + autocorrelation + Levinson-Durbin,
 * coefficient quantisation, partitioned-Rice parameter estimate), written so that every
 * decision is reproducible bit-for-bit on a GPU: integer arithmetic everywhere except the
 * LPC analysis, which is IEEE double with a FIXED operation order (64 "lane" partial sums of
 * explicit fma() followed by an xor-butterfly) and a polynomial log2.
 *
 * PARITY STATUS: byte-level parity with libFLAC's *encoder output* is UNPINNED (no libFLAC,
 * no golden bytes in the reference tree: every reference test is a round trip,
 * tests/bindings.py:27-230, tests/array.py:26-146).  What is pinned: (i) the bitstream is
 * checked against hand-assembled RFC 9639 frames (tests/golden/), (ii) decode(encode(x)) == x
 * for the reference's own test recipes, (iii) the float quantisation follows utils.c
 * operation by operation (float/double mix, truncation, x86 cast behaviour).
 *
 * Build: make -C oracle    (gcc -O2 -mfma -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Error bit-codes: same values as the reference, flacarray.h:20-40 */
#define ERROR_NONE 0
#define ERROR_ALLOC (1 << 0)
#define ERROR_INVALID_LEVEL (1 << 1)
#define ERROR_ZERO_NSTREAM (1 << 2)
#define ERROR_ZERO_STREAMSIZE (1 << 3)
#define ERROR_ENCODE_PROCESS (1 << 9)
#define ERROR_DECODE_INIT (1 << 13)
#define ERROR_DECODE_PROCESS (1 << 14)
#define ERROR_DECODE_STREAMSIZE (1 << 16)
#define ERROR_DECODE_SAMPLE_RANGE (1 << 17)

#define MAX_BLOCK 4096
#define MAX_ORDER 32
#define ROW_SAMPLES 256
#define ROW_CAP_BITS 12288 /* a 256-sample row of residual codes longer than this => VERBATIM */
#define RICE_LIMIT 31      /* Rice2 escape code; largest usable parameter is 30 */

/* ------------------------------------------------------------------------------------------
 * CRC-8 (poly 0x07) and CRC-16 (poly 0x8005), both MSB-first, init 0   (RFC 9639 9.1.8, 9.3)
 * ---------------------------------------------------------------------------------------- */
static uint8_t crc8_tab[256];
static uint16_t crc16_tab[256];
static int crc_ready = 0;

static void crc_init(void) {
    if (crc_ready) return;
    for (int i = 0; i < 256; ++i) {
        uint8_t c = (uint8_t)i;
        for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
        crc8_tab[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int b = 0; b < 8; ++b) d = (d & 0x8000) ? (uint16_t)((d << 1) ^ 0x8005) : (uint16_t)(d << 1);
        crc16_tab[i] = d;
    }
    crc_ready = 1;
}
static uint8_t crc8(const uint8_t *p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) c = crc8_tab[c ^ p[i]];
    return c;
}
static uint16_t crc16(const uint8_t *p, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ crc16_tab[(c >> 8) ^ p[i]]);
    return c;
}

/* ------------------------------------------------------------------------------------------
 * MSB-first bit writer on a growable byte buffer
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *buf;
    size_t cap;    /* bytes */
    uint64_t nbits; /* bits written */
    int err;
} bitw_t;

static void bw_reserve(bitw_t *w, uint64_t more_bits) {
    size_t need = (size_t)((w->nbits + more_bits + 7) / 8) + 8;
    if (need <= w->cap) return;
    size_t ncap = w->cap ? w->cap : 4096;
    while (ncap < need) ncap *= 2;
    uint8_t *nb = (uint8_t *)realloc(w->buf, ncap);
    if (!nb) { w->err = 1; return; }
    memset(nb + w->cap, 0, ncap - w->cap);
    w->buf = nb;
    w->cap = ncap;
}
/* write the low n bits of v (n <= 64), a byte at a time */
static void bw_put(bitw_t *w, uint64_t v, unsigned n) {
    if (n == 0) return;
    if (((w->nbits + n + 7) >> 3) + 8 > w->cap) bw_reserve(w, n);
    if (w->err) return;
    while (n) {
        const unsigned bitoff = (unsigned)(w->nbits & 7), room = 8 - bitoff;
        const unsigned take = n < room ? n : room;
        const uint8_t bits = (uint8_t)((v >> (n - take)) & ((1u << take) - 1u));
        w->buf[w->nbits >> 3] |= (uint8_t)(bits << (room - take));
        w->nbits += take;
        n -= take;
    }
}
static void bw_zeros(bitw_t *w, uint64_t n) {
    bw_reserve(w, n);
    if (w->err) return;
    w->nbits += n; /* buffer is kept zeroed beyond nbits */
}
static void bw_align(bitw_t *w) { w->nbits = (w->nbits + 7) & ~(uint64_t)7; }

/* ------------------------------------------------------------------------------------------
 * Encoder settings per level (libFLAC's published presets; levels 6-8 use tukey(0.5) instead
 * of subdivide_tukey -- documented divergence)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int blocksize;
    int max_lpc_order;
    int max_porder;
    int qlp_precision;
} enc_params_t;

static enc_params_t level_params(uint32_t level) {
    enc_params_t p;
    static const int lpc[9] = {0, 0, 0, 6, 8, 8, 8, 12, 12};
    static const int po[9] = {3, 3, 3, 4, 4, 5, 6, 6, 6};
    p.blocksize = (level <= 2) ? 1152 : 4096;
    p.max_lpc_order = lpc[level];
    p.max_porder = po[level];
    /* libFLAC auto precision for bps > 16: <=384:13, <=1152:14, else 15 */
    p.qlp_precision = (p.blocksize <= 384) ? 13 : (p.blocksize <= 1152 ? 14 : 15);
    return p;
}

/* tukey(0.5) window of length L, as float */
void oracle_tukey_window(int L, float *w) {
    for (int n = 0; n < L; ++n) w[n] = 1.0f;
    int Np = (int)(0.25f * (float)L) - 1;
    if (Np > 0) {
        for (int n = 0; n <= Np; ++n) {
            w[n] = (float)(0.5 - 0.5 * cos(3.14159265358979323846 * (double)n / (double)Np));
            w[L - Np - 1 + n] = (float)(0.5 - 0.5 * cos(3.14159265358979323846 * (double)(n + Np) / (double)Np));
        }
    }
}

/* deterministic log2 using only IEEE +,-,*,/ (same code runs on the GPU) */
double oracle_det_log2(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &b, 8);
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
    return (double)e + (2.8853900817779268 * s) * p; /* 2/ln(2) */
}

static int ilog2_u64(uint64_t v) { /* floor(log2(v)), v > 0 */
    int n = 0;
    while (v >>= 1) n++;
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Partitioned-Rice parameter search on integer partition sums
 * (after libFLAC find_best_partition_order_/set_partitioned_rice_/count_rice_bits_in_partition_)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int porder;
    int params[64];
    uint64_t est_bits; /* estimate incl. 6 bits of method+order */
} rice_choice_t;

static int max_porder_for(int bs, int level_max, int pred_order) {
    int p = 0;
    int b = bs;
    while (!(b & 1) && p < 15) { p++; b >>= 1; }
    if (p > level_max) p = level_max;
    /* geometry rule shared with the GPU kernel: a partition is a whole number of 64-sample chunks */
    while (p > 0 && ((bs >> p) % 64) != 0) p--;
    while (p > 0 && (bs >> p) <= pred_order) p--;
    return p;
}

static bool rice_search(const int32_t *res, int bs, int pred_order, int level_max_porder, rice_choice_t *out) {
    int pmax = max_porder_for(bs, level_max_porder, pred_order);
    uint64_t S[7][64];
    int ps = bs >> pmax;
    for (int p = 0; p < (1 << pmax); ++p) {
        uint64_t s = 0;
        int lo = p * ps, hi = lo + ps;
        if (lo < pred_order) lo = pred_order;
        for (int i = lo; i < hi; ++i) s += (uint64_t)(res[i] < 0 ? -(int64_t)res[i] : (int64_t)res[i]);
        S[pmax][p] = s;
    }
    for (int po = pmax - 1; po >= 0; --po)
        for (int p = 0; p < (1 << po); ++p) S[po][p] = S[po + 1][2 * p] + S[po + 1][2 * p + 1];

    bool have = false;
    uint64_t best = 0;
    for (int po = pmax; po >= 0; --po) {
        int psz = bs >> po;
        uint64_t bits = 6;
        int params[64];
        bool ok = true;
        for (int p = 0; p < (1 << po); ++p) {
            int n = psz;
            if (p == 0) {
                if (n <= pred_order) { ok = false; break; }
                n -= pred_order;
            }
            uint64_t fpd = 0x40000u / (uint32_t)n;
            uint64_t mean = S[po][p];
            int k;
            if (mean < 2 || (((mean - 1) * fpd) >> 18) == 0) k = 0;
            else k = ilog2_u64(((mean - 1) * fpd) >> 18) + 1;
            if (k >= RICE_LIMIT) k = RICE_LIMIT - 1;
            uint64_t pb = 4 + (uint64_t)(1 + k) * (uint64_t)n + (k ? (mean >> (k - 1)) : (mean << 1)) - (uint64_t)(n >> 1);
            if (pb > 0xffffffffULL) pb = 0xffffffffULL;
            bits += pb;
            if (bits > 0xffffffffULL) bits = 0xffffffffULL;
            params[p] = k;
        }
        if (!ok) break;
        if (!have || bits < best) {
            have = true;
            best = bits;
            out->porder = po;
            out->est_bits = bits;
            memcpy(out->params, params, sizeof(int) * (size_t)(1 << po));
        }
    }
    return have;
}

/* exact residual-section bits (method+order+params+codes) and the row-cap test */
static uint64_t rice_exact_bits(const int32_t *res, int bs, int pred_order, const rice_choice_t *rc, bool *row_overflow) {
    int ps = bs >> rc->porder;
    bool rice2 = false;
    for (int p = 0; p < (1 << rc->porder); ++p)
        if (rc->params[p] >= 15) rice2 = true;
    int plen = rice2 ? 5 : 4;
    uint64_t total = 6;
    uint64_t row_bits = 0;
    *row_overflow = false;
    for (int i = pred_order; i < bs; ++i) {
        if ((i % ROW_SAMPLES) == 0) row_bits = 0;
        int p = i / ps;
        int k = rc->params[p];
        uint32_t u = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
        uint64_t b = (uint64_t)(u >> k) + 1 + (uint64_t)k;
        if (i == (p == 0 ? pred_order : p * ps)) b += (uint64_t)plen;
        total += b;
        row_bits += b;
        if (row_bits > ROW_CAP_BITS) *row_overflow = true;
    }
    return total;
}

static void rice_write(bitw_t *w, const int32_t *res, int bs, int pred_order, const rice_choice_t *rc) {
    int ps = bs >> rc->porder;
    bool rice2 = false;
    for (int p = 0; p < (1 << rc->porder); ++p)
        if (rc->params[p] >= 15) rice2 = true;
    int plen = rice2 ? 5 : 4;
    bw_put(w, rice2 ? 1 : 0, 2);
    bw_put(w, (uint64_t)rc->porder, 4);
    for (int i = pred_order; i < bs; ++i) {
        int p = i / ps;
        int k = rc->params[p];
        if (i == (p == 0 ? pred_order : p * ps)) bw_put(w, (uint64_t)k, (unsigned)plen);
        uint32_t u = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
        bw_zeros(w, u >> k);
        bw_put(w, 1, 1);
        bw_put(w, u & ((k ? (1u << k) : 1u) - 1u), (unsigned)k);
    }
}

/* ------------------------------------------------------------------------------------------
 * LPC analysis (double, fixed operation order)
 * ---------------------------------------------------------------------------------------- */
static void autocorr_lanes(const int32_t *x, const float *win, int bs, int nlag, double *autoc) {
    static __thread double d[MAX_BLOCK];
    double part[64][MAX_ORDER + 1];
    for (int i = 0; i < bs; ++i) d[i] = (double)x[i] * (double)win[i];
    /* partial sum l runs over samples [32 l, 32 l + 32) and then [2048 + 32 l, 2048 + 32 l + 32), as far as they lie
     * below bs: the order in which the GPU kernels (one lane per partial sum, frame image split into two halves)
     * accumulate.  Floating-point rounding makes this order part of the encoder specification. */
    for (int l = 0; l < 64; ++l) {
        for (int lag = 0; lag < nlag; ++lag) part[l][lag] = 0.0;
        for (int piece = 0; piece < 2; ++piece) {
            int lo = 2048 * piece + 32 * l, hi = lo + 32;
            if (hi > bs) hi = bs;
            for (int i = lo; i < hi; ++i)
                for (int lag = 0; lag < nlag; ++lag)
                    if (i >= lag) part[l][lag] = fma(d[i], d[i - lag], part[l][lag]);
        }
    }
    for (int off = 1; off < 64; off <<= 1) {
        for (int lag = 0; lag < nlag; ++lag) {
            double t[64];
            for (int l = 0; l < 64; ++l) t[l] = part[l][lag] + part[l ^ off][lag];
            for (int l = 0; l < 64; ++l) part[l][lag] = t[l];
        }
    }
    for (int lag = 0; lag < nlag; ++lag) autoc[lag] = part[0][lag];
}

/* Levinson-Durbin; returns usable max order; coef[o-1][0..o-1] are the order-o predictor
 * coefficients as float; err[o-1] the residual energy */
static int levinson(const double *autoc, int max_order, float coef[][MAX_ORDER], double *err_out) {
    double lpc[MAX_ORDER];
    double err = autoc[0];
    for (int i = 0; i < max_order; ++i) {
        double r = -autoc[i + 1];
        for (int j = 0; j < i; ++j) r = r - lpc[j] * autoc[i - j];
        r = r / err;
        lpc[i] = r;
        int j;
        for (j = 0; j < (i >> 1); ++j) {
            double tmp = lpc[j];
            lpc[j] = lpc[j] + r * lpc[i - 1 - j];
            lpc[i - 1 - j] = lpc[i - 1 - j] + r * tmp;
        }
        if (i & 1) lpc[j] = lpc[j] + lpc[j] * r;
        err = err * (1.0 - r * r);
        for (j = 0; j <= i; ++j) coef[i][j] = (float)(-lpc[j]);
        err_out[i] = err;
        if (err == 0.0) return i + 1;
    }
    return max_order;
}

static int best_lpc_order(const double *err, int max_order, int total_samples, int overhead_bits) {
    double error_scale = 0.5 / (double)total_samples;
    int best_index = 0;
    double best_bits = 4294967295.0;
    for (int indx = 0; indx < max_order; ++indx) {
        int order = indx + 1;
        double bps;
        if (err[indx] > 0.0) {
            bps = 0.5 * oracle_det_log2(error_scale * err[indx]);
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

/* returns 0 on success */
static int quantize_coefs(const float *c, int order, int precision, int32_t *q, int *shift) {
    precision--;
    int32_t qmax = (1 << precision) - 1, qmin = -(1 << precision);
    double cmax = 0.0;
    for (int i = 0; i < order; ++i) {
        double d = fabs((double)c[i]);
        if (d > cmax) cmax = d;
    }
    if (cmax <= 0.0) return 2;
    /* log2cmax = floor(log2(cmax)) from the exponent field (cmax is a normal float value) */
    uint64_t b;
    memcpy(&b, &cmax, 8);
    int log2cmax = (int)((b >> 52) & 0x7ff) - 1023;
    int sh = precision - log2cmax - 1;
    if (sh > 15) sh = 15;
    else if (sh < -16) return 1;
    double error = 0.0;
    if (sh >= 0) {
        for (int i = 0; i < order; ++i) {
            error = error + (double)c[i] * (double)(1 << sh);
            double rq = (error >= 0.0) ? floor(error + 0.5) : ceil(error - 0.5);
            if (rq > (double)qmax) rq = (double)qmax;
            else if (rq < (double)qmin) rq = (double)qmin;
            error = error - rq;
            q[i] = (int32_t)rq;
        }
        *shift = sh;
    } else {
        int nshift = -sh;
        for (int i = 0; i < order; ++i) {
            error = error + (double)c[i] / (double)(1 << nshift);
            double rq = (error >= 0.0) ? floor(error + 0.5) : ceil(error - 0.5);
            if (rq > (double)qmax) rq = (double)qmax;
            else if (rq < (double)qmin) rq = (double)qmin;
            error = error - rq;
            q[i] = (int32_t)rq;
        }
        *shift = 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * One frame
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t type; /* 0 const, 1 verbatim, 2 fixed, 3 lpc */
    int32_t order;
    int32_t porder;
    int32_t wasted;
    int32_t shift;
    int32_t precision;
    int32_t nbytes;
    int32_t blocksize;
} oracle_frame_info;

static int blocksize_code(int bs) {
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

static void put_utf8(bitw_t *w, uint64_t v) {
    if (v < 0x80) bw_put(w, v, 8);
    else if (v < 0x800) { bw_put(w, 0xC0 | (v >> 6), 8); bw_put(w, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x10000) { bw_put(w, 0xE0 | (v >> 12), 8); bw_put(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_put(w, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x200000) { bw_put(w, 0xF0 | (v >> 18), 8); bw_put(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_put(w, 0x80 | (v & 0x3F), 8); }
    else if (v < 0x4000000) { bw_put(w, 0xF8 | (v >> 24), 8); bw_put(w, 0x80 | ((v >> 18) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_put(w, 0x80 | (v & 0x3F), 8); }
    else { bw_put(w, 0xFC | (v >> 30), 8); bw_put(w, 0x80 | ((v >> 24) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 18) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 12) & 0x3F), 8); bw_put(w, 0x80 | ((v >> 6) & 0x3F), 8); bw_put(w, 0x80 | (v & 0x3F), 8); }
}

/* One subframe: samples xin[0], xin[stride], ... (stride 2 picks one channel of an interleaved pair).
 * bps_base: 32, or 33 for a side channel (its values are required to fit 32 bits here, see encode_frame; only the field
 * widths know about the 33rd bit).  w == NULL: analysis only.  Returns the estimated bits of the subframe the analysis
 * settles on (the quantity candidates are compared by: encode_frame's stereo decision). */
static uint64_t encode_subframe(bitw_t *w, const int32_t *xin_s, int stride, int bs, const enc_params_t *P, const float *win,
                                oracle_frame_info *info, int bps_base) {
    static __thread int32_t xin[MAX_BLOCK], x[MAX_BLOCK], rfix[MAX_BLOCK], rlpc[MAX_BLOCK];
    for (int i = 0; i < bs; ++i) xin[i] = xin_s[(size_t)i * (size_t)stride];

    /* ---- wasted bits ---- */
    uint32_t orv = 0;
    for (int i = 0; i < bs; ++i) orv |= (uint32_t)xin[i];
    int wasted = 0;
    if (orv) while (!((orv >> wasted) & 1)) wasted++;
    for (int i = 0; i < bs; ++i) x[i] = xin[i] >> wasted;
    int bps = bps_base - wasted;

    /* ---- candidates ---- */
    uint64_t verbatim_bits = 8 + (uint64_t)wasted + (uint64_t)bs * (uint64_t)bps;
    bool is_const = true;
    for (int i = 1; i < bs; ++i)
        if (x[i] != x[0]) { is_const = false; break; }

    int type = 1, order = 0, shift = 0, precision = 0;
    int32_t qcoef[MAX_ORDER];
    rice_choice_t rc, rc_fix, rc_lpc;
    const int32_t *res = NULL;
    memset(&rc, 0, sizeof rc);
    uint64_t best_bits = verbatim_bits;

    if (is_const) {
        type = 0;
        best_bits = 8 + (uint64_t)wasted + (uint64_t)bps;
    } else if (bs > 4) {
        /* fixed predictors, orders 0..4: total |e_k| over i >= k, invalid if any |e_k| > INT32_MAX */
        uint64_t tot[5] = {0, 0, 0, 0, 0};
        bool valid[5] = {true, true, true, true, true};
        for (int i = 0; i < bs; ++i) {
            int64_t e[5];
            e[0] = x[i];
            e[1] = (i >= 1) ? (int64_t)x[i] - x[i - 1] : 0;
            e[2] = (i >= 2) ? (int64_t)x[i] - 2 * (int64_t)x[i - 1] + x[i - 2] : 0;
            e[3] = (i >= 3) ? (int64_t)x[i] - 3 * (int64_t)x[i - 1] + 3 * (int64_t)x[i - 2] - x[i - 3] : 0;
            e[4] = (i >= 4) ? (int64_t)x[i] - 4 * (int64_t)x[i - 1] + 6 * (int64_t)x[i - 2] - 4 * (int64_t)x[i - 3] + x[i - 4] : 0;
            for (int k = 0; k < 5; ++k) {
                uint64_t a = (uint64_t)(e[k] < 0 ? -e[k] : e[k]);
                tot[k] += a;
                if (a > 2147483647ULL) valid[k] = false;
            }
        }
        int fo = -1;
        uint64_t smallest = UINT64_MAX;
        for (int k = 0; k < 5; ++k)
            if (valid[k] && tot[k] < smallest) { fo = k; smallest = tot[k]; }
        if (fo >= 0) {
            for (int i = 0; i < bs; ++i) {
                int64_t e;
                if (i < fo) e = x[i];
                else switch (fo) {
                    case 0: e = x[i]; break;
                    case 1: e = (int64_t)x[i] - x[i - 1]; break;
                    case 2: e = (int64_t)x[i] - 2 * (int64_t)x[i - 1] + x[i - 2]; break;
                    case 3: e = (int64_t)x[i] - 3 * (int64_t)x[i - 1] + 3 * (int64_t)x[i - 2] - x[i - 3]; break;
                    default: e = (int64_t)x[i] - 4 * (int64_t)x[i - 1] + 6 * (int64_t)x[i - 2] - 4 * (int64_t)x[i - 3] + x[i - 4]; break;
                }
                rfix[i] = (int32_t)e;
            }
            if (rice_search(rfix, bs, fo, P->max_porder, &rc_fix)) {
                uint64_t est = 8 + (uint64_t)wasted + (uint64_t)fo * (uint64_t)bps + rc_fix.est_bits;
                if (est < best_bits) { best_bits = est; type = 2; order = fo; rc = rc_fix; res = rfix; }
            }
        }
        /* LPC */
        int mlo = P->max_lpc_order;
        if (mlo > bs - 1) mlo = bs - 1;
        if (mlo > 0) {
            double autoc[MAX_ORDER + 1], err[MAX_ORDER];
            static __thread float coef[MAX_ORDER][MAX_ORDER];
            autocorr_lanes(x, win, bs, mlo + 1, autoc);
            if (autoc[0] != 0.0) {
                int usable = levinson(autoc, mlo, coef, err);
                int prec = P->qlp_precision;
                int lo = best_lpc_order(err, usable, bs, bps + prec);
                if (bps <= 17) {
                    int lim = 32 - bps - ilog2_u64((uint64_t)lo);
                    if (prec > lim) prec = lim;
                }
                int32_t q[MAX_ORDER];
                int sh;
                if (prec >= 2 && quantize_coefs(coef[lo - 1], lo, prec, q, &sh) == 0) {
                    bool ok = true;
                    for (int i = 0; i < bs; ++i) {
                        if (i < lo) { rlpc[i] = x[i]; continue; }
                        int64_t sum = 0;
                        for (int j = 0; j < lo; ++j) sum += (int64_t)q[j] * (int64_t)x[i - 1 - j];
                        int64_t r = (int64_t)x[i] - (sum >> sh);
                        if (r > 2147483647LL || r < -2147483647LL) { ok = false; break; }
                        rlpc[i] = (int32_t)r;
                    }
                    if (ok && rice_search(rlpc, bs, lo, P->max_porder, &rc_lpc)) {
                        uint64_t est = 8 + (uint64_t)wasted + 4 + 5 + (uint64_t)lo * (uint64_t)(prec + bps) + rc_lpc.est_bits;
                        if (est < best_bits) {
                            best_bits = est; type = 3; order = lo; rc = rc_lpc; res = rlpc;
                            shift = sh; precision = prec;
                            memcpy(qcoef, q, sizeof(int32_t) * (size_t)lo);
                        }
                    }
                }
            }
        }
        /* exact size of the winner; fall back to VERBATIM when it is larger or a row is too long */
        if (type >= 2) {
            bool rowov;
            uint64_t exact = 8 + (uint64_t)wasted + (uint64_t)order * (uint64_t)bps + rice_exact_bits(res, bs, order, &rc, &rowov);
            if (type == 3) exact += 4 + 5 + (uint64_t)order * (uint64_t)precision;
            if (rowov || exact > verbatim_bits) type = 1;
        }
    }

    if (!w) return best_bits;

    /* ---- subframe ---- */
    static const int type_code[4] = {0x00, 0x01, 0x08, 0x20};
    int tc = type_code[type];
    if (type == 2) tc |= order;
    if (type == 3) tc |= (order - 1);
    bw_put(w, 0, 1);
    bw_put(w, (uint64_t)tc, 6);
    bw_put(w, wasted ? 1 : 0, 1);
    if (wasted) { bw_zeros(w, (uint64_t)(wasted - 1)); bw_put(w, 1, 1); }
    uint64_t mask = (1ULL << bps) - 1;  /* bps <= 33; the values are sign-extended into the field */
    if (type == 0) {
        bw_put(w, (uint64_t)(int64_t)x[0] & mask, (unsigned)bps);
    } else if (type == 1) {
        for (int i = 0; i < bs; ++i) bw_put(w, (uint64_t)(int64_t)x[i] & mask, (unsigned)bps);
    } else {
        for (int i = 0; i < order; ++i) bw_put(w, (uint64_t)(int64_t)x[i] & mask, (unsigned)bps);
        if (type == 3) {
            bw_put(w, (uint64_t)(precision - 1), 4);
            bw_put(w, (uint64_t)shift, 5);
            for (int j = 0; j < order; ++j) bw_put(w, (uint64_t)(uint32_t)qcoef[j] & ((1u << precision) - 1), (unsigned)precision);
        }
        rice_write(w, res, bs, order, &rc);
    }
    if (info) {
        info->type = type;
        info->order = (type >= 2) ? order : 0;
        info->porder = (type >= 2) ? rc.porder : 0;
        info->wasted = wasted;
        info->shift = (type == 3) ? shift : 0;
        info->precision = (type == 3) ? precision : 0;
        info->blocksize = bs;
    }
    return best_bits;
}

/* One frame of nch (1 or 2) channels; xin is sample-interleaved for nch == 2 (channel 0 = the low
 * 32 bits of the reference's int64 samples, utils.c:96-107).
 *
 * Stereo decision (libFLAC tries left/right, left/side, side/right and mid/side on the reference's two-channel path,
 * compress.c:482-540).  On (low word, high word) pairs left/side and mid/side never pay: the side channel costs what
 * the low word costs and the high word is the cheap one (tools/stereo_estimate.py).  side/right can: for small values
 * of both signs the high word is 0 or -1 and side = low - high pulls the negative values one step towards zero.  So:
 * the first channel is coded as SIDE (assignment 0b1001) when the high word is not zero throughout, every side value
 * fits 32 bits (the 33rd bit then is a sign extension: only the field widths change), the low word stays below
 * STEREO_SMALL in magnitude (the gain is 10 % for values of a bit or two, 0.5 % at sigma 16, nothing from sigma 128 on,
 * while the trial doubles the analysis), and the analysis estimates fewer bits for it than for the low word; otherwise
 * the channels are independent (0b0001). */
#define STEREO_SMALL 256
static void encode_frame(bitw_t *w, const int32_t *xin, int nch, int bs, uint64_t frame_no, const enc_params_t *P,
                         const float *win, oracle_frame_info *info) {
    size_t frame_start = (size_t)(w->nbits >> 3);
    static __thread int32_t side[MAX_BLOCK];
    bool use_side = false;
    if (nch == 2) {
        bool fits = true, right_zero = true, small = true;
        for (int i = 0; i < bs; ++i) {
            const int64_t d = (int64_t)xin[2 * i] - (int64_t)xin[2 * i + 1];
            if (xin[2 * i + 1] != 0) right_zero = false;
            if (xin[2 * i] >= STEREO_SMALL || xin[2 * i] <= -STEREO_SMALL) small = false;
            if (d > 2147483647LL || d < -2147483648LL) { fits = false; break; }
            side[i] = (int32_t)d;
        }
        if (fits && small && !right_zero) {
            const uint64_t est_left = encode_subframe(NULL, xin, 2, bs, P, win, NULL, 32);
            const uint64_t est_side = encode_subframe(NULL, side, 1, bs, P, win, NULL, 33);
            use_side = est_side < est_left;
        }
    }

    /* ---- frame header (RFC 9639 9.1) ---- */
    int bsc = blocksize_code(bs);
    bw_put(w, 0xFFF8, 16);
    bw_put(w, (uint64_t)bsc, 4);
    bw_put(w, 9, 4);  /* 44.1 kHz: libFLAC default sample rate, the reference never sets one */
    bw_put(w, use_side ? 9u : (uint64_t)(nch - 1), 4);  /* mono / two independent channels / side + right */
    bw_put(w, 7, 3);  /* 32 bits per sample */
    bw_put(w, 0, 1);
    put_utf8(w, frame_no);
    if (bsc == 6) bw_put(w, (uint64_t)(bs - 1), 8);
    else if (bsc == 7) bw_put(w, (uint64_t)(bs - 1), 16);
    bw_put(w, crc8(w->buf + frame_start, (size_t)(w->nbits >> 3) - frame_start), 8);

    for (int c = 0; c < nch; ++c) {
        if (c == 0 && use_side) encode_subframe(w, side, 1, bs, P, win, info ? &info[c] : NULL, 33);
        else encode_subframe(w, xin + c, nch, bs, P, win, info ? &info[c] : NULL, 32);
    }

    bw_align(w);
    bw_reserve(w, 16);
    uint16_t c = crc16(w->buf + frame_start, (size_t)(w->nbits >> 3) - frame_start);
    bw_put(w, c, 16);
    if (info)
        for (int k = 0; k < nch; ++k) info[k].nbytes = (int32_t)((w->nbits >> 3) - frame_start);
}

/* ------------------------------------------------------------------------------------------
 * One stream: "fLaC" + STREAMINFO + SEEKTABLE (one point per frame) + frames
 * ---------------------------------------------------------------------------------------- */
static int64_t stream_header_bytes(int64_t nframes) { return 4 + 4 + 34 + 4 + 18 * nframes; }

static int encode_stream(const int32_t *x, int nch, int64_t n, uint32_t level, uint8_t **out, int64_t *out_bytes,
                         oracle_frame_info *infos) {
    enc_params_t P = level_params(level);
    int B = P.blocksize;
    int64_t nf = (n + B - 1) / B;
    if (18 * nf >= (1 << 24)) return ERROR_ENCODE_PROCESS;
    bitw_t w;
    memset(&w, 0, sizeof w);
    int64_t hb = stream_header_bytes(nf);
    bw_reserve(&w, (uint64_t)hb * 8);
    w.nbits = (uint64_t)hb * 8; /* header is filled in after the frames are sized */
    float *win = (float *)malloc(sizeof(float) * (size_t)B);
    float *win_tail = (float *)malloc(sizeof(float) * (size_t)B);
    int tail_bs = (int)(n - (nf - 1) * B);
    oracle_tukey_window(B, win);
    oracle_tukey_window(tail_bs, win_tail);
    int64_t *foff = (int64_t *)malloc(sizeof(int64_t) * (size_t)nf);
    for (int64_t f = 0; f < nf; ++f) {
        int bs = (f == nf - 1) ? tail_bs : B;
        foff[f] = (int64_t)(w.nbits >> 3) - hb;
        encode_frame(&w, x + f * B * nch, nch, bs, (uint64_t)f, &P, (bs == B) ? win : win_tail, infos ? &infos[f * nch] : NULL);
    }
    if (w.err) { free(w.buf); free(win); free(win_tail); free(foff); return ERROR_ALLOC; }
    /* header */
    uint8_t *h = w.buf;
    memcpy(h, "fLaC", 4);
    h[4] = 0x00; h[5] = 0; h[6] = 0; h[7] = 34; /* STREAMINFO, not last */
    uint8_t *s = h + 8;
    s[0] = (uint8_t)(B >> 8); s[1] = (uint8_t)B; s[2] = (uint8_t)(B >> 8); s[3] = (uint8_t)B;
    memset(s + 4, 0, 6); /* min/max frame size unknown */
    uint64_t ts = ((uint64_t)n < (1ULL << 36)) ? (uint64_t)n : 0;
    /* 20 bits rate (44100) | 3 bits ch-1 | 5 bits bps-1 | 36 bits total samples */
    uint64_t packed = ((uint64_t)44100 << 44) | ((uint64_t)(nch - 1) << 41) | ((uint64_t)31 << 36) | ts;
    for (int i = 0; i < 8; ++i) s[10 + i] = (uint8_t)(packed >> (56 - 8 * i));
    memset(s + 18, 0, 16); /* MD5 not computed */
    uint8_t *t = h + 42;
    uint32_t stl = (uint32_t)(18 * nf);
    t[0] = 0x83; t[1] = (uint8_t)(stl >> 16); t[2] = (uint8_t)(stl >> 8); t[3] = (uint8_t)stl; /* SEEKTABLE, last */
    for (int64_t f = 0; f < nf; ++f) {
        uint8_t *p = t + 4 + 18 * f;
        uint64_t sn = (uint64_t)f * (uint64_t)B, off = (uint64_t)foff[f];
        int bs = (f == nf - 1) ? tail_bs : B;
        for (int i = 0; i < 8; ++i) { p[i] = (uint8_t)(sn >> (56 - 8 * i)); p[8 + i] = (uint8_t)(off >> (56 - 8 * i)); }
        p[16] = (uint8_t)(bs >> 8); p[17] = (uint8_t)bs;
    }
    *out = w.buf;
    *out_bytes = (int64_t)(w.nbits >> 3);
    free(win); free(win_tail); free(foff);
    return ERROR_NONE;
}

/* flacarray.h:209-247 encode_i32 / encode_i64 (+ _threaded); plumbing per compress.c:133-270 */
static int encode_any(const int32_t *data, int nch, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t *n_bytes,
                      int64_t *starts, unsigned char **bytes, int use_threads) {
    if (level > 8) return ERROR_INVALID_LEVEL;
    if (n_stream == 0) return ERROR_ZERO_NSTREAM;
    if (stream_size == 0) return ERROR_ZERO_STREAMSIZE;
    crc_init();
    *n_bytes = 0;
    *bytes = NULL;
    for (int64_t i = 0; i < n_stream; ++i) starts[i] = 0;
    uint8_t **bufs = (uint8_t **)calloc((size_t)n_stream, sizeof(uint8_t *));
    int64_t *sz = (int64_t *)calloc((size_t)n_stream, sizeof(int64_t));
    if (!bufs || !sz) { free(bufs); free(sz); return ERROR_ALLOC; }
    int errors = ERROR_NONE;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : errors) if (use_threads)
    for (int64_t i = 0; i < n_stream; ++i)
        errors |= encode_stream(data + i * stream_size * nch, nch, stream_size, level, &bufs[i], &sz[i], NULL);
    if (errors == ERROR_NONE) {
        int64_t total = 0;
        for (int64_t i = 0; i < n_stream; ++i) { starts[i] = total; total += sz[i]; }
        unsigned char *blob = (unsigned char *)malloc((size_t)total);
        if (!blob) errors |= ERROR_ALLOC;
        else {
            for (int64_t i = 0; i < n_stream; ++i) memcpy(blob + starts[i], bufs[i], (size_t)sz[i]);
            *bytes = blob;
            *n_bytes = total;
        }
    }
    for (int64_t i = 0; i < n_stream; ++i) free(bufs[i]);
    free(bufs);
    free(sz);
    return errors;
}

int oracle_encode_i32(const int32_t *data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t *n_bytes,
                      int64_t *starts, unsigned char **bytes, int use_threads) {
    return encode_any(data, 1, n_stream, stream_size, level, n_bytes, starts, bytes, use_threads);
}
/* int64 samples are two interleaved int32 channels, low word first on this little-endian target
 * (compress.c:482-511 with utils.c:110-123) */
int oracle_encode_i64(const int64_t *data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t *n_bytes,
                      int64_t *starts, unsigned char **bytes, int use_threads) {
    return encode_any((const int32_t *)data, 2, n_stream, stream_size, level, n_bytes, starts, bytes, use_threads);
}

/* per-frame (per-subframe for nch == 2: infos[f * nch + c]) decisions of one stream, for parity debugging */
int oracle_encode_stream_info(const int32_t *data, int64_t stream_size, uint32_t level, oracle_frame_info *infos) {
    crc_init();
    uint8_t *b = NULL;
    int64_t nb = 0;
    int e = encode_stream(data, 1, stream_size, level, &b, &nb, infos);
    free(b);
    return e;
}
int oracle_encode_stream_info_i64(const int64_t *data, int64_t stream_size, uint32_t level, oracle_frame_info *infos) {
    crc_init();
    uint8_t *b = NULL;
    int64_t nb = 0;
    int e = encode_stream((const int32_t *)data, 2, stream_size, level, &b, &nb, infos);
    free(b);
    return e;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------
 * Decoder: any native-FLAC mono stream (all subframe types, wasted bits, Rice/Rice2, escapes,
 * every blocksize/sample-size code, metadata blocks skipped)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *p;
    int64_t nbytes;
    int64_t pos; /* bit position */
    int err;
} bitr_t;

static uint64_t br_get(bitr_t *r, unsigned n) {
    uint64_t v = 0;
    if ((r->pos + n + 7) >> 3 > r->nbytes) { r->err = 1; return 0; }
    while (n) {
        const unsigned bitoff = (unsigned)(r->pos & 7), room = 8 - bitoff;
        const unsigned take = n < room ? n : room;
        const unsigned byte = r->p[r->pos >> 3];
        v = (v << take) | ((byte >> (room - take)) & ((1u << take) - 1u));
        r->pos += take;
        n -= take;
    }
    return v;
}
static int64_t br_get_signed(bitr_t *r, unsigned n) {
    if (n == 0) return 0;
    uint64_t v = br_get(r, n);
    if (n < 64 && (v >> (n - 1))) v |= ~0ULL << n;
    return (int64_t)v;
}
static uint32_t br_unary(bitr_t *r) {
    uint32_t q = 0;
    while (1) {
        if ((r->pos >> 3) >= r->nbytes) { r->err = 1; return 0; }
        const unsigned bitoff = (unsigned)(r->pos & 7);
        const unsigned rest = (unsigned)(r->p[r->pos >> 3] << bitoff) & 0xffu; /* remaining bits, left aligned */
        if (rest) {
            const unsigned z = (unsigned)__builtin_clz(rest) - 24u;
            r->pos += z + 1;
            return q + z;
        }
        q += 8 - bitoff;
        r->pos += 8 - bitoff;
    }
}

static int decode_residual(bitr_t *r, int32_t *res, int bs, int order) {
    int method = (int)br_get(r, 2);
    if (method > 1) return 1;
    int po = (int)br_get(r, 4);
    int plen = method ? 5 : 4;
    int esc = method ? 31 : 15;
    if ((bs >> po) << po != bs && po > 0) return 1;
    int ps = bs >> po;
    int i = order;
    for (int p = 0; p < (1 << po); ++p) {
        int n = (p == 0) ? ps - order : ps;
        if (n < 0) return 1;
        int k = (int)br_get(r, (unsigned)plen);
        if (k == esc) {
            int nb = (int)br_get(r, 5);
            for (int j = 0; j < n; ++j) res[i++] = (int32_t)br_get_signed(r, (unsigned)nb);
        } else {
            for (int j = 0; j < n; ++j) {
                uint32_t q = br_unary(r);
                uint32_t u = (q << k) | (uint32_t)br_get(r, (unsigned)k);
                res[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (r->err) return 1;
    }
    return 0;
}

/* one subframe of `bps` (up to 33: side channels) bits per sample into 64-bit samples */
static int decode_subframe(bitr_t *r, int bs, int bps, int64_t *out) {
    static __thread int32_t res[65536];
    if (br_get(r, 1)) return -1;
    int tc = (int)br_get(r, 6);
    int wasted = 0;
    if (br_get(r, 1)) wasted = (int)br_unary(r) + 1;
    bps -= wasted;
    if (bps <= 0) return -1;
    if (tc == 0) {
        int64_t v = br_get_signed(r, (unsigned)bps);
        for (int i = 0; i < bs; ++i) out[i] = v;
    } else if (tc == 1) {
        for (int i = 0; i < bs; ++i) out[i] = br_get_signed(r, (unsigned)bps);
    } else if (tc >= 8 && tc <= 12) {
        int order = tc - 8;
        if (order > bs) return -1;
        for (int i = 0; i < order; ++i) out[i] = br_get_signed(r, (unsigned)bps);
        if (decode_residual(r, res, bs, order)) return -1;
        for (int i = order; i < bs; ++i) {
            int64_t p;
            switch (order) {
                case 0: p = 0; break;
                case 1: p = out[i - 1]; break;
                case 2: p = 2 * out[i - 1] - out[i - 2]; break;
                case 3: p = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
                default: p = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
            }
            out[i] = p + res[i];
        }
    } else if (tc >= 32) {
        int order = (tc & 31) + 1;
        if (order > bs) return -1;
        for (int i = 0; i < order; ++i) out[i] = br_get_signed(r, (unsigned)bps);
        int prec = (int)br_get(r, 4) + 1;
        if (prec == 16) return -1;
        int sh = (int)br_get_signed(r, 5);
        if (sh < 0) return -1;
        int32_t q[32];
        for (int j = 0; j < order; ++j) q[j] = (int32_t)br_get_signed(r, (unsigned)prec);
        if (decode_residual(r, res, bs, order)) return -1;
        for (int i = order; i < bs; ++i) {
            int64_t sum = 0;
            for (int j = 0; j < order; ++j) sum += (int64_t)q[j] * out[i - 1 - j];
            out[i] = (sum >> sh) + res[i];
        }
    } else {
        return -1;
    }
    if (r->err) return -1;
    if (wasted) for (int i = 0; i < bs; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
    return 0;
}

/* decode one frame at byte offset *off into sample-interleaved int32 (nch channels); returns the
 * blocksize or -1.  Channel assignments: 0 mono, 1 left/right, 8 left/side, 9 side/right,
 * 10 mid/side (RFC 9639 9.1.3; the side channel carries one extra bit). */
static int decode_frame(const uint8_t *s, int64_t nbytes, int64_t *off, int si_bps, int nch, int32_t *out) {
    static __thread int64_t c0[65536], c1[65536];
    bitr_t r = {s, nbytes, (*off) * 8, 0};
    int64_t start = *off;
    if (br_get(&r, 14) != 0x3FFE) return -1;
    if (br_get(&r, 1)) return -1;
    int variable = (int)br_get(&r, 1);
    int bsc = (int)br_get(&r, 4);
    int src = (int)br_get(&r, 4);
    int ch = (int)br_get(&r, 4);
    int ssc = (int)br_get(&r, 3);
    if (br_get(&r, 1)) return -1;
    if (variable) return -1; /* fixed-blocksize streams only */
    if (nch == 1 ? (ch != 0) : !(ch == 1 || (ch >= 8 && ch <= 10))) return -1;
    /* utf-8 number */
    int first = (int)br_get(&r, 8);
    int extra = 0;
    if (first & 0x80) { int m = 0x40; while (first & m) { extra++; m >>= 1; } if (extra == 0 || extra > 6) return -1; }
    for (int i = 0; i < extra; ++i) if ((br_get(&r, 8) & 0xC0) != 0x80) return -1;
    int bs;
    if (bsc == 0) return -1;
    else if (bsc == 1) bs = 192;
    else if (bsc <= 5) bs = 576 << (bsc - 2);
    else if (bsc == 6) bs = (int)br_get(&r, 8) + 1;
    else if (bsc == 7) bs = (int)br_get(&r, 16) + 1;
    else bs = 256 << (bsc - 8);
    if (src == 12) br_get(&r, 8);
    else if (src == 13 || src == 14) br_get(&r, 16);
    else if (src == 15) return -1;
    uint8_t c8 = (uint8_t)br_get(&r, 8);
    if (r.err) return -1;
    if (crc8(s + start, (size_t)((r.pos >> 3) - 1 - start)) != c8) return -1;
    static const int ssbits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ssbits[ssc];
    if (bps == 0) bps = si_bps;
    if (bps < 0) return -1;
    if (nch == 1) {
        if (decode_subframe(&r, bs, bps, c0)) return -1;
        for (int i = 0; i < bs; ++i) out[i] = (int32_t)c0[i];
    } else {
        if (decode_subframe(&r, bs, bps + (ch == 9 ? 1 : 0), c0)) return -1;
        if (decode_subframe(&r, bs, bps + ((ch == 8 || ch == 10) ? 1 : 0), c1)) return -1;
        for (int i = 0; i < bs; ++i) {
            int64_t L, R;
            if (ch == 1) { L = c0[i]; R = c1[i]; }
            else if (ch == 8) { L = c0[i]; R = c0[i] - c1[i]; }
            else if (ch == 9) { R = c1[i]; L = c0[i] + c1[i]; }
            else {
                int64_t mid = (int64_t)((uint64_t)c0[i] << 1) | (c1[i] & 1), side = c1[i];
                L = (mid + side) >> 1;
                R = (mid - side) >> 1;
            }
            out[2 * i] = (int32_t)L;
            out[2 * i + 1] = (int32_t)R;
        }
    }
    r.pos = (r.pos + 7) & ~(int64_t)7;
    uint16_t c16 = (uint16_t)br_get(&r, 16);
    if (r.err) return -1;
    if (crc16(s + start, (size_t)((r.pos >> 3) - 2 - start)) != c16) return -1;
    *off = r.pos >> 3;
    return bs;
}

static int decode_stream(const uint8_t *s, int64_t nbytes, int nch, int64_t stream_size, int64_t first, int64_t n_decode, int32_t *out) {
    if (nbytes < 42 || memcmp(s, "fLaC", 4) != 0) return ERROR_DECODE_INIT;
    int64_t off = 4;
    int si_bps = 0, si_ch = 0, si_bs = 0;
    const uint8_t *seek = NULL;
    int64_t npoints = 0;
    while (1) {
        if (off + 4 > nbytes) return ERROR_DECODE_INIT;
        int last = s[off] >> 7, type = s[off] & 0x7f;
        int64_t len = ((int64_t)s[off + 1] << 16) | ((int64_t)s[off + 2] << 8) | s[off + 3];
        off += 4;
        if (off + len > nbytes) return ERROR_DECODE_INIT;
        if (type == 0 && len >= 34) {
            si_bs = (s[off] == s[off + 2] && s[off + 1] == s[off + 3]) ? ((s[off + 2] << 8) | s[off + 3]) : 0;
            si_ch = ((s[off + 12] >> 1) & 7) + 1;
            si_bps = (((s[off + 12] & 1) << 4) | (s[off + 13] >> 4)) + 1;
        }
        if (type == 3) { seek = s + off; npoints = len / 18; }
        off += len;
        if (last) break;
    }
    if (si_ch != nch) return ERROR_DECODE_INIT;
    static __thread int32_t frame[2 * 65536];
    int64_t done = 0; /* samples seen */
    /* seek (what libFLAC's seek_absolute does for the reference, decompress.c:283): use the seek
     * point of the frame that holds `first` when the table has one */
    if (seek && si_bs > 0 && first >= si_bs) {
        int64_t f0 = first / si_bs;
        if (f0 < npoints) {
            const uint8_t *p = seek + 18 * f0;
            uint64_t sn = 0, so = 0;
            for (int i = 0; i < 8; ++i) { sn = (sn << 8) | p[i]; so = (so << 8) | p[8 + i]; }
            if (sn == (uint64_t)f0 * (uint64_t)si_bs && off + (int64_t)so < nbytes) { off += (int64_t)so; done = f0 * si_bs; }
        }
    }
    while (done < first + n_decode) {
        if (off >= nbytes) return ERROR_DECODE_PROCESS;
        int bs = decode_frame(s, nbytes, &off, si_bps, nch, frame);
        if (bs < 0) return ERROR_DECODE_PROCESS;
        for (int i = 0; i < bs; ++i) {
            int64_t g = done + i;
            if (g >= first && g < first + n_decode)
                for (int c = 0; c < nch; ++c) out[(g - first) * nch + c] = frame[i * nch + c];
        }
        done += bs;
    }
    (void)stream_size;
    return ERROR_NONE;
}

/* flacarray.h:249-271 decode_i32 / decode_i64; semantics per decompress.c:194-313 */
static int decode_any(const unsigned char *bytes, const int64_t *starts, const int64_t *nbytes, int nch, int64_t n_stream,
                      int64_t stream_size, int64_t first_sample, int64_t last_sample, int32_t *data, int use_threads) {
    crc_init();
    int64_t first_decode = 0, n_decode = stream_size;
    if (first_sample >= 0 && last_sample >= 0) {
        if (last_sample > stream_size) return ERROR_DECODE_SAMPLE_RANGE;
        if (first_sample > stream_size - 1) return ERROR_DECODE_SAMPLE_RANGE;
        if (first_sample >= last_sample) return ERROR_DECODE_SAMPLE_RANGE;
        first_decode = first_sample;
        n_decode = last_sample - first_sample;
    }
    int errors = ERROR_NONE;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : errors) if (use_threads)
    for (int64_t i = 0; i < n_stream; ++i)
        errors |= decode_stream(bytes + starts[i], nbytes[i], nch, stream_size, first_decode, n_decode, data + i * n_decode * nch);
    return errors;
}
int oracle_decode_i32(const unsigned char *bytes, const int64_t *starts, const int64_t *nbytes, int64_t n_stream,
                      int64_t stream_size, int64_t first_sample, int64_t last_sample, int32_t *data, int use_threads) {
    return decode_any(bytes, starts, nbytes, 1, n_stream, stream_size, first_sample, last_sample, data, use_threads);
}
int oracle_decode_i64(const unsigned char *bytes, const int64_t *starts, const int64_t *nbytes, int64_t n_stream,
                      int64_t stream_size, int64_t first_sample, int64_t last_sample, int64_t *data, int use_threads) {
    return decode_any(bytes, starts, nbytes, 2, n_stream, stream_size, first_sample, last_sample, (int32_t *)data, use_threads);
}

/* ------------------------------------------------------------------------------------------
 * Float <-> int quantisation, operation by operation after utils.c:160-243 and :350-368.
 * The reference is built for baseline x86-64 (no FMA contraction, SSE scalar float/double);
 * out-of-range (int32_t) casts give INT32_MIN there (cvttsd2si "integer indefinite"), which is
 * what to_i32() reproduces explicitly.
 * ---------------------------------------------------------------------------------------- */
static int32_t to_i32(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}
static int64_t to_i64(double v) {
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)v;
}

int oracle_float32_to_int32(const float *input, int64_t n_stream, int64_t stream_size, const float *quanta,
                            int32_t *output, float *offsets, float *gains) {
    const int32_t flac_max = 2147483647;
    for (int64_t is = 0; is < n_stream; ++is) {
        const float *in = input + is * stream_size;
        float smin = in[0], smax = in[0];
        for (int64_t i = 1; i < stream_size; ++i) {
            float v = in[i];
            if (v < smin) smin = v;
            if (v > smax) smax = v;
        }
        float off = (float)(0.5 * (double)(float)(smin + smax));          /* utils.c:194 */
        float amp;
        if ((float)(smin - off) > (float)(smax - off)) amp = (float)(1.01 * (double)(float)(smin - off));
        else amp = (float)(1.01 * (double)(float)(smax - off));          /* utils.c:198-202 */
        float min_quanta = amp / (float)flac_max;                         /* utils.c:203 */
        float sq = (quanta == NULL) ? min_quanta : quanta[is];
        int64_t nquant = to_i64((double)off / (double)sq);                /* utils.c:221 */
        off = (float)((double)sq * (double)nquant);                       /* utils.c:222 */
        float gain = (sq == 0) ? 1.0f : (float)(1.0 / (double)sq);        /* utils.c:224-230 */
        offsets[is] = off;
        gains[is] = gain;
        int32_t *o = output + is * stream_size;
        for (int64_t i = 0; i < stream_size; ++i) {
            float st = in[i] - off;
            float pr = gain * st;                                          /* float multiply ... */
            double v = (st >= 0) ? (double)pr + 0.5 : (double)pr - 0.5;    /* ... double add  :236-239 */
            o[i] = to_i32(v);
        }
    }
    return ERROR_NONE;
}

void oracle_int32_to_float32(const int32_t *input, int64_t n_stream, int64_t stream_size, const float *offsets,
                             const float *gains, float *output) {
    for (int64_t is = 0; is < n_stream; ++is) {
        float coeff = (float)(1.0 / (double)gains[is]);                   /* utils.c:361 */
        for (int64_t i = 0; i < stream_size; ++i) {
            float prod = coeff * (float)input[is * stream_size + i];
            output[is * stream_size + i] = offsets[is] + prod;            /* utils.c:364 */
        }
    }
}

/* utils.c:245-327 float64_to_int64: all arithmetic in double; the (int64_t) casts truncate toward
 * zero and give INT64_MIN out of range on x86-64 (cvttsd2si), reproduced by to_i64() */
int oracle_float64_to_int64(const double *input, int64_t n_stream, int64_t stream_size, const double *quanta,
                            int64_t *output, double *offsets, double *gains) {
    const int64_t flac_max = 9223372036854775807LL;
    for (int64_t s = 0; s < n_stream; ++s) {
        const double *x = input + s * stream_size;
        double smin = x[0], smax = x[0];
        for (int64_t i = 1; i < stream_size; ++i) {
            if (x[i] < smin) smin = x[i];
            if (x[i] > smax) smax = x[i];
        }
        double off = 0.5 * (smin + smax);
        double amp = ((smin - off) > (smax - off)) ? 1.01 * (smin - off) : 1.01 * (smax - off);
        double min_quanta = amp / (double)flac_max;
        double squanta = quanta ? quanta[s] : min_quanta;
        int64_t nquant = to_i64(off / squanta);
        off = squanta * (double)nquant;
        double gain = (squanta == 0) ? 1.0 : 1.0 / squanta;
        offsets[s] = off;
        gains[s] = gain;
        for (int64_t i = 0; i < stream_size; ++i) {
            double t = x[i] - off;
            output[s * stream_size + i] = (t >= 0) ? to_i64(gain * t + 0.5) : to_i64(gain * t - 0.5);
        }
    }
    return ERROR_NONE;
}

/* utils.c:329-348 int64_to_float64 */
void oracle_int64_to_float64(const int64_t *input, int64_t n_stream, int64_t stream_size, const double *offsets,
                             const double *gains, double *output) {
    for (int64_t s = 0; s < n_stream; ++s) {
        double coeff = 1.0 / gains[s];
        for (int64_t i = 0; i < stream_size; ++i) output[s * stream_size + i] = offsets[s] + coeff * (double)input[s * stream_size + i];
    }
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
