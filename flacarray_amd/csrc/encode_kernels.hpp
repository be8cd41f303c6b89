// encode_kernels.hpp -- HIP kernels of the FLAC encode path for gfx950 (wave64).
//
// Replaces what the reference reaches through libFLAC in encode()/encode_threaded()
// (src/flacarray/libflacarray/compress.c:133-270, 274-435): per-frame predictor search,
// residual, Rice partitioning and the bitstream writer, plus the concatenation of the
// per-stream buffers (compress.c:402-429).
//
//   K3  encode_frames_kernel   one wavefront per frame (<= 4096 samples staged in LDS):
//                              analysis + bit packing into a per-frame scratch slot
//   K4  stream_scan_kernel /   frame sizes -> per-frame offsets, per-stream nbytes, starts
//       starts_scan_kernel
//   K5  write_headers_kernel   "fLaC" + STREAMINFO + SEEKTABLE of every stream
//       compact_frames_kernel  slot -> final blob (byte-shifted copy) + CRC-16 of each frame
//
// LDS image of one frame (20312 B, 8 waves per CU):
//   smp   65 chunks x 68 words   chunk c holds samples [64(c-1), 64c); 4 pad words per chunk make
//                                the lane-per-chunk ds_read_b128 conflict free; chunk 0 is zeros
//   ring  512 words              analysis scratch, then the circular bit buffer of the writer
//   psum  64 x u64, kpar 64 B    Rice partition sums / chosen parameters
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "flac_math.hpp"

namespace fa {

// Translation units (flacarray_amd/build.py): the library is built from two, so that the frame kernels can be
// compiled with the max-ILP scheduling strategy (K3 -4 %, K7 -2 %) without the compaction kernel, which that
// strategy slows by 10 %.  FA_UNIT_COMPACT: this header provides only K5 (csrc/compact_unit.hip);
// FA_SPLIT_UNITS: it provides everything but K5 (csrc/flacarray_hip.hip in the split build); neither: everything.
#ifndef FA_UNIT_COMPACT

struct FrameInfo {  // optional per-frame decision record (parity debugging)
    int32_t type, order, porder, wasted, shift, precision, nbytes, blocksize;
};

struct EncodeArgs {
    const int32_t* data;  // [n_stream][stream_size]
    int64_t n_stream;
    int64_t stream_size;
    int64_t nframes;  // frames per stream
    int32_t B;        // nominal blocksize
    int32_t tail_bs;  // samples in the last frame of a stream
    int32_t max_lpc_order;
    int32_t max_porder;
    int32_t precision;
    const float* win;       // [B]
    const float* win_tail;  // [tail_bs]
    uint8_t* slots;         // [n_stream*nframes][slot_stride]
    int64_t slot_stride;    // kSlotBytes per channel
    uint32_t* frame_bytes;  // [n_stream*nframes]
    FrameInfo* info;        // [n_stream*nframes] or null
    unsigned long long* stamps;  // diagnostic build (-DFA_STAMPS): per-phase cycle sums
    const uint4* hdr;       // [nframes] frame header fields by frame number (see frame_header_entry); two-channel arrays:
                            // [2 nframes], the second half with channel assignment side + right
    int32_t pmax_full, pmax_tail;  // max_porder_for(B / tail_bs, max_porder, 0): the part that does not depend on the predictor order
    double escale_full, escale_tail;  // 0.5 / blocksize (best_lpc_order's error scale), divided once on the host
    int32_t tail_only;  // 1: the grid is one workgroup per STREAM and encodes only its last frame, into slot [stream] (the
                        // single-pass encoder takes the full frames, encode_fused.hpp)
};

// Frame header of frame number f (RFC 9639 9.1) as the fields the preamble writer ORs into the ring.
// It depends on (f, blocksize, channels) only, so the host tabulates it once per encode call instead
// of every wave deriving UTF-8 and CRC-8 with ~400 scalar instructions.
//   x: sync | blocksize code | sample-rate code | channel / sample-size byte (32 bits)
//   y: first (up to 4) bytes of the UTF-8 coded frame number
//   z: remaining UTF-8 bytes (low 16 bits) | explicit blocksize bytes (high 16 bits)
//   w: CRC-8 | bits of y << 8 | bits of the low half of z << 16 | bits of the high half of z << 24
FA_HD uint4 frame_header_entry(uint64_t fn, int bs, int nch, bool side_right = false) {
    const int bsc = blocksize_code(bs);
    const uint32_t b2 = (uint32_t)((bsc << 4) | 9);
    // mono, two independent channels, or side + right (0b1001); 32 bits per sample, reserved 0
    const uint32_t b3 = ((side_right ? 9u : (uint32_t)(nch - 1)) << 4) | 0x0Eu;
    const int nbu = (fn < 0x80) ? 1 : (fn < 0x800) ? 2 : (fn < 0x10000) ? 3 : (fn < 0x200000) ? 4 : (fn < 0x4000000) ? 5 : 6;
    uint64_t ub = fn;  // UTF-8 coded frame number, big-endian in the low nbu bytes
    if (nbu > 1) {
        ub = ((0xFF00u >> nbu) & 0xFFu) | (fn >> (6 * (nbu - 1)));
        for (int i = 1; i < nbu; ++i) ub = (ub << 8) | 0x80u | ((fn >> (6 * (nbu - 1 - i))) & 0x3Fu);
    }
    uint8_t c8 = crc8_byte(crc8_byte(0, 0xFF), 0xF8);
    c8 = crc8_byte(c8, (uint8_t)b2);
    c8 = crc8_byte(c8, (uint8_t)b3);
    for (int i = nbu - 1; i >= 0; --i) c8 = crc8_byte(c8, (uint8_t)(ub >> (8 * i)));
    uint32_t bsv = 0, nb3 = 0;
    if (bsc == 6) { c8 = crc8_byte(c8, (uint8_t)(bs - 1)); bsv = (uint32_t)(bs - 1); nb3 = 8; }
    else if (bsc == 7) { c8 = crc8_byte(c8, (uint8_t)((bs - 1) >> 8)); c8 = crc8_byte(c8, (uint8_t)(bs - 1)); bsv = (uint32_t)(bs - 1); nb3 = 16; }
    const int n1 = nbu > 4 ? 4 : nbu;
    const uint32_t nb1 = 8u * (uint32_t)n1, nb2 = 8u * (uint32_t)(nbu - n1);
    uint4 e;
    e.x = 0xFFF80000u | (b2 << 8) | b3;
    e.y = (uint32_t)(ub >> (8 * (nbu - n1)));
    e.z = ((uint32_t)ub & ((1u << nb2) - 1u)) | (bsv << 16);
    e.w = (uint32_t)c8 | (nb1 << 8) | (nb2 << 16) | (nb3 << 24);
    return e;
}

#ifdef FA_STAMPS
__device__ __forceinline__ unsigned long long fa_memtime() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define FA_STAMP(k)                                              \
    do {                                                         \
        const unsigned long long t_ = fa_memtime();              \
        st_[k] += t_ - t_prev_;                                  \
        t_prev_ = fa_memtime();                                  \
    } while (0)
#define FA_STAMP_INIT                 \
    unsigned long long st_[16];       \
    for (int i_ = 0; i_ < 16; ++i_) st_[i_] = 0; \
    unsigned long long t_prev_ = fa_memtime()
#define FA_STAMP_FLUSH                                                                   \
    do {                                                                                 \
        if (threadIdx.x == 0 && a.stamps && (blockIdx.x & 63) == 0) {                    \
            for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&a.stamps[i_], st_[i_]);           \
            atomicAdd(&a.stamps[16], 1ULL);                                              \
        }                                                                                \
    } while (0)
#elif defined(FA_PHASE_MARKS)
// ISA bookkeeping build (tools/isa_phases.py): every stamp becomes a comment in the assembly, fenced so that no
// instruction moves across it; the tool counts VALU / SALU / LDS / memory instructions between the marks.
#define FA_STAMP(k)                                   \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        asm volatile("; FA_MARK " #k ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)
#define FA_STAMP_INIT FA_STAMP(init)
#define FA_STAMP_FLUSH FA_STAMP(end)
#else
#define FA_STAMP(k) do { } while (0)
#define FA_STAMP_INIT do { } while (0)
#define FA_STAMP_FLUSH do { } while (0)
#endif

constexpr int kChunkStride = 68;
constexpr int kSmpWords = 65 * kChunkStride;  // 4420
constexpr int kRingWords = 512;
constexpr int kRingMask = kRingWords - 1;
// ring[kRingWords] mirrors ring[0]: a code that straddles the end of the ring ORs its second word
// there (no wrap arithmetic per code); whoever reads ring[0] merges and clears the mirror
constexpr int kLdsWords = kSmpWords + kRingWords + 2 + 128 + 16;  // 5078 words = 20312 B (8 per CU = 162.5 KB)

__device__ __forceinline__ int smp_idx(int s) { return kChunkStride * ((s >> 6) + 1) + (s & 63); }

// ---- wave-level helpers (64 lanes), DPP / swizzle based: no LDS traffic, no bpermute ---------
// A single wavefront owns the whole workgroup, so cross-lane LDS hand-offs need only the LDS
// queue drained (DS operations of one wave execute in order); s_barrier / vmcnt drains are
// avoided on purpose: they would stall every row of the writer on its own global stores.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1;    // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;    // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // == xor 4 once quads are uniform
constexpr int kDppMirror = 0x140;      // == xor 8 once 8-lane groups are uniform
constexpr int kSwzXor16 = 0x401F;      // ds_swizzle bit mode: and 0x1f, or 0, xor 0x10

template <int STEP>
__device__ __forceinline__ int xchg_i32(int v) {
    if constexpr (STEP == 0) return dpp_i32<kDppXor1>(v);
    else if constexpr (STEP == 1) return dpp_i32<kDppXor2>(v);
    else if constexpr (STEP == 2) return dpp_i32<kDppHalfMirror>(v);
    else if constexpr (STEP == 3) return dpp_i32<kDppMirror>(v);
    else return __builtin_amdgcn_ds_swizzle(v, kSwzXor16);
}
template <int STEP>
__device__ __forceinline__ double xchg_f64(double v) {
    const int lo = xchg_i32<STEP>(__double2loint(v)), hi = xchg_i32<STEP>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int STEP>
__device__ __forceinline__ uint64_t xchg_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)xchg_i32<STEP>((int)(uint32_t)v), hi = (uint32_t)xchg_i32<STEP>((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}

// xor-butterfly sum: p[l] += p[l^1], ^2, ^4, ^8, ^16, ^32 -- the oracle's summation order
__device__ __forceinline__ double wave_sum_butterfly(double v) {
    v = v + xchg_f64<0>(v);
    v = v + xchg_f64<1>(v);
    v = v + xchg_f64<2>(v);
    v = v + xchg_f64<3>(v);
    v = v + xchg_f64<4>(v);
    return readlane_f64(v, 0) + readlane_f64(v, 32);
}
// N sums at once, step by step: every exchange of a step is issued before the first add that needs one, so the N
// chains overlap (one LDS round trip for all the swizzles of the last step) instead of running one after the other.
// Same order of additions per value as wave_sum_butterfly: bit-identical results.
template <int N>
__device__ __forceinline__ void wave_sum_butterfly_n(double (&v)[N]) {
    double o[N];
#define FA_BF_STEP(S)                                          \
    _Pragma("unroll") for (int i = 0; i < N; ++i) o[i] = xchg_f64<S>(v[i]); \
    _Pragma("unroll") for (int i = 0; i < N; ++i) v[i] = v[i] + o[i];
    FA_BF_STEP(0)
    FA_BF_STEP(1)
    FA_BF_STEP(2)
    FA_BF_STEP(3)
    FA_BF_STEP(4)
#undef FA_BF_STEP
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = readlane_f64(v[i], 0) + readlane_f64(v[i], 32);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    double o;
    o = xchg_f64<0>(v); v = (o > v) ? o : v;
    o = xchg_f64<1>(v); v = (o > v) ? o : v;
    o = xchg_f64<2>(v); v = (o > v) ? o : v;
    o = xchg_f64<3>(v); v = (o > v) ? o : v;
    o = xchg_f64<4>(v); v = (o > v) ? o : v;
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 32);
    return (b > a) ? b : a;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v += xchg_u64<0>(v);
    v += xchg_u64<1>(v);
    v += xchg_u64<2>(v);
    v += xchg_u64<3>(v);
    v += xchg_u64<4>(v);
    return readlane_u64(v, 0) + readlane_u64(v, 32);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= (uint32_t)xchg_i32<0>((int)v);
    v |= (uint32_t)xchg_i32<1>((int)v);
    v |= (uint32_t)xchg_i32<2>((int)v);
    v |= (uint32_t)xchg_i32<3>((int)v);
    v |= (uint32_t)xchg_i32<4>((int)v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, xchg_i32<0>(v));
    v = min(v, xchg_i32<1>(v));
    v = min(v, xchg_i32<2>(v));
    v = min(v, xchg_i32<3>(v));
    v = min(v, xchg_i32<4>(v));
    return min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, xchg_i32<0>(v));
    v = max(v, xchg_i32<1>(v));
    v = max(v, xchg_i32<2>(v));
    v = max(v, xchg_i32<3>(v));
    v = max(v, xchg_i32<4>(v));
    return max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
}
// inclusive prefix sum across the wave (row_shr scans inside each 16-lane row, then row_bcast)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// |a - b| + c on unsigned operands in ONE instruction.  The generic __usad is rewritten by the
// optimiser into min / max / subtract / add (four instructions) when its operands are sign-biased
// values, which is exactly how the fixed-predictor loop uses it.
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// OR the low n bits (1..32) of v into the circular bit buffer at absolute bit position pos
__device__ __forceinline__ void ring_put(uint32_t* ring, uint32_t pos, uint32_t v, int n) {
    uint32_t w = pos >> 5;
    int off = (int)(pos & 31);
    uint64_t t = (((uint64_t)v) << (64 - n)) >> off;
    uint32_t hi = (uint32_t)(t >> 32), lo = (uint32_t)t;
    if (hi) atomicOr(&ring[w & kRingMask], hi);
    if (lo) atomicOr(&ring[(w + 1) & kRingMask], lo);
}

// inclusive prefix sum of a u64 across the wave (row_shr scans + row broadcasts, per 32-bit half
// with carry handled by doing the scan on the two halves of an exact double is not possible for
// u64, so the 64-bit add is explicit)
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
#define FA_SCAN_STEP(CTRL, RM, BC)                                                                         \
    {                                                                                                      \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, RM, 0xF, BC);        \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, RM, 0xF, BC); \
        v += ((uint64_t)hi_ << 32) | lo_;                                                                  \
    }
    FA_SCAN_STEP(0x111, 0xF, true)
    FA_SCAN_STEP(0x112, 0xF, true)
    FA_SCAN_STEP(0x114, 0xF, true)
    FA_SCAN_STEP(0x118, 0xF, true)
    FA_SCAN_STEP(0x142, 0xA, false)
    FA_SCAN_STEP(0x143, 0xC, false)
#undef FA_SCAN_STEP
    return v;
}
__device__ __forceinline__ uint64_t gather_u64(uint64_t v, int src_lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Rice partition-order search, every order of [po_lo, po_hi] evaluated at once: lane L of the
// layout holds partition p of order po (order po occupies 2^po lanes from 2^(po_hi+1)-2^(po+1)).
// T = inclusive prefix sum over lanes of the per-chunk magnitude sums.  Updates the running best
// (oracle rule: orders from high to low, strictly smaller wins) and the per-partition parameter.
__device__ __forceinline__ void rice_search_batch(uint64_t T, int bs, int pred_order, int po_hi, int po_lo, int lane,
                                                  bool* have, uint64_t* best, int* best_po, int* kbest) {
    const int M = (2 << po_hi) - 1;
    const int Lp = M - lane;                        // >= 1 for lanes that hold a slot
    const int po = (Lp >= 1) ? (31 - __clz(Lp)) : 0;
    const bool slot = (Lp >= 1) && (po >= po_lo);
    const int p = lane - (M + 1 - (2 << po));
    const uint32_t psz = (uint32_t)(bs >> po);
    // chunk lanes covered by the partition: [p*lpp, (p+1)*lpp); order 0 covers the whole frame
    const int lpp = (po == 0) ? 64 : (int)(psz / kChunk);
    int hi_l = (p + 1) * lpp - 1, lo_l = p * lpp - 1;
    hi_l = hi_l > 63 ? 63 : (hi_l < 0 ? 0 : hi_l);
    const uint64_t Th = gather_u64(T, hi_l);
    const uint64_t Tl = gather_u64(T, lo_l < 0 ? 0 : (lo_l > 63 ? 63 : lo_l));
    const uint64_t S = slot ? (Th - ((lo_l >= 0) ? Tl : 0)) : 0;
    uint64_t pb = 0;
    int k = 0;
    if (slot) {
        const uint32_t n = psz - ((p == 0) ? (uint32_t)pred_order : 0u);
        k = rice_param(S, n);
        pb = rice_part_bits(S, n, k);
    }
    const uint64_t SC = wave_incl_scan_u64(pb);
    for (int o = po_hi; o >= po_lo; --o) {
        const int base = M + 1 - (2 << o);
        const int end = base + (1 << o) - 1;
        uint64_t bits = readlane_u64(SC, end) - (base > 0 ? readlane_u64(SC, base - 1) : 0) + 6;
        if (bits > 0xffffffffULL) bits = 0xffffffffULL;
        if (!*have || bits < *best) {
            *have = true;
            *best = bits;
            *best_po = o;
            // parameter of partition `lane` at this order lives in layout lane base + lane
            *kbest = __builtin_amdgcn_ds_bpermute(((base + lane) & 63) << 2, k);
        }
    }
}

// The same search in 32-bit arithmetic, valid when every lane's sum is below 2^24 (so the frame
// total is below 2^30): partition sums, bit estimates (<= S + 2^17 + 4 per partition) and their
// totals then fit 32 bits, and every value equals what the 64-bit version computes.
__device__ __forceinline__ int rice_param_u32(uint32_t mean, uint32_t n) {
    if (mean < 2) return 0;
    const uint32_t fpd = 0x40000u / n;
    const uint32_t m1 = mean - 1;
    const uint32_t v = (__umulhi(m1, fpd) << 14) | ((m1 * fpd) >> 18);  // ((mean-1)*fpd) >> 18, below 2^32
    return v ? (32 - __clz((int)v)) : 0;  // <= 30 here, so the Rice2 clamp never applies
}
__device__ __forceinline__ void rice_search_batch_u32(uint32_t T, int bs, int pred_order, int po_hi, int po_lo, int lane,
                                                      bool* have, uint32_t* best, int* best_po, int* kbest) {
    const int M = (2 << po_hi) - 1;
    const int Lp = M - lane;
    const int po = (Lp >= 1) ? (31 - __clz(Lp)) : 0;
    const bool slot = (Lp >= 1) && (po >= po_lo);
    const int p = lane - (M + 1 - (2 << po));
    const uint32_t psz = (uint32_t)(bs >> po);
    const int lpp = (po == 0) ? 64 : (int)(psz / kChunk);
    int hi_l = (p + 1) * lpp - 1, lo_l = p * lpp - 1;
    hi_l = hi_l > 63 ? 63 : (hi_l < 0 ? 0 : hi_l);
    const uint32_t Th = (uint32_t)__builtin_amdgcn_ds_bpermute(hi_l << 2, (int)T);
    const uint32_t Tl = (uint32_t)__builtin_amdgcn_ds_bpermute((lo_l < 0 ? 0 : (lo_l > 63 ? 63 : lo_l)) << 2, (int)T);
    const uint32_t S = slot ? (Th - ((lo_l >= 0) ? Tl : 0u)) : 0u;
    uint32_t pb = 0;
    int k = 0;
    if (slot) {
        const uint32_t n = psz - ((p == 0) ? (uint32_t)pred_order : 0u);
        k = rice_param_u32(S, n);
        pb = 4u + (uint32_t)(1 + k) * n + (k ? (S >> (k - 1)) : (S << 1)) - (n >> 1);
    }
    const uint32_t SC = wave_incl_scan_u32(pb);
    for (int o = po_hi; o >= po_lo; --o) {
        const int base = M + 1 - (2 << o);
        const int end = base + (1 << o) - 1;
        const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)SC, end) -
                              (base > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)SC, base - 1) : 0u) + 6u;
        if (!*have || bits < *best) {
            *have = true;
            *best = bits;
            *best_po = o;
            *kbest = __builtin_amdgcn_ds_bpermute(((base + lane) & 63) << 2, k);
        }
    }
}

// The same 32-bit search for partition orders 0..pmax (pmax <= 5) without a single branch and cut into
// stages, so that the caller can place the stages between groups of independent arithmetic: every
// stage is a short latency chain (DPP scan, cross-lane gathers, lane reads) that then overlaps the
// caller's work instead of stalling the wave.  Requires every lane's sum below 2^24.
struct FastRiceSearch {
    int bs, pred_order, pmax, lane;
    uint32_t T, Th, Tl, S, n, pb, SC, best;
    int M, p, lo_l, k, bpo, kb;
    bool slot;
    __device__ __forceinline__ void start(uint32_t tl, int bs_, int pred_order_, int pmax_, int lane_) {
        bs = bs_; pred_order = pred_order_; pmax = pmax_; lane = lane_;
        T = wave_incl_scan_u32(tl);
    }
    __device__ __forceinline__ void gather() {
        M = (2 << pmax) - 1;
        const int Lp = M - lane;
        slot = (Lp >= 1);
        const int po = slot ? (31 - __clz(Lp)) : 0;
        p = lane - (M + 1 - (2 << po));
        const uint32_t psz = (uint32_t)(bs >> po);
        const int lpp = (po == 0) ? 64 : (int)(psz / kChunk);
        int hi_l = (p + 1) * lpp - 1;
        lo_l = p * lpp - 1;
        hi_l = hi_l > 63 ? 63 : (hi_l < 0 ? 0 : hi_l);
        Th = (uint32_t)__builtin_amdgcn_ds_bpermute(hi_l << 2, (int)T);
        Tl = (uint32_t)__builtin_amdgcn_ds_bpermute((lo_l < 0 ? 0 : (lo_l > 63 ? 63 : lo_l)) << 2, (int)T);
        n = psz - ((p == 0) ? (uint32_t)pred_order : 0u);
        n = slot ? n : 1u;
    }
    __device__ __forceinline__ void params() {  // rice_param_u32 / rice_part_bits, as selects
        S = slot ? (Th - ((lo_l >= 0) ? Tl : 0u)) : 0u;
        const uint32_t fpd = 0x40000u / n;
        const uint32_t m1 = S - 1;
        const uint32_t v = (__umulhi(m1, fpd) << 14) | ((m1 * fpd) >> 18);
        k = (S < 2 || v == 0) ? 0 : (32 - __clz((int)v));
        const uint32_t rest = (k != 0) ? (S >> ((k - 1) & 31)) : (S << 1);
        pb = 4u + (uint32_t)(1 + k) * n + rest - (n >> 1);
        pb = slot ? pb : 0u;
    }
    __device__ __forceinline__ void totals() {
        SC = wave_incl_scan_u32(pb);
        best = 0xffffffffu;
        bpo = 0;
        kb = 0;
    }
    // orders are visited from 5 down to 0; strictly smaller wins, the first valid one always does
    __device__ __forceinline__ void order(int o) {
        const bool valid = (o <= pmax);
        const int base = valid ? (M + 1 - (2 << o)) : 0;
        const int end = valid ? (base + (1 << o) - 1) : 0;
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)SC, end);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)SC, base > 0 ? base - 1 : 0);
        const uint32_t bits = hi - (base > 0 ? lo : 0u) + 6u;
        const int kg = __builtin_amdgcn_ds_bpermute(((base + lane) & 63) << 2, k);
        const bool take = valid && (bits < best);
        best = take ? bits : best;
        bpo = take ? o : bpo;
        kb = take ? kg : kb;
    }
};

// full search for a candidate with per-chunk magnitude sums `tl` (exact, as double or u64)
__device__ __forceinline__ uint64_t rice_search_all(uint64_t tl, int bs, int pred_order, int pmax, int lane, int* best_po,
                                                    int* kbest) {
    *best_po = 0;
    *kbest = 0;
    if (__all(tl < (1u << 24))) {
        const uint32_t T = wave_incl_scan_u32((uint32_t)tl);
        bool have = false;
        uint32_t best = 0;
        if (pmax >= 6) rice_search_batch_u32(T, bs, pred_order, pmax, 6, lane, &have, &best, best_po, kbest);
        rice_search_batch_u32(T, bs, pred_order, pmax > 5 ? 5 : pmax, 0, lane, &have, &best, best_po, kbest);
        return best;
    }
    const uint64_t T = wave_incl_scan_u64(tl);
    bool have = false;
    uint64_t best = 0;
    if (pmax >= 6) rice_search_batch(T, bs, pred_order, pmax, 6, lane, &have, &best, best_po, kbest);
    rice_search_batch(T, bs, pred_order, pmax > 5 ? 5 : pmax, 0, lane, &have, &best, best_po, kbest);
    return best;
}

// Rice partition-order search on the wave: lane p holds the magnitude sum of partition p at
// order pmax.  Returns estimated bits (incl. 6 bits method+order); best order in *best_po;
// *kbest = parameter of partition `lane` at the best order.
__device__ __forceinline__ uint64_t rice_search_wave(uint64_t S, int bs, int pred_order, int pmax, int lane, int* best_po,
                                                     int* kbest) {
    uint64_t best = 0;
    bool have = false;
    int bpo = 0, kb = 0;
    for (int po = pmax; po >= 0; --po) {
        int nparts = 1 << po;
        uint32_t psz = (uint32_t)(bs >> po);
        uint64_t pb = 0;
        int k = 0;
        if (lane < nparts) {
            uint32_t n = psz - ((lane == 0) ? (uint32_t)pred_order : 0u);
            k = rice_param(S, n);
            pb = rice_part_bits(S, n, k);
        }
        uint64_t bits = 6 + wave_sum_u64(pb);
        if (bits > 0xffffffffULL) bits = 0xffffffffULL;
        if (!have || bits < best) {
            have = true;
            best = bits;
            bpo = po;
            kb = k;
        }
        uint64_t a = (uint64_t)__shfl((unsigned long long)S, (2 * lane) & 63, 64);
        uint64_t b = (uint64_t)__shfl((unsigned long long)S, (2 * lane + 1) & 63, 64);
        S = (lane < (nparts >> 1)) ? (a + b) : 0;
    }
    *best_po = bpo;
    *kbest = kb;
    return best;
}

// (Re)load the frame's samples from global memory into the LDS chunk image, applying the
// wasted-bits shift.  Returns this lane's OR of valid samples and whether all equal `first`.
// stride / choff select one channel of a sample-interleaved two-channel frame.
// side: the channel is (word 0 - word 1) of every interleaved pair (stride 2; the caller has checked that it fits 32 bits).
__device__ __forceinline__ void load_frame(const int32_t* __restrict__ src, int bs, int wasted, int32_t* smp, int lane,
                                           uint32_t* orv_out, bool* alleq_out, int32_t first, int stride = 1, int choff = 0,
                                           bool side = false) {
    const bool aligned = (stride == 1) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const int nrows = (bs + kRow - 1) / kRow;
    uint32_t orv = 0;
    bool alleq = true;
    for (int j = 0; j < nrows; ++j) {
        int base = kRow * j + 4 * lane;
        int4 v = make_int4(0, 0, 0, 0);
        if (aligned && base + 3 < bs) {
            v = *reinterpret_cast<const int4*>(src + base);
        } else {
            if (base + 0 < bs) v.x = src[(size_t)(base + 0) * stride + choff];
            if (base + 1 < bs) v.y = src[(size_t)(base + 1) * stride + choff];
            if (base + 2 < bs) v.z = src[(size_t)(base + 2) * stride + choff];
            if (base + 3 < bs) v.w = src[(size_t)(base + 3) * stride + choff];
            if (side) {
                if (base + 0 < bs) v.x -= src[(size_t)(base + 0) * stride + 1];
                if (base + 1 < bs) v.y -= src[(size_t)(base + 1) * stride + 1];
                if (base + 2 < bs) v.z -= src[(size_t)(base + 2) * stride + 1];
                if (base + 3 < bs) v.w -= src[(size_t)(base + 3) * stride + 1];
            }
        }
        if (base + 0 < bs) { orv |= (uint32_t)v.x; alleq = alleq && (v.x == first); }
        if (base + 1 < bs) { orv |= (uint32_t)v.y; alleq = alleq && (v.y == first); }
        if (base + 2 < bs) { orv |= (uint32_t)v.z; alleq = alleq && (v.z == first); }
        if (base + 3 < bs) { orv |= (uint32_t)v.w; alleq = alleq && (v.w == first); }
        v.x >>= wasted; v.y >>= wasted; v.z >>= wasted; v.w >>= wasted;
        *reinterpret_cast<int4*>(&smp[smp_idx(base)]) = v;
    }
    *orv_out = orv;
    *alleq_out = alleq;
}

// ------------------------------------------------------------------------------------------
// K3: one wavefront encodes one frame
// ------------------------------------------------------------------------------------------
// MLO: level's maximum LPC order: 0 (fixed predictors only), 6, 8 or 12.
// NCH: channels per frame.  2 = the reference's int64 arrays (compress.c:482-511): sample-interleaved
// low / high words, coded as two independent subframes (channel assignment 0b0001) one after the
// other into the same bit ring.
constexpr int kStereoSmall = 256;  // |low word| below this in the whole frame: the side + right trial is worth its analysis

#ifndef FA_K3_WAVES_ATTR
#define FA_K3_WAVES_ATTR
#endif
template <int NCH>
constexpr int k3_lds_words() { return kLdsWords + ((NCH == 2) ? 256 : 0); }  // (+ analysis scratch: the ring holds live bits while channel 1 is analysed)

// One frame: frame g of the array (stream g / nframes, frame g % nframes) is analysed and packed into `slot` by the
// calling wavefront; `lds` is the wave's frame image (k3_lds_words<NCH>() words), `lane` the lane number.  Returns the frame's bytes (the
// CRC-16 field, the last two of them, is left zero: K5 / the placement step fills it in).  Callers: the slot kernel
// below (one frame per workgroup) and the placing kernel of encode_placed.hpp (a ticket loop).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// after_staging: called once per frame when the samples of the (first) subframe are in LDS and every load of the staging
// phase has come back -- the placing encoder draws its next ticket there, in front of 10-20 us of LDS and register work.
template <int MLO, int NCH, class Hook = NoHook>
__device__ __forceinline__ uint32_t encode_frame_body(const EncodeArgs& a, const int64_t g, uint8_t* const slot, int32_t* const lds, const int lane,
                                                      Hook&& after_staging = NoHook()) {
    int32_t* smp = lds;
    uint32_t* ring = reinterpret_cast<uint32_t*>(lds + kSmpWords);
    uint64_t* psum = reinterpret_cast<uint64_t*>(lds + kSmpWords + kRingWords + 2);
    uint8_t* kpar = reinterpret_cast<uint8_t*>(lds + kSmpWords + kRingWords + 2 + 128);
    uint32_t* scr = (NCH == 2) ? reinterpret_cast<uint32_t*>(lds + kLdsWords) : ring;
    (void)psum;

    // (the host keeps n_stream * nframes below 2^31: a 32-bit division, not the 64-bit software one)
    const int64_t s = (int64_t)((uint32_t)g / (uint32_t)a.nframes);
    const int64_t f = g - s * a.nframes;
    const int bs = (f == a.nframes - 1) ? a.tail_bs : a.B;
    const int32_t* src = a.data + (s * a.stream_size + f * (int64_t)a.B) * NCH;
    const float* win = (bs == a.B) ? a.win : a.win_tail;
    const int nrows = (bs + kRow - 1) / kRow;
    const bool active = (kChunk * lane < bs);

    FA_STAMP_INIT;
    // frame-level writer state: bit position, flushed 256-byte blocks, the zeroed ring
    uint32_t total_bytes = 0;
    uint32_t pos = 0;
    uint32_t blocks_flushed = 0;
    if constexpr (NCH == 2) {  // (one-channel frames zero the ring when they start to emit: it is analysis scratch before)
        for (int i = lane; i < kRingWords; i += 64) ring[i] = 0;
        if (lane == 0) ring[kRingWords] = 0;
    }
    // frame header fields of this frame number (tabulated by the host) and their bit count
    const uint4 fhe = a.hdr[f];
    const uint32_t fh_bits = 32u + ((fhe.w >> 8) & 0xFFu) + ((fhe.w >> 16) & 0xFFu) + (fhe.w >> 24) + 8u;

    // Two-channel frames, stereo decision (libFLAC tries left/right, left/side, side/right, mid/side on the reference's
    // two-channel path, compress.c:482-540; on (low word, high word) pairs only side/right can pay, tools/stereo_estimate.py):
    // the first channel is coded as SIDE = low - high (assignment 0b1001, a 33-bit channel whose values are required to fit
    // 32 bits, so only field widths change) when the high word is not zero throughout, every difference fits, the low word
    // stays below kStereoSmall in magnitude (the gain is 10 % for values of a bit or two, 0.5 % at sigma 16 and nothing from
    // sigma 128 on, while the trial doubles the analysis: profiles/r03_stereo_estimate.log), and the analysis estimates
    // fewer bits for the side than for the low word.  Passes: 0 = low word, analysis only;
    // 1 = side, analysis only; 2 = the winner, written as channel 0; 3 = the high word, written as channel 1.
    int first_pass = (NCH == 2) ? 2 : 0;
    if constexpr (NCH == 2) {
        // (a probe of one pair per lane first: frames whose low word is not small -- nearly all frames of nearly all
        // arrays -- are done after one load; the full scan runs only for the others)
        auto pair_ok = [&](int lo_w, int hi_w) __attribute__((always_inline)) {
            const int d = (int)((uint32_t)lo_w - (uint32_t)hi_w);
            return (((lo_w ^ hi_w) & (lo_w ^ d)) >= 0)  // no signed overflow in low - high
                   && (lo_w < kStereoSmall) && (lo_w > -kStereoSmall);
        };
        const int probe = (int)(((int64_t)lane * bs) >> 6);
        const int2 pp = *reinterpret_cast<const int2*>(src + 2 * probe);
        if (__all(pair_ok(pp.x, pp.y))) {
            bool fits = true, rnz = false;
#pragma unroll 4
            for (int i = lane; i < bs; i += 64) {
                const int2 pr = *reinterpret_cast<const int2*>(src + 2 * i);
                fits = fits && pair_ok(pr.x, pr.y);
                rnz = rnz || (pr.y != 0);
            }
            if (__all(fits) && __any(rnz)) first_pass = 0;
        }
    }
    uint64_t est_left = 0;
    bool use_side = false;
    uint4 fhe_use = fhe;
#pragma unroll 1
    for (int pass = first_pass; pass < ((NCH == 2) ? 4 : 1); ++pass) {
    const int ch = (pass == 3) ? 1 : 0;
    const bool dry = (NCH == 2) && (pass < 2);
    const bool side = (NCH == 2) && (pass == 1 || (pass == 2 && use_side));
    const uint4 fhe = fhe_use;  // (shadows the frame's entry: side + right frames take the table's second half)
    // writer state at the start of this subframe: a VERBATIM retry rewinds to it.  Everything not yet
    // flushed lies in block bf0 (one word per lane) and, when bf0 opens a ring cycle, the mirror word.
    const uint32_t pos0 = pos, bf0 = blocks_flushed;
    uint32_t save_w = 0, save_m = 0;
    if (NCH == 2 && ch > 0) {
        save_w = ring[(bf0 * 64 + lane) & kRingMask];
        save_m = ring[kRingWords];
    }
    // ---- P0: stage samples, wasted bits, constant test ---------------------------------
    for (int i = lane; i < kChunkStride; i += 64) smp[i] = 0;  // chunk -1 = zero history
    const int32_t first = side ? (int32_t)((uint32_t)src[0] - (uint32_t)src[1]) : src[ch];
    uint32_t orv = 0;
    bool narrow, is_const;
    const bool full = (bs == kMaxBlock) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (full) {
        // all 16 row loads in flight at once (64 KB per CU outstanding at 8 waves); the running
        // minimum / maximum (v_min3 / v_max3: half an instruction per sample each) answer both
        // "constant?" and "narrow?"
        int mn = first, mx = first;
        if constexpr (NCH == 1) {
            int4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const int4*>(src + kRow * j + 4 * lane);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                orv |= (uint32_t)(v[j].x | v[j].y | v[j].z | v[j].w);
                mn = min(min(mn, v[j].x), v[j].y);
                mn = min(min(mn, v[j].z), v[j].w);
                mx = max(max(mx, v[j].x), v[j].y);
                mx = max(max(mx, v[j].z), v[j].w);
                *reinterpret_cast<int4*>(&smp[smp_idx(kRow * j + 4 * lane)]) = v[j];
            }
        } else {
            // interleaved pairs: 8 consecutive words hold this lane's 4 samples of both channels
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                int4 va[8], vb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int32_t* q = src + 2 * (kRow * (8 * half + j) + 4 * lane);
                    va[j] = *reinterpret_cast<const int4*>(q);
                    vb[j] = *reinterpret_cast<const int4*>(q + 4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int4 v = (ch == 0) ? make_int4(va[j].x, va[j].z, vb[j].x, vb[j].z) : make_int4(va[j].y, va[j].w, vb[j].y, vb[j].w);
                    if (side) v = make_int4(v.x - va[j].y, v.y - va[j].w, v.z - vb[j].y, v.w - vb[j].w);
                    orv |= (uint32_t)(v.x | v.y | v.z | v.w);
                    mn = min(min(mn, v.x), v.y);
                    mn = min(min(mn, v.z), v.w);
                    mx = max(max(mx, v.x), v.y);
                    mx = max(max(mx, v.z), v.w);
                    *reinterpret_cast<int4*>(&smp[smp_idx(kRow * (8 * half + j) + 4 * lane)]) = v;
                }
            }
        }
        mn = wave_min_i32(mn);
        mx = wave_max_i32(mx);
        is_const = (mn == mx);
        narrow = (mn >= -(1 << 24)) && (mx < (1 << 24));  // every |x| <= 2^24: fixed-predictor errors fit 32-bit ints
    } else {
        bool alleq = true;
        load_frame(src, bs, 0, smp, lane, &orv, &alleq, first, NCH, ch, side);
        is_const = __all(alleq);
        narrow = false;  // generic path: no narrow shortcut
    }
    orv = wave_or_u32(orv);
    const int wasted = orv ? (__ffs((int)orv) - 1) : 0;
    const int bps = (side ? 33 : 32) - wasted;
    lds_fence();
    if (wasted) {
        for (int j = 0; j < nrows; ++j) {
            int4* p = reinterpret_cast<int4*>(&smp[smp_idx(kRow * j + 4 * lane)]);
            int4 v = *p;
            v.x >>= wasted; v.y >>= wasted; v.z >>= wasted; v.w >>= wasted;
            *p = v;
        }
        lds_fence();
    }

    FA_STAMP(0);
    if (pass == first_pass) after_staging();
    const uint64_t verbatim_bits = 8 + (uint64_t)wasted + (uint64_t)bs * (uint64_t)bps;
    int type = 1;  // 0 const, 1 verbatim, 2 fixed, 3 lpc
    int order = 0, porder = 0, shift = 0, precision = 0;
    int kbest = 0;           // Rice parameter of partition `lane` for the winner
    bool lds_is_residual = false;
    int fo = -1;             // best fixed order
    int32_t qkeep[(MLO > 0) ? MLO : 1];  // quantised LPC coefficients of the winner
#pragma unroll
    for (int j = 0; j < ((MLO > 0) ? MLO : 1); ++j) qkeep[j] = 0;

    uint64_t best_bits = verbatim_bits;  // estimated bits of the subframe the analysis settles on (the stereo decision compares them)
    if (is_const) {
        type = 0;
        best_bits = 8 + (uint64_t)wasted + (uint64_t)bps;
    } else if (bs > 4) {

        // ---- P2: fixed predictors 0..4 over the lane's chunk (exact in double) ---------
        double tot0 = 0.0, tot1 = 0.0, tot2 = 0.0, tot3 = 0.0, tot4 = 0.0;
        double mx0 = 0.0, mx1 = 0.0, mx2 = 0.0, mx3 = 0.0, mx4 = 0.0;
        if (active && narrow) {
            // 32-bit integer path: |e_k| < 2^28, so a 16-sample u32 partial sum cannot overflow and
            // every order is valid.  v_sad_u32 on sign-biased values gives |a - b| + acc in one op.
            const int cbase = kChunkStride * (lane + 1);
            const int4 hh = *reinterpret_cast<const int4*>(&smp[cbase - kChunkStride + 60]);
            const uint32_t BIAS = 0x80000000u;
            int p1 = hh.w;
            int pe1 = hh.w - hh.z;
            int pe2 = pe1 - (hh.z - hh.y);
            int pe3 = pe2 - ((hh.z - hh.y) - (hh.y - hh.x));
            uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
            const int g0 = kChunk * lane;
            // MASK: samples before index k do not count for order k; samples past the end do not count
            auto group = [&](auto mask_tag, int t) __attribute__((always_inline)) {
                constexpr bool MASK = decltype(mask_tag)::value;
                const int4 xv = *reinterpret_cast<const int4*>(&smp[cbase + 4 * t]);
                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = xs[e];
                    const int e1 = x - p1, e2 = e1 - pe1, e3 = e2 - pe2;
                    const uint32_t xb = (uint32_t)x ^ BIAS;
                    const uint32_t n0 = sad_u32(xb, BIAS, s0);
                    const uint32_t n1 = sad_u32(xb, (uint32_t)p1 ^ BIAS, s1);
                    const uint32_t n2 = sad_u32((uint32_t)e1 ^ BIAS, (uint32_t)pe1 ^ BIAS, s2);
                    const uint32_t n3 = sad_u32((uint32_t)e2 ^ BIAS, (uint32_t)pe2 ^ BIAS, s3);
                    const uint32_t n4 = sad_u32((uint32_t)e3 ^ BIAS, (uint32_t)pe3 ^ BIAS, s4);
                    if constexpr (MASK) {
                        const int gi = g0 + 4 * t + e;
                        const bool v = gi < bs;
                        s0 = v ? n0 : s0;
                        s1 = (v && gi >= 1) ? n1 : s1;
                        s2 = (v && gi >= 2) ? n2 : s2;
                        s3 = (v && gi >= 3) ? n3 : s3;
                        s4 = (v && gi >= 4) ? n4 : s4;
                    } else {
                        s0 = n0; s1 = n1; s2 = n2; s3 = n3; s4 = n4;
                    }
                    p1 = x; pe1 = e1; pe2 = e2; pe3 = e3;
                }
            };
            auto fold = [&]() __attribute__((always_inline)) {
                a0 += s0; a1 += s1; a2 += s2; a3 += s3; a4 += s4;
                s0 = s1 = s2 = s3 = s4 = 0;
            };
            group(std::true_type{}, 0);  // the only group that can hold samples 0..3 of the frame
            if (full) {
#pragma unroll
                for (int t = 1; t < 4; ++t) group(std::false_type{}, t);
                fold();
                for (int t4 = 1; t4 < 4; ++t4) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) group(std::false_type{}, 4 * t4 + tt);
                    fold();
                }
            } else {
                // (short frames: rare, kept rolled so that its compare masks do not inflate the SGPR demand)
#pragma unroll 1
                for (int t = 1; t < 16; ++t) {
                    group(std::true_type{}, t);
                    if ((t & 3) == 3) fold();
                }
            }
            tot0 = (double)a0; tot1 = (double)a1; tot2 = (double)a2; tot3 = (double)a3; tot4 = (double)a4;
        } else if (active) {
            const int cbase = kChunkStride * (lane + 1);
            int4 h = *reinterpret_cast<const int4*>(&smp[cbase - kChunkStride + 60]);
            double p1 = (double)h.w;
            double pe1 = (double)h.w - (double)h.z;
            double e1b = (double)h.z - (double)h.y;
            double pe2 = pe1 - e1b;
            double e2b = e1b - ((double)h.y - (double)h.x);
            double pe3 = pe2 - e2b;
            const int g0 = kChunk * lane;
#pragma unroll 1
            for (int t = 0; t < 16; ++t) {
                int4 xv = *reinterpret_cast<const int4*>(&smp[cbase + 4 * t]);
                int xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gi = g0 + 4 * t + e;
                    const double xd = (double)xs[e];
                    const double e1 = xd - p1;
                    const double e2 = e1 - pe1;
                    const double e3 = e2 - pe2;
                    const double e4 = e3 - pe3;
                    const bool v = gi < bs;
                    const double a0 = v ? fa_fabs(xd) : 0.0;
                    const double a1 = (v && gi >= 1) ? fa_fabs(e1) : 0.0;
                    const double a2 = (v && gi >= 2) ? fa_fabs(e2) : 0.0;
                    const double a3 = (v && gi >= 3) ? fa_fabs(e3) : 0.0;
                    const double a4 = (v && gi >= 4) ? fa_fabs(e4) : 0.0;
                    tot0 += a0; tot1 += a1; tot2 += a2; tot3 += a3; tot4 += a4;
                    mx0 = a0 > mx0 ? a0 : mx0;
                    mx1 = a1 > mx1 ? a1 : mx1;
                    mx2 = a2 > mx2 ? a2 : mx2;
                    mx3 = a3 > mx3 ? a3 : mx3;
                    mx4 = a4 > mx4 ? a4 : mx4;
                    p1 = xd; pe1 = e1; pe2 = e2; pe3 = e3;
                }
            }
        }
        FA_STAMP(1);
        {
            const double T0 = wave_sum_butterfly(tot0), T1 = wave_sum_butterfly(tot1), T2 = wave_sum_butterfly(tot2),
                         T3 = wave_sum_butterfly(tot3), T4 = wave_sum_butterfly(tot4);
            double M0 = 0.0, M1 = 0.0, M2 = 0.0, M3 = 0.0, M4 = 0.0;
            if (!narrow) {
                M0 = wave_max_f64(mx0); M1 = wave_max_f64(mx1); M2 = wave_max_f64(mx2); M3 = wave_max_f64(mx3);
                M4 = wave_max_f64(mx4);
            }
            const double lim = 2147483647.0;
            double smallest = 1.8446744073709552e19;  // 2^64, above any total
            if (M0 <= lim && T0 < smallest) { fo = 0; smallest = T0; }
            if (M1 <= lim && T1 < smallest) { fo = 1; smallest = T1; }
            if (M2 <= lim && T2 < smallest) { fo = 2; smallest = T2; }
            if (M3 <= lim && T3 < smallest) { fo = 3; smallest = T3; }
            if (M4 <= lim && T4 < smallest) { fo = 4; smallest = T4; }
        }
        FA_STAMP(2);
        int po_fix = 0, k_fix = 0;
        const double tl_fix = (fo == 0) ? tot0 : (fo == 1) ? tot1 : (fo == 2) ? tot2 : (fo == 3) ? tot3 : tot4;
        const int pmax_geo = (bs == a.B) ? a.pmax_full : a.pmax_tail;  // block size and level only: computed by the host
        auto pmax_for = [&](int pred_order) __attribute__((always_inline)) {
            int pm = pmax_geo;
            while (pm > 0 && (bs >> pm) <= pred_order) pm--;
            return pm;
        };
        const int pmax_fix = pmax_for(fo < 0 ? 0 : fo);
        // Full frames with modest sums (the common case): the fixed predictor's partition search is
        // branch-free and is issued together with the autocorrelation loop below, which hides its
        // scans, gathers and lane reads.  Otherwise it runs here, on its own.
        const bool fuse_search = (MLO > 0) && full && fo >= 0 && pmax_fix <= 5 && a.max_lpc_order > 0 && __all(tl_fix < 16777216.0);
        uint64_t est_fix = 0;
        auto apply_fixed = [&]() __attribute__((always_inline)) {
            if (est_fix < best_bits) {
                best_bits = est_fix;
                type = 2;
                order = fo;
                porder = po_fix;
                kbest = k_fix;
            }
        };
        if (fo >= 0 && !fuse_search) {
            est_fix = 8 + (uint64_t)wasted + (uint64_t)fo * (uint64_t)bps +
                      rice_search_all(active ? (uint64_t)tl_fix : 0, bs, fo, pmax_fix, lane, &po_fix, &k_fix);
            apply_fixed();
            lds_fence();
        }

        FA_STAMP(3);
        // ---- P3: LPC analysis ----------------------------------------------------------
        if constexpr (MLO > 0) {
            int mlo = a.max_lpc_order;
            if (mlo > bs - 1) mlo = bs - 1;
            if (mlo > 0) {
                double acc[MLO + 1];
#pragma unroll
                for (int j = 0; j <= MLO; ++j) acc[j] = 0.0;
                // Lane l accumulates its lag products over samples [32 l, 32 l + 32) and then [2048 + 32 l, 2048 + 32 l + 32)
                // (those below bs): the summation order of the specification, shared with the single-pass kernel
                // (encode_fused.hpp), whose frame image is split that way.  Only these sums round, so only they fix an order.
                if (fuse_search) {
                    // every lane is active; one basic block: 64 samples of lag products + the search
                    FastRiceSearch fs;
                    fs.start((uint32_t)tl_fix, bs, fo, pmax_fix, lane);
#pragma unroll
                    for (int piece = 0; piece < 2; ++piece) {
                        const int g0 = 2048 * piece + 32 * lane;
                        const int pbase = smp_idx(g0);
                        float4 wv[8];
                        float wh[MLO];
#pragma unroll
                        for (int t = 0; t < 8; ++t) wv[t] = *reinterpret_cast<const float4*>(win + g0 + 4 * t);
#pragma unroll
                        for (int j = 0; j < MLO; ++j) wh[j] = (g0 - 1 - j >= 0) ? win[g0 - 1 - j] : 0.0f;
                        double hist[MLO];
#pragma unroll
                        for (int j = 0; j < MLO; ++j) hist[j] = (double)smp[smp_idx(g0 - 1 - j)] * (double)wh[j];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            int4 xv = *reinterpret_cast<const int4*>(&smp[pbase + 4 * t]);
                            const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                            const float ws[4] = {wv[t].x, wv[t].y, wv[t].z, wv[t].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const double d = (double)xs[e] * (double)ws[e];
                                acc[0] = __builtin_fma(d, d, acc[0]);
#pragma unroll
                                for (int j = 0; j < MLO; ++j) acc[j + 1] = __builtin_fma(d, hist[j], acc[j + 1]);
#pragma unroll
                                for (int j = MLO - 1; j > 0; --j) hist[j] = hist[j - 1];
                                hist[0] = d;
                            }
                            // one stage of the search per group of 4 samples (tt is a constant after unrolling)
                            const int tt = 8 * piece + t;
                            if (tt == 1) fs.gather();
                            if (tt == 3) fs.params();
                            if (tt == 5) fs.totals();
                            if (tt >= 7 && tt <= 12) fs.order(12 - tt);
                        }
                    }
                    po_fix = fs.bpo;
                    k_fix = fs.kb;
                    est_fix = 8 + (uint64_t)wasted + (uint64_t)fo * (uint64_t)bps + (uint64_t)fs.best;
                } else {
#pragma unroll 1
                    for (int piece = 0; piece < 2; ++piece) {
                        const int g0 = 2048 * piece + 32 * lane;
                        if (g0 < bs) {
                            const int pbase = smp_idx(g0);
                            float4 wv[8];
                            float wh[MLO];
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                const int gb = g0 + 4 * t;
                                wv[t].x = (gb + 0 < bs) ? win[gb + 0] : 0.0f;
                                wv[t].y = (gb + 1 < bs) ? win[gb + 1] : 0.0f;
                                wv[t].z = (gb + 2 < bs) ? win[gb + 2] : 0.0f;
                                wv[t].w = (gb + 3 < bs) ? win[gb + 3] : 0.0f;
                            }
#pragma unroll
                            for (int j = 0; j < MLO; ++j) wh[j] = (g0 - 1 - j >= 0) ? win[g0 - 1 - j] : 0.0f;
                            double hist[MLO];  // hist[j] = d[i-1-j]
#pragma unroll
                            for (int j = 0; j < MLO; ++j) hist[j] = (double)smp[smp_idx(g0 - 1 - j)] * (double)wh[j];
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                int4 xv = *reinterpret_cast<const int4*>(&smp[pbase + 4 * t]);
                                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                                const float ws[4] = {wv[t].x, wv[t].y, wv[t].z, wv[t].w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    // (a sample at or past bs has window 0: it adds exact zeros and leaves zero history)
                                    const double d = (g0 + 4 * t + e < bs) ? (double)xs[e] * (double)ws[e] : 0.0;
                                    acc[0] = __builtin_fma(d, d, acc[0]);
#pragma unroll
                                    for (int j = 0; j < MLO; ++j) acc[j + 1] = __builtin_fma(d, hist[j], acc[j + 1]);
#pragma unroll
                                    for (int j = MLO - 1; j > 0; --j) hist[j] = hist[j - 1];
                                    hist[0] = d;
                                }
                            }
                        }
                    }
                }
                FA_STAMP(4);
                if (fuse_search) apply_fixed();
                double autoc[MLO + 1];
#pragma unroll
                for (int j = 0; j <= MLO; ++j) autoc[j] = wave_sum_butterfly(acc[j]);

                FA_STAMP(5);
                if (autoc[0] != 0.0) {
                    // scratch: the (still unused) ring area for one-channel frames, its own area otherwise
                    float* coef = reinterpret_cast<float*>(scr);               // MLO*MLO floats
                    double* err = reinterpret_cast<double*>(scr + 160);        // MLO doubles
                    int* meta = reinterpret_cast<int*>(scr + 220);             // usable order
                    if (lane == 0) meta[0] = levinson<MLO>(autoc, mlo, coef, err);
                    lds_fence();
                    FA_STAMP(13);
                    const int usable = meta[0];
                    int prec = a.precision;
                    // order choice: lane o evaluates order o+1 (expected bits from the residual energy),
                    // then the first strictly smaller value wins, exactly as the serial rule
                    int lo;
                    {
                        double mybits = 1e300;
                        if (lane < usable) {
                            const double e = err[lane];
                            const double error_scale = (bs == a.B) ? a.escale_full : a.escale_tail;  // 0.5 / bs
                            double bpsv;
                            if (e > 0.0) {
                                bpsv = 0.5 * det_log2(error_scale * e);
                                if (!(bpsv >= 0.0)) bpsv = 0.0;
                            } else if (e < 0.0) {
                                bpsv = 1e32;
                            } else {
                                bpsv = 0.0;
                            }
                            mybits = bpsv * (double)(bs - (lane + 1)) + (double)((lane + 1) * (bps + prec));
                        }
                        double bestb = 4294967295.0;
                        int bi = 0;
#pragma unroll
                        for (int o = 0; o < MLO; ++o) {
                            const double b = readlane_f64(mybits, o);
                            if (o < usable && b < bestb) { bestb = b; bi = o; }
                        }
                        lo = bi + 1;
                    }
                    FA_STAMP(14);
                    if (bps <= 17) {
                        const int limp = 32 - bps - ilog2_u64((uint64_t)lo);
                        if (prec > limp) prec = limp;
                    }
                    int sh = 0;
                    int32_t qreg[MLO];
#pragma unroll
                    for (int j = 0; j < MLO; ++j) qreg[j] = 0;
                    int ok = 0;
                    if (prec >= 2) ok = (quantize_coefs_t<MLO>(coef + (lo - 1) * MLO, lo, prec, qreg, &sh) == 0) ? 1 : 0;
                    FA_STAMP(6);
                    if (ok) {
                        // coefficients pre-scaled by 2^-sh: products and partial sums keep their significands
                        // (|sum of q*x| < 2^49), so floor(sum of (q 2^-sh) x) == floor((sum of q x) 2^-sh) exactly
                        const double scale = bitsd((uint64_t)(1023 - sh) << 52);  // 2^-sh
                        double qd[MLO];
#pragma unroll
                        for (int j = 0; j < MLO; ++j) qd[j] = (double)qreg[j] * scale;
                        // ---- P4: LPC residual in place + magnitude sums ----------------
                        double tl = 0.0, mxr = 0.0;
                        if (active) {
                            const int cbase = kChunkStride * (lane + 1);
                            const int g0 = kChunk * lane;
                            double hx[MLO];
#pragma unroll
                            for (int j = 0; j < MLO; ++j) hx[j] = (double)smp[cbase - kChunkStride + 63 - j];
                            // MASK: only groups that can hold warm-up samples (the first MLO of the
                            // frame) or reach past the end of a short frame need the per-sample test
                            auto group = [&](auto mask_tag, int t) __attribute__((always_inline)) {
                                constexpr bool MASK = decltype(mask_tag)::value;
                                int4* px = reinterpret_cast<int4*>(&smp[cbase + 4 * t]);
                                const int4 xv = *px;
                                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                                int rs[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const double xd = (double)xs[e];
                                    double sum = 0.0;
#pragma unroll
                                    for (int j = 0; j < MLO; ++j) sum = __builtin_fma(qd[j], hx[j], sum);
                                    const double pred = fa_floor(sum);
                                    const double r = xd - pred;
                                    if constexpr (MASK) {
                                        const int gi = g0 + 4 * t + e;
                                        const bool v = (gi < bs) && (gi >= lo);
                                        const double ar = v ? fa_fabs(r) : 0.0;
                                        tl += ar;
                                        mxr = __builtin_fmax(mxr, ar);
                                        rs[e] = v ? (int)r : xs[e];
                                    } else {
                                        const double ar = fa_fabs(r);
                                        tl += ar;
                                        mxr = __builtin_fmax(mxr, ar);  // one v_max_f64 (|r| as a source modifier) instead of compare + two selects
                                        rs[e] = (int)r;
                                    }
#pragma unroll
                                    for (int j = MLO - 1; j > 0; --j) hx[j] = hx[j - 1];
                                    hx[0] = xd;
                                }
                                *px = make_int4(rs[0], rs[1], rs[2], rs[3]);
                            };
                            constexpr int kWarmGroups = (MLO + 3) / 4;
                            if (full) {
#pragma unroll
                                for (int t = 0; t < kWarmGroups; ++t) group(std::true_type{}, t);
#pragma unroll 4
                                for (int t = kWarmGroups; t < 16; ++t) group(std::false_type{}, t);
                            } else {
#pragma unroll 1
                                for (int t = 0; t < 16; ++t) group(std::true_type{}, t);
                            }
                        }
                        FA_STAMP(7);
                        lds_is_residual = true;
                        const double MX = wave_max_f64(mxr);
                        const int pmax = pmax_for(lo);
                        int po_l = 0, k_l = 0;
                        uint64_t rbits;
                        if (full && pmax <= 5 && __all(tl < 16777216.0)) {
                            // the common case: the straight-line 32-bit search (same values as rice_search_all)
                            FastRiceSearch fs;
                            fs.start((uint32_t)tl, bs, lo, pmax, lane);
                            fs.gather();
                            fs.params();
                            fs.totals();
#pragma unroll
                            for (int o = 5; o >= 0; --o) fs.order(o);
                            po_l = fs.bpo;
                            k_l = fs.kb;
                            rbits = fs.best;
                        } else {
                            rbits = rice_search_all(active ? (uint64_t)tl : 0, bs, lo, pmax, lane, &po_l, &k_l);
                        }
                        if (MX <= 2147483647.0) {
                            const uint64_t est = 8 + (uint64_t)wasted + 4 + 5 + (uint64_t)lo * (uint64_t)(prec + bps) + rbits;
                            if (est < best_bits) {
                                best_bits = est;
                                type = 3;
                                order = lo;
                                porder = po_l;
                                kbest = k_l;
                                shift = sh;
                                precision = prec;
#pragma unroll
                                for (int j = 0; j < MLO; ++j) qkeep[j] = qreg[j];
                            }
                        }
                    }
                    lds_fence();
                }
            }
        }
    }

    FA_STAMP(8);
    if (dry) {  // an analysis-only pass of the stereo decision: nothing is written
        if (pass == 0) est_left = best_bits;
        else {
            use_side = best_bits < est_left;
            if (use_side) fhe_use = a.hdr[a.nframes + f];
        }
        continue;
    }
    // ---- emit (with one possible VERBATIM retry) ---------------------------------------
    for (int attempt = 0; attempt < 2; ++attempt) {
        // make the LDS image hold what this subframe type needs
        if (type == 3) {
            // residual already in place
        } else {
            if (lds_is_residual) {
                uint32_t o2; bool e2;
                load_frame(src, bs, wasted, smp, lane, &o2, &e2, first, NCH, ch, side);
                lds_is_residual = false;
                lds_fence();
            }
            if (type == 2 && order > 0) {
                // fixed residual of order `order`, in place (history read before any write)
                if (active) {
                    const int cbase = kChunkStride * (lane + 1);
                    const int g0 = kChunk * lane;
                    int4 h = *reinterpret_cast<const int4*>(&smp[cbase - kChunkStride + 60]);
                    int64_t x1 = h.w, x2 = h.z, x3 = h.y, x4 = h.x;
                    for (int t = 0; t < 16; ++t) {
                        int4* px = reinterpret_cast<int4*>(&smp[cbase + 4 * t]);
                        int4 xv = *px;
                        int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                        int rs[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int gi = g0 + 4 * t + e;
                            const int64_t x0 = xs[e];
                            int64_t r;
                            if (order == 1) r = x0 - x1;
                            else if (order == 2) r = x0 - 2 * x1 + x2;
                            else if (order == 3) r = x0 - 3 * x1 + 3 * x2 - x3;
                            else r = x0 - 4 * x1 + 6 * x2 - 4 * x3 + x4;
                            rs[e] = (gi >= order && gi < bs) ? (int)r : xs[e];
                            x4 = x3; x3 = x2; x2 = x1; x1 = x0;
                        }
                        *px = make_int4(rs[0], rs[1], rs[2], rs[3]);
                    }
                }
                lds_is_residual = true;
            }
        }
        // Rice parameter table; ring as it was when this subframe began (all zero for channel 0)
        kpar[lane] = (uint8_t)kbest;
        for (int i = lane; i < kRingWords; i += 64) ring[i] = 0;
        if (lane == 0) ring[kRingWords] = save_m;
        lds_fence();
        if (NCH == 2 && ch > 0) {
            ring[(bf0 * 64 + lane) & kRingMask] = save_w;
            lds_fence();
        }
        pos = pos0;
        blocks_flushed = bf0;

        bool rice2 = false;
        if (type >= 2) rice2 = __any((lane < (1 << porder)) && (kbest >= 15));
        const int plen = rice2 ? 5 : 4;

        FA_STAMP(9);
        // ---- preamble: frame header, subframe header, warm-up, LPC fields, residual header ----
        // One field (<= 32 bits) per lane, lane order = bit order; a scan of the field widths
        // gives every lane its bit position and the fields are ORed into the zeroed ring.
        auto put_bits = [&](uint32_t P, uint64_t val, uint32_t nb) __attribute__((always_inline)) {
            // val has nb (1..33: a side channel's sample fields) significant bits; place it at absolute bit position P
            const uint32_t off = P & 31u;
            const uint64_t X = val << (64u - nb - off);
            const uint32_t a0 = (P >> 3) & 0x7FCu;  // byte offset of the word inside the 2 KB ring
            atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ring) + a0), (uint32_t)(X >> 32));
            atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ring) + a0 + 4), (uint32_t)X);  // may be the mirror word
        };
        {
            const int tc = (type == 0) ? 0x00 : (type == 1) ? 0x01 : (type == 2) ? (0x08 | order) : (0x20 | (order - 1));
            const uint64_t smask = (1ull << bps) - 1ull;  // (bps <= 33; values are sign-extended into the field)
            const int nwarm = (type == 0) ? 1 : (type >= 2) ? order : 0;
            constexpr int kWarmLanes = (MLO > 4) ? MLO : 4;
            constexpr int kL_warm = 6, kL_lpc = kL_warm + kWarmLanes, kL_coef = kL_lpc + 1, kL_rice = kL_coef + ((MLO > 0) ? MLO : 1);
            static_assert(kL_rice < 64, "preamble fields must fit the wave");
            uint64_t fv = 0;
            uint32_t fnb = 0;
            const bool fh = (ch == 0);  // the frame header precedes the first subframe only
            if (lane == 0) { if (fh) { fv = fhe.x; fnb = 32; } }
            else if (lane == 1) { if (fh) { fv = fhe.y; fnb = (fhe.w >> 8) & 0xFFu; } }
            else if (lane == 2) { if (fh) { fv = fhe.z & 0xFFFFu; fnb = (fhe.w >> 16) & 0xFFu; } }
            else if (lane == 3) { if (fh) { fv = fhe.z >> 16; fnb = fhe.w >> 24; } }
            else if (lane == 4) {
                fv = (uint32_t)((tc << 1) | (wasted ? 1 : 0));
                fnb = 8;
                if (fh) { fv |= (fhe.w & 0xFFu) << 8; fnb = 16; }
            }
            else if (lane == 5) { if (wasted) { fv = 1; fnb = (uint32_t)wasted; } }  // unary: wasted-1 zeros, then 1
            else if (lane < kL_lpc) { if (lane - kL_warm < nwarm) { fv = (uint64_t)(int64_t)smp[smp_idx(lane - kL_warm)] & smask; fnb = (uint32_t)bps; } }
            else if (lane == kL_lpc) { if (type == 3) { fv = ((uint32_t)(precision - 1) << 5) | (uint32_t)shift; fnb = 9; } }
            else if (lane < kL_rice) {
                if constexpr (MLO > 0) {
                    if (type == 3 && lane - kL_coef < order) {
                        int32_t q = 0;
#pragma unroll
                        for (int j = 0; j < MLO; ++j) q = (lane - kL_coef == j) ? qkeep[j] : q;
                        fv = (uint32_t)q & ((1u << precision) - 1u);
                        fnb = (uint32_t)precision;
                    }
                }
            }
            else if (lane == kL_rice) { if (type >= 2) { fv = ((rice2 ? 1u : 0u) << 4) | (uint32_t)porder; fnb = 6; } }
            const uint32_t incl = wave_incl_scan_u32(fnb);
            if (fnb) put_bits(pos0 + incl - fnb, fv, fnb);
            pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        lds_fence();

        FA_STAMP(10);
        // ---- rows: 4 consecutive samples per lane, scan of code lengths, OR into the ring ----
        // Branch-free per item: an item that does not exist (warm-up sample, past the end of the
        // frame) has length 0 and ORs zeros.
        bool overflow = false;
        auto flush_blocks = [&]() __attribute__((always_inline)) {
            // no fence: DS operations of one wavefront are processed in issue order, so these
            // reads see every earlier ds_or of this wave
            const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pos >> 11));
            blocks_flushed = (uint32_t)__builtin_amdgcn_readfirstlane((int)blocks_flushed);
            while (blocks_flushed < done) {
                const uint32_t wi = (blocks_flushed * 64 + lane) & kRingMask;
                uint32_t wv = ring[wi];
                ring[wi] = 0;
                if ((blocks_flushed & 7u) == 0 && lane == 0) { wv |= ring[kRingWords]; ring[kRingWords] = 0; }
                reinterpret_cast<uint32_t*>(slot)[blocks_flushed * 64 + lane] = __builtin_bswap32(wv);
                blocks_flushed++;
            }
        };
        if (type == 1) {
            const uint64_t mask = (1ull << bps) - 1ull;
            for (int j = 0; j < nrows; ++j) {
                const int gb = kRow * j + 4 * lane;
                const int4 rv = *reinterpret_cast<const int4*>(&smp[smp_idx(gb)]);
                const int rs[4] = {rv.x, rv.y, rv.z, rv.w};
                int nvalid = bs - gb;
                nvalid = nvalid < 0 ? 0 : (nvalid > 4 ? 4 : nvalid);
                const uint32_t lane_len = (uint32_t)(nvalid * bps);
                const uint32_t incl = wave_incl_scan_u32(lane_len);
                const uint32_t row_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t p = pos + incl - lane_len;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint64_t v = (e < nvalid) ? ((uint64_t)(int64_t)rs[e] & mask) : 0ull;
                    put_bits(p, v, (uint32_t)bps);
                    p += (e < nvalid) ? (uint32_t)bps : 0u;
                }
                pos += row_total;
                flush_blocks();
            }
        } else if (type >= 2) {
            const uint32_t ps = (uint32_t)(bs >> porder);
            // floor(gi / ps) = umulhi(gi, magic); ps is a power of two for full frames (no 64-bit division then)
            const uint32_t magic = ((ps & (ps - 1u)) == 0u) ? (uint32_t)(0x100000000ULL >> (31 - __clz((int)ps)))
                                                           : (uint32_t)((0x100000000ULL + ps - 1) / ps);
            // GUARD: the row may hold warm-up samples or reach past the end of the frame
            auto rice_row = [&](auto guard_tag, int j) __attribute__((always_inline)) -> bool {
                constexpr bool GUARD = decltype(guard_tag)::value;
                const int gb = kRow * j + 4 * lane;
                const int4 rv = *reinterpret_cast<const int4*>(&smp[smp_idx(gb)]);
                const int rs[4] = {rv.x, rv.y, rv.z, rv.w};
                const uint32_t pidx = __umulhi((uint32_t)gb, magic);
                const uint32_t k = kpar[pidx];
                const uint32_t pstart = (pidx == 0) ? (uint32_t)order : pidx * ps;
                uint32_t iq[4], ival[4], ilen[4];
                uint32_t lane_len = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gi = gb + e;
                    const uint32_t u = ((uint32_t)rs[e] << 1) ^ (uint32_t)(rs[e] >> 31);
                    const uint32_t q = u >> k;
                    const uint32_t val = (1u << k) | (u & ((1u << k) - 1u));
                    if constexpr (GUARD) {
                        const bool valid = (gi >= order) && (gi < bs);
                        const uint32_t pre = ((uint32_t)gi == pstart) ? (uint32_t)plen : 0u;
                        iq[e] = valid ? (q + pre) : 0u;  // zeros before the stop bit (+ room for the parameter)
                        ival[e] = valid ? val : 0u;
                        ilen[e] = valid ? (q + pre + k + 1u) : 0u;
                    } else {
                        // only the first sample of a lane's group can open a partition (64 | partition size)
                        const uint32_t pre = (e == 0 && (uint32_t)gb == pstart) ? (uint32_t)plen : 0u;
                        iq[e] = q + pre;
                        ival[e] = val;
                        ilen[e] = q + pre + k + 1u;
                    }
                    lane_len += ilen[e];
                }
                const uint32_t incl = wave_incl_scan_u32(lane_len);
                const uint32_t row_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (row_total > (uint32_t)kRowCapBits) return false;
                uint32_t p = pos + incl - lane_len;
                // the (at most one) partition parameter this lane owns in this row
                if constexpr (GUARD) {
                    const uint32_t de = pstart - (uint32_t)gb;  // 0..3 if the partition starts in this lane's group
                    if (de < 4u && (int)pstart >= order && (int)pstart < bs) {
                        uint32_t pp = p;
                        if (de > 0) pp += ilen[0];
                        if (de > 1) pp += ilen[1];
                        if (de > 2) pp += ilen[2];
                        put_bits(pp, k, (uint32_t)plen);
                    }
                } else {
                    if ((uint32_t)gb == pstart) put_bits(p, k, (uint32_t)plen);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    put_bits(p + iq[e], ival[e], k + 1u);
                    p += ilen[e];
                }
                pos += row_total;
                flush_blocks();
                return true;
            };
            if (!rice_row(std::true_type{}, 0)) overflow = true;
            if (full) {
                // Rows 1..15 of a full frame (partition size a power of two >= 64, no warm-up samples, no
                // tail): the codes and the length scan of row j+1 are prepared before row j is written,
                // so that the scan's dependent DPP chain and lane read overlap the writer's address
                // arithmetic and LDS atomics instead of stalling the wave.
                struct RowPrep {
                    uint32_t u[4], q[4], k, lane_len, incl, total;
                    bool newp;
                };
                const int l2ps = 31 - __clz((int)ps);
                const int rowbase = smp_idx(4 * lane);  // smp_idx(kRow * j + 4 * lane) = rowbase + 4 * kChunkStride * j
                auto rice_prep = [&](int j, RowPrep& R) __attribute__((always_inline)) {
                    const uint32_t gb = (uint32_t)(kRow * j + 4 * lane);
                    const int4 rv = *reinterpret_cast<const int4*>(&smp[rowbase + 4 * kChunkStride * j]);
                    const int rs[4] = {rv.x, rv.y, rv.z, rv.w};
                    R.k = kpar[gb >> l2ps];
                    // gb > 0 here, so the short first partition (it starts at `order`) never opens in these rows
                    R.newp = (gb & (ps - 1u)) == 0u;
                    uint32_t len = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t u = ((uint32_t)rs[e] << 1) ^ (uint32_t)(rs[e] >> 31);
                        uint32_t q = u >> R.k;
                        if (e == 0) q += R.newp ? (uint32_t)plen : 0u;  // room for the partition's parameter
                        R.u[e] = u;
                        R.q[e] = q;
                        len += q;
                    }
                    R.lane_len = len + 4u * (R.k + 1u);
                    R.incl = wave_incl_scan_u32(R.lane_len);
                    R.total = (uint32_t)__builtin_amdgcn_readlane((int)R.incl, 63);
                };
                // one Rice code whose stop bit lands at bit P: stop bit + k low bits left-aligned in a word (whatever u
                // holds above bit k is shifted out or falls on the stop bit), funnel-shifted to P mod 32 -- no 64-bit
                // shift, no mask, no subtraction (the single-pass encoder's row writer, encode_fused.hpp)
                auto put_code = [&](uint32_t P, uint32_t u, uint32_t shl) __attribute__((always_inline)) {
                    const uint32_t vL = (u << shl) | 0x80000000u;
                    uint32_t wi;  // (asm: the compiler turns the bit-field extract into shift + mask and then needs an add)
                    asm("v_bfe_u32 %0, %1, 5, %2" : "=v"(wi) : "v"(P), "n"(__builtin_ctz((unsigned)kRingWords)));
                    uint32_t* const w = ring + wi;
                    atomicOr(w, __builtin_amdgcn_alignbit(0u, vL, P));
                    atomicOr(w + 1, __builtin_amdgcn_alignbit(vL, 0u, P));  // may be the mirror word
                };
                auto rice_put = [&](const RowPrep& R) __attribute__((always_inline)) {
                    const uint32_t k = R.k, kp1 = k + 1u;
                    const uint32_t p0 = pos + R.incl - R.lane_len;
                    uint32_t p = p0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        put_code(p + R.q[e], R.u[e], 31u - k);
                        p += R.q[e] + kp1;
                    }
                    // (last, so that the four unconditional codes share a basic block with the next row's scan)
                    if (R.newp) put_bits(p0, k, (uint32_t)plen);
                };
                if (!overflow) {
                    // two rows per trip, so that the two preparation records swap roles without register copies
                    constexpr int kLast = kMaxBlock / kRow - 1;
                    RowPrep ra, rb;
                    rice_prep(1, ra);
                    for (int j = 1; j <= kLast; j += 2) {
                        if (ra.total > (uint32_t)kRowCapBits) { overflow = true; break; }
                        rice_prep(j < kLast ? j + 1 : kLast, rb);  // (the last row is prepared twice, harmlessly)
                        rice_put(ra);
                        pos += ra.total;
                        flush_blocks();
                        if (j == kLast) break;
                        if (rb.total > (uint32_t)kRowCapBits) { overflow = true; break; }
                        rice_prep(j + 2 <= kLast ? j + 2 : kLast, ra);
                        rice_put(rb);
                        pos += rb.total;
                        flush_blocks();
                    }
                }
            } else {
                for (int j = 1; j < nrows && !overflow; ++j)
                    if (!rice_row(std::true_type{}, j)) overflow = true;
            }
        }
        FA_STAMP(11);
        if (!overflow && type >= 2) {
            const uint64_t exact = (uint64_t)pos - (pos0 + (ch == 0 ? fh_bits : 0u));
            if (exact > verbatim_bits) overflow = true;
        }
        if (overflow) {
            type = 1;
            order = 0;
            porder = 0;
            continue;  // second attempt writes the VERBATIM subframe
        }
        break;
    }
    if (lane == 0 && a.info) {
        FrameInfo fi;
        fi.type = type;
        fi.order = (type >= 2) ? order : 0;
        fi.porder = (type >= 2) ? porder : 0;
        fi.wasted = wasted;
        fi.shift = (type == 3) ? shift : 0;
        fi.precision = (type == 3) ? precision : 0;
        fi.nbytes = 0;  // filled in below, once the frame is complete
        fi.blocksize = bs;
        a.info[g * NCH + ch] = fi;
    }
    }  // pass loop (one per channel; two-channel frames: up to two analysis-only passes first)

    // ---- tail: byte align, 16 zero bits for the CRC-16 (filled in by K5), final flush ----
    pos = (pos + 7u) & ~7u;
    pos += 16;
    total_bytes = pos >> 3;
    {
        const uint32_t nwords = (total_bytes + 3) >> 2;
        for (uint32_t w0 = blocks_flushed * 64; w0 < nwords; w0 += 64) {
            const uint32_t wl = w0 + lane;
            if (wl < nwords) {
                uint32_t wv = ring[wl & kRingMask];
                if ((wl & kRingMask) == 0) wv |= ring[kRingWords];
                reinterpret_cast<uint32_t*>(slot)[wl] = __builtin_bswap32(wv);
            }
        }
    }
    FA_STAMP(12);
    FA_STAMP_FLUSH;
    if (lane == 0) {
        a.frame_bytes[g] = total_bytes;
        if (a.info)
            for (int c = 0; c < NCH; ++c) a.info[g * NCH + c].nbytes = (int32_t)total_bytes;
    }
    return total_bytes;
}

template <int MLO, int NCH = 1>
FA_GLOBAL __global__ __launch_bounds__(64) FA_K3_WAVES_ATTR void encode_frames_kernel(EncodeArgs a) {
#ifdef FA_LDS_PAD
    __shared__ __attribute__((aligned(16))) int32_t lds[k3_lds_words<NCH>() + FA_LDS_PAD];  // occupancy experiment
#else
    __shared__ __attribute__((aligned(16))) int32_t lds[k3_lds_words<NCH>()];
#endif
    const int64_t g = a.tail_only ? ((int64_t)blockIdx.x * a.nframes + a.nframes - 1) : (int64_t)blockIdx.x;
    uint8_t* slot = a.slots + (size_t)(a.tail_only ? (int64_t)blockIdx.x : g) * (size_t)a.slot_stride;
#ifdef FA_STAMPS  // diagnostic build: how many workgroups are alive at once (stamps[41] = the most seen)
    if (threadIdx.x == 0 && a.stamps) atomicMax(&a.stamps[41], atomicAdd(&a.stamps[40], 1ULL) + 1ULL);
#endif
    (void)encode_frame_body<MLO, NCH>(a, g, slot, lds, (int)threadIdx.x);
#ifdef FA_STAMPS
    if (threadIdx.x == 0 && a.stamps) atomicAdd(&a.stamps[40], ~0ULL);
#endif
}

// ------------------------------------------------------------------------------------------
// K4: sizes -> offsets
// ------------------------------------------------------------------------------------------
// one 256-thread block per stream: exclusive scan of its frame sizes
FA_GLOBAL __global__ __launch_bounds__(256) void stream_scan_kernel(const uint32_t* __restrict__ frame_bytes, int64_t nframes,
                                                          int64_t* __restrict__ frame_off, int64_t* __restrict__ stream_nbytes) {
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    const int64_t s = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nframes; base += 256) {
        const int64_t f = base + tid;
        const uint64_t v = (f < nframes) ? frame_bytes[s * nframes + f] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint64_t t = (tid >= off) ? sh[tid - off] : 0;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        const uint64_t c = carry;
        if (f < nframes) frame_off[s * nframes + f] = (int64_t)(c + sh[tid] - v);
        __syncthreads();
        if (tid == 255) carry = c + sh[255];
        __syncthreads();
    }
    if (tid == 0) stream_nbytes[s] = (int64_t)carry + stream_header_bytes(nframes);
}

// single 1024-thread block: exclusive scan of stream sizes -> starts, total
FA_GLOBAL __global__ __launch_bounds__(1024) void starts_scan_kernel(const int64_t* __restrict__ stream_nbytes, int64_t n_stream,
                                                           int64_t* __restrict__ starts, int64_t* __restrict__ total) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_stream; base += 1024) {
        const int64_t i = base + tid;
        const uint64_t v = (i < n_stream) ? (uint64_t)stream_nbytes[i] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint64_t t = (tid >= off) ? sh[tid - off] : 0;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        const uint64_t c = carry;
        if (i < n_stream) starts[i] = (int64_t)(c + sh[tid] - v);
        __syncthreads();
        if (tid == 1023) carry = c + sh[1023];
        __syncthreads();
    }
    if (tid == 0) *total = (int64_t)carry;
}

#endif  // !FA_UNIT_COMPACT

#if defined(FA_UNIT_COMPACT) || !defined(FA_SPLIT_UNITS)
#define FA_HAVE_K5 1
#endif

#ifdef FA_HAVE_K5
// ------------------------------------------------------------------------------------------
// K5a: stream headers.  One 256-thread block per stream.
// ------------------------------------------------------------------------------------------
FA_GLOBAL __global__ __launch_bounds__(256) void write_headers_kernel(uint8_t* __restrict__ out, const int64_t* __restrict__ starts,
                                                            const int64_t* __restrict__ frame_off, int64_t nframes,
                                                            int64_t stream_size, int32_t B, int32_t tail_bs, int32_t nch) {
    const int64_t s = blockIdx.x;
    uint8_t* h = out + starts[s];
    const int tid = threadIdx.x;
    if (tid == 0) {
        h[0] = 'f'; h[1] = 'L'; h[2] = 'a'; h[3] = 'C';
        h[4] = 0x00; h[5] = 0; h[6] = 0; h[7] = 34;
        uint8_t* si = h + 8;
        si[0] = (uint8_t)(B >> 8); si[1] = (uint8_t)B; si[2] = (uint8_t)(B >> 8); si[3] = (uint8_t)B;
        for (int i = 4; i < 10; ++i) si[i] = 0;
        const uint64_t ts = ((uint64_t)stream_size < (1ULL << 36)) ? (uint64_t)stream_size : 0;
        const uint64_t packed = ((uint64_t)44100 << 44) | ((uint64_t)(nch - 1) << 41) | ((uint64_t)31 << 36) | ts;
        for (int i = 0; i < 8; ++i) si[10 + i] = (uint8_t)(packed >> (56 - 8 * i));
        for (int i = 18; i < 34; ++i) si[i] = 0;
        uint8_t* t = h + 42;
        const uint32_t stl = (uint32_t)(18 * nframes);
        t[0] = 0x83; t[1] = (uint8_t)(stl >> 16); t[2] = (uint8_t)(stl >> 8); t[3] = (uint8_t)stl;
    }
    for (int64_t f = tid; f < nframes; f += 256) {
        uint8_t* p = h + 46 + 18 * f;
        const uint64_t sn = (uint64_t)f * (uint64_t)B;
        const uint64_t off = (uint64_t)frame_off[s * nframes + f];
        const int bs = (f == nframes - 1) ? tail_bs : B;
        for (int i = 0; i < 8; ++i) { p[i] = (uint8_t)(sn >> (56 - 8 * i)); p[8 + i] = (uint8_t)(off >> (56 - 8 * i)); }
        p[16] = (uint8_t)(bs >> 8);
        p[17] = (uint8_t)bs;
    }
}

#endif  // FA_HAVE_K5

// ------------------------------------------------------------------------------------------
// K5b: move every frame from its slot to its final byte offset and fill in its CRC-16.
// One wavefront per frame, grid-stride; CRC tables (computed on the host once) live in LDS.
//   tab[0..1023]   slicing tables: crc contribution of byte k (k=0 most significant) of a word
//   tab[1024..1535] Z256 hi/lo: state advanced by 256 zero bytes
//   tab[1536..2047] xpow[n] = x^(8n) mod P, n < 512
// ------------------------------------------------------------------------------------------
constexpr int kCrcTabWords = 2048;  // uint16 entries
// 256-byte blocks a wave keeps in flight per trip: 32 waves per CU x 2 KB = 64 KB outstanding per CU, what the HBM
// latency-bandwidth product asks for (tools/ubench/copy_bw.hip: 32 KB per CU in flight copies at 4.7-4.9 TB/s,
// 64-128 KB at 5.6-5.9 TB/s)
#ifndef FA_K5_GROUP
#define FA_K5_GROUP 8
#endif
constexpr int kK5Group = FA_K5_GROUP;

__device__ __forceinline__ uint16_t crc_mulmod(uint16_t a, uint16_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        r = (r << 1) ^ ((r & 0x8000u) ? 0x18005u : 0u);
        if ((b >> i) & 1) r ^= a;
    }
    return (uint16_t)r;
}

// One frame, one wavefront: n bytes at srcw (a slot: word aligned, CRC field zero, readable in whole groups of GROUP
// 256-byte blocks past the frame's end) go to dst (any byte alignment) with the CRC-16 filled in.  tab: the tables
// above, in LDS.  Callers: K5 below, and the placing encoder (encode_placed.hpp), whose waves move their own frames.
template <int GROUP = kK5Group>
__device__ __forceinline__ void compact_one_frame(const int lane, const uint32_t* __restrict__ srcw, const uint32_t n, uint8_t* __restrict__ dst,
                                                  const uint16_t* tab) {
    const uint8_t* srcb = reinterpret_cast<const uint8_t*>(srcw);
    const uint32_t L = n - 2;  // bytes covered by the CRC
    // destination-aligned words: dst word w holds source bytes [hcopy + 4w, hcopy + 4w + 4)
    const uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
    const uint32_t hcopy = head < n ? head : n;
    const uint32_t nw = (n - hcopy) >> 2;
    uint32_t* dstw = reinterpret_cast<uint32_t*>(dst + hcopy);
    const uint32_t sh = hcopy & 3;
    // ---- one pass: CRC-16 over source bytes [0, L) and the copy of every full destination word
    //      that does not contain a CRC byte; GROUP 256-byte blocks in flight per iteration ----
    const uint32_t NB = (L + 255) >> 8;
    uint16_t t = 0;
    uint32_t last_full = 0;  // number of blocks in which this lane held a full word
    uint16_t partial = 0;
    for (uint32_t b0 = 0; b0 < NB; b0 += GROUP) {
        uint32_t w[GROUP], w1[GROUP];
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            const uint32_t wi = 64u * (b0 + i) + (uint32_t)lane;  // source word index (slots and workspace have slack)
            w[i] = srcw[wi];
            w1[i] = srcw[wi + 1];
        }
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            const uint32_t b = b0 + i;
            const uint32_t o = 256u * b + 4u * (uint32_t)lane;
            if (b < NB) {
                if (o + 4 <= L) {
#ifdef FA_K5_NOCRC  // timing experiment only (wrong CRC-16): what the six table lookups per word cost
                    t = (uint16_t)(t ^ w[i] ^ (w[i] >> 16));
#else
                    const uint16_t adv = (uint16_t)(tab[1024 + (t >> 8)] ^ tab[1280 + (t & 255)]);
                    const uint16_t c = (uint16_t)(tab[w[i] & 255] ^ tab[256 + ((w[i] >> 8) & 255)] ^ tab[512 + ((w[i] >> 16) & 255)] ^ tab[768 + (w[i] >> 24)]);
                    t = (uint16_t)(adv ^ c);
#endif
                    last_full = b + 1;
                }
            }
            // destination word that starts at source byte o + sh (needs o >= hcopy - sh, i.e. not before the head)
            const uint32_t so = o + sh;  // source byte offset of this destination word
            if (so >= hcopy && so + 4 <= L) {
                const uint32_t v = sh ? __builtin_amdgcn_alignbyte(w1[i], w[i], sh) : w[i];
                dstw[(so - hcopy) >> 2] = v;
            }
        }
    }
    // the one word that holds the last 1..3 bytes under the CRC (if L is not a multiple of 4): its lane reads it again
    // here, so that the unrolled loop above carries no byte loop (code size: the placing encoder shares its instruction
    // cache with a 58 KB frame body)
    if ((L & 3u) != 0 && (uint32_t)lane == ((L >> 2) & 63u)) {
        const uint32_t wl = srcw[L >> 2];
        uint16_t c = 0;
        for (uint32_t k = 0; k < (L & 3u); ++k) c = crc16_byte(c, (uint8_t)(wl >> (8 * k)));
        partial = c;
    }
    uint16_t contrib = partial;
    if (last_full > 0) {
        const uint32_t after = L - (256u * (last_full - 1) + 4u * (uint32_t)lane + 4u);  // bytes after the lane's last word
        contrib ^= crc_mulmod(t, tab[1536 + after]);
    }
    uint32_t cr = contrib;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) cr ^= (uint32_t)__shfl_xor((int)cr, off, 64);
    const uint16_t crc = (uint16_t)cr;
    // ---- edges: head bytes, the words around the CRC, tail bytes ----
    if ((uint32_t)lane < hcopy) {
        uint8_t v = srcb[lane];
        if ((uint32_t)lane == n - 2) v = (uint8_t)(crc >> 8);
        if ((uint32_t)lane == n - 1) v = (uint8_t)crc;
        dst[lane] = v;
    }
    // bytes from the first destination word that was not written above to the end of the frame
    uint32_t done = hcopy;  // first source byte not yet copied
    if (L >= hcopy + 4) done = hcopy + (((L - hcopy) >> 2) << 2);
    // the loop above wrote dst words with so + 4 <= L  <=>  (so - hcopy) / 4 < (L - hcopy) / 4 (so = hcopy + 4k)
    const uint32_t rest = n - done;  // < 8 + 2
    if ((uint32_t)lane < rest) {
        const uint32_t bo = done + lane;
        uint8_t v = srcb[bo];
        if (bo == n - 2) v = (uint8_t)(crc >> 8);
        if (bo == n - 1) v = (uint8_t)crc;
        dst[bo] = v;
    }
    (void)nw;
}

#ifdef FA_HAVE_K5
// (eight waves per SIMD: this kernel lives on memory-level parallelism, whatever the scheduling strategy of the build)
FA_GLOBAL __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void compact_frames_kernel(const uint8_t* __restrict__ slots,
                                                             const uint32_t* __restrict__ frame_bytes,
                                                             const int64_t* __restrict__ frame_off,
                                                             const int64_t* __restrict__ starts, int64_t nframes,
                                                             int64_t total_frames, const uint16_t* __restrict__ crc_tab,
                                                             uint8_t* __restrict__ out, int64_t slot_stride) {
    __shared__ uint16_t tab[kCrcTabWords];
    for (int i = threadIdx.x; i < kCrcTabWords; i += 256) tab[i] = crc_tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t hb = stream_header_bytes(nframes);
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < total_frames; g += (int64_t)gridDim.x * 4) {
        const int64_t s = g / nframes;
        compact_one_frame(lane, reinterpret_cast<const uint32_t*>(slots + (size_t)g * (size_t)slot_stride), frame_bytes[g],
                          out + starts[s] + hb + frame_off[g], tab);
    }
}

// host-side launchers of the two K5 kernels (defined in the unit that holds the kernels)
void launch_write_headers(hipStream_t st, int64_t n_stream, uint8_t* out, const int64_t* starts, const int64_t* frame_off,
                          int64_t nframes, int64_t stream_size, int32_t B, int32_t tail_bs, int32_t nch)
#if defined(FA_HAVE_K5_LAUNCHERS)
{
    hipLaunchKernelGGL(write_headers_kernel, dim3((unsigned)n_stream), dim3(256), 0, st, out, starts, frame_off, nframes,
                       stream_size, B, tail_bs, nch);
}
#else
;
#endif
void launch_compact_frames(hipStream_t st, int64_t nblk, const uint8_t* slots, const uint32_t* frame_bytes,
                           const int64_t* frame_off, const int64_t* starts, int64_t nframes, int64_t total_frames,
                           const uint16_t* crc_tab, uint8_t* out, int64_t slot_stride)
#if defined(FA_HAVE_K5_LAUNCHERS)
{
    hipLaunchKernelGGL(compact_frames_kernel, dim3((unsigned)nblk), dim3(256), 0, st, slots, frame_bytes, frame_off, starts,
                       nframes, total_frames, crc_tab, out, slot_stride);
}
#else
;
#endif
#endif  // FA_HAVE_K5 (crc_mulmod, compact_frames_kernel, launchers)

#ifndef FA_HAVE_K5
void launch_write_headers(hipStream_t st, int64_t n_stream, uint8_t* out, const int64_t* starts, const int64_t* frame_off,
                          int64_t nframes, int64_t stream_size, int32_t B, int32_t tail_bs, int32_t nch);
void launch_compact_frames(hipStream_t st, int64_t nblk, const uint8_t* slots, const uint32_t* frame_bytes,
                           const int64_t* frame_off, const int64_t* starts, int64_t nframes, int64_t total_frames,
                           const uint16_t* crc_tab, uint8_t* out, int64_t slot_stride);
#endif

}  // namespace fa
