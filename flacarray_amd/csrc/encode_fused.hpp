// encode_fused.hpp -- K3F: single-pass FLAC frame encoder for gfx950 (wave64), full frames of 4096 mono samples.
//
// Replaces, for the frames it covers, the three-step sequence of encode_kernels.hpp (K3 analyse + pack into a
// per-frame slot, K4 scan of the sizes, K5 slot -> blob copy with CRC-16): one kernel analyses the frame, sizes it
// exactly, obtains its byte offset in the output blob from a look-back over the frames before it, and writes the
// bitstream -- CRC-16 included -- straight to its final position.  HBM traffic is the algorithmic 4 + c bytes per
// sample; there are no slots.  (Reference boundary: everything behind FLAC__stream_encoder_process_interleaved,
// src/flacarray/libflacarray/compress.c:374-378, plus the concatenation of compress.c:402-429.)
//
// One wavefront encodes one frame; four wavefronts form a workgroup that shares the CRC slicing tables in LDS and
// draws one ticket (4 consecutive frames) from a device-wide counter, so frame numbers are handed out in start
// order and a wave only ever waits for frames that started before it (no dependence on the dispatch order).
//
// Frame image, "split" layout -- lane l owns two 32-sample chunks:
//   A_l = samples [32 l, 32 l + 32)           in LDS (8.3 KB per wave incl. a zero history chunk), 16-byte units
//                                              XOR-swizzled so that lane-per-chunk and row-major ds_read_b128 are
//                                              both conflict free without padding;
//   B_l = samples [2048 + 32 l, 2048 + 32 l + 32)  in 32 VGPRs (history across lanes by DPP wave_shr:1).
// LDS per wave 10.4 KB (image + 2 KB bit ring) instead of 20.3 KB: occupancy is set by registers (3 waves per SIMD)
// rather than by LDS (2 per SIMD).  The second half is written to the LDS image after the rows of the first half
// have been emitted, so the row writer always reads row-major from LDS.
//
// Frame sizes -> byte offsets: the workgroup that draws ticket 0 does not encode; its first wave is the SCANNER.
// Every frame stores its size in size_pub[g] (bit 31 = valid).  The scanner follows the frontier of published sizes,
// 256 entries per step, and stores the absolute byte offset of every frame it passes into off_pub[g] (never zero).
// A frame polls only its own off_pub word: no two frames wait on the same address, and nobody polls the words the
// publishers write (a first version in which every waiting wave polled shared per-group counters ran 2-3x slower:
// thousands of pollers on the few cache lines the publishers needed, profiles/r02_single_pass/README.md).
#pragma once
#include "encode_kernels.hpp"
#include "quantize_kernels.hpp"

namespace fa {

constexpr int kFSmpWords = 65 * 32;  // zero history chunk + 64 chunks of the first half
#ifndef FA_F_SCANW
#define FA_F_SCANW 4  // size words per lane and scanner step (x 64 = entries per step)
#endif
#ifndef FA_F_RING
#define FA_F_RING 1024  // words of the bit ring (power of two >= 512)
#endif
constexpr int kFRingWords = FA_F_RING;
constexpr int kFRingMask = kFRingWords - 1;
constexpr int kFRingBlocks = kFRingWords / 64;
// completed 256-byte blocks are written out when the next row would not fit beside them (flush_blocks): the first
// flush is the point where the frame's byte offset must be known
constexpr int kFRowBlocks = 12;  // longest row of the writer (512 samples: two spec rows of 12288 bits each) in 256-byte blocks
static_assert(kFRingBlocks - kFRowBlocks - 1 >= 1, "the ring must hold the longest row and a partial block beside what waits to be flushed");
constexpr int kFWaveWords = kFSmpWords + kFRingWords + 4 + 16;  // image, ring, mirror word (+pad), Rice parameter table
constexpr int kFWaves = 4;                                      // wavefronts (frames) per workgroup
constexpr int kFCrcSlice = 1024;                                // 4 x 256 transformed slicing tables (uint16)
constexpr int kFCrcXpow = 520;                                  // x^(8 (i - 255)) mod P, i < 520
#ifndef FA_F_SLEEP
#define FA_F_SLEEP 8  // x 64 cycles between two polls of the look-back
#endif
#ifndef FA_F_WBATCH
#define FA_F_WBATCH 4  // groups of 4 samples whose window values are in flight together in the lag loops
#endif
#ifndef FA_F_CEIL
#define FA_F_CEIL 1  // the sample seeds the prediction chain, r = ceil(x - sum): one instruction per sample fewer.  Slower or flat in
                     // rounds 2-3 (the kernel was not issue bound then); -1.6 % since the wait for the offset is gone (r04o)
#endif
#ifndef FA_F_BFLY_N
#define FA_F_BFLY_N 1  // the nine lag sums step by step side by side (one LDS round trip for all swizzles) instead of one after the other
#endif
#ifndef FA_F_PARKN
#define FA_F_PARKN 32  // blocks that can wait in registers (an array of at most 32 registers is indexed in place, s_set_gpr_idx)
#endif
#ifndef FA_F_PARKBLK
#define FA_F_PARKBLK 1  // completed blocks wait in registers, not the wave, while the frame's byte offset is unknown
#endif
#ifndef FA_F_PARK
#define FA_F_PARK 1  // units 4..7 of the register image wait in the bit ring during the first half's lag products
#endif
#ifndef FA_F_WAVES
#define FA_F_WAVES 3  // waves per SIMD the register allocation aims at
#endif
constexpr uint32_t kLbSpinLimit = 1u << 22;  // polls before a wave gives up (error flag, frame dropped): every wave terminates

struct FusedArgs {
    const int32_t* data;  // [n_stream][stream_size]; float32 samples for the F32IN kernels
    const float* f_offsets;  // F32IN: per-stream offset and gain of the quantisation (utils.c:160-243)
    const float* f_gains;
    int64_t n_stream, stream_size, nframes, total_frames;
    int32_t max_lpc_order, max_porder, precision, pmax_full;
    int32_t tail_bs;        // samples of a stream's last frame; below 4096 that frame is not this kernel's (the slot encoder
                            // wrote it and published its size before the launch; the scanner places it like any other)
    double escale_full;     // 0.5 / 4096
    const float* win;       // [4096] tukey(0.5)
    const uint4* hdr;       // [nframes] frame header fields by frame number (frame_header_entry)
    uint8_t* blob;          // output, capacity bytes
    int64_t capacity;
    int64_t hb;             // stream header bytes (fLaC + STREAMINFO + SEEKTABLE)
    uint32_t* frame_bytes;  // [F]
    int64_t* frame_abs;     // [F] absolute byte offset of every frame in the blob
    FrameInfo* info;        // [F] or null
    uint32_t* ticket;       // one word, zero before the launch
    uint32_t* size_pub;     // [F] bit 31 | bytes of frame g, zero before the launch
    unsigned long long* off_pub;  // [F] absolute byte offset of frame g once every frame before it has published, zero before the launch
    int64_t* total;         // bytes of the whole blob (written by the scanner)
    int* err;               // error flags (OR)
    const uint16_t* crc_tab;  // [kFCrcSlice + kFCrcXpow]
    unsigned long long* stamps;  // diagnostic build (-DFA_STAMPS): per-phase cycle sums
    int64_t* starts;        // K3G only (encode_placed.hpp writes the stream index and headers itself): [n_stream]
    int64_t* nbytes;        // K3G only: [n_stream]
};

// word index of 16-byte unit u (0..7) of chunk c1 (0 = zero history, 1 + l = A_l)
__device__ __forceinline__ int fsmp_unit(int c1, int u) { return 32 * c1 + 4 * (u ^ (c1 & 7)); }
__device__ __forceinline__ int fsmp_idx(int s) {  // sample s of the first half (s >= -32)
    const int c1 = (s >> 5) + 1;
    return fsmp_unit(c1, (s >> 2) & 7) + (s & 3);
}

__device__ __forceinline__ int lane_id_opaque() {  // the lane number, recomputed wherever it is asked for (two instructions)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// zig-zag fold of a residual (RFC 9639 9.2.7.3).  The residual passes store the folded value: the exact-size pass and
// the row writer both start from it, the Rice parameter search needs only the magnitude sums taken before the fold.
__device__ __forceinline__ int rice_fold(int r) { return (int)(((uint32_t)r << 1) ^ (uint32_t)(r >> 31)); }
__device__ __forceinline__ int dpp_wave_shr1(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false); }

__device__ __forceinline__ uint16_t crc16_mulmod(uint16_t a, uint16_t b) {  // a * b mod x^16 + x^15 + x^2 + 1
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        r = (r << 1) ^ ((r & 0x8000u) ? 0x18005u : 0u);
        if ((b >> i) & 1) r ^= a;
    }
    return (uint16_t)r;
}

template <typename T>
__device__ __forceinline__ T lb_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Rice partition-order search for the split layout, 32-bit arithmetic (every lane's two sums below 2^24), orders
// 0..pmax (pmax <= 5), branch free and in stages like FastRiceSearch.  Slot lane L holds partition p of order po as in
// rice_search_batch; chunk sums are ordered A_0..A_63, B_0..B_63 (32 samples each).
struct SplitRiceSearch {
    int pred_order, pmax, lane;
    uint32_t TA, TB, S, n, pb, SC, best;
    int M, p, k, bpo, kb;
    bool slot;
    __device__ __forceinline__ void start(uint32_t ta, uint32_t tb, int pred_order_, int pmax_, int lane_) {
        pred_order = pred_order_; pmax = pmax_; lane = lane_;
        TA = wave_incl_scan_u32(ta);
        TB = wave_incl_scan_u32(tb);
    }
    __device__ __forceinline__ void gather() {
        M = (2 << pmax) - 1;
        const int Lp = M - lane;
        slot = (Lp >= 1);
        const int po = slot ? (31 - __clz(Lp)) : 0;
        p = lane - (M + 1 - (2 << po));
        const uint32_t psz = 4096u >> po;
        // chunks (of 32 samples) per partition; order 0 is the whole frame
        const int cpp = 128 >> po;
        const bool inB = (po > 0) && (p >= (1 << (po - 1)));
        const int ph = inB ? (p - (1 << (po - 1))) : p;
        int hi_l = (po == 0) ? 63 : ((ph + 1) * cpp - 1);
        int lo_l = (po == 0) ? -1 : (ph * cpp - 1);
        hi_l &= 63;
        const int lo_c = lo_l < 0 ? 0 : (lo_l & 63);
        const uint32_t ah = (uint32_t)__builtin_amdgcn_ds_bpermute(hi_l << 2, (int)TA);
        const uint32_t bh = (uint32_t)__builtin_amdgcn_ds_bpermute(hi_l << 2, (int)TB);
        const uint32_t al = (uint32_t)__builtin_amdgcn_ds_bpermute(lo_c << 2, (int)TA);
        const uint32_t bl = (uint32_t)__builtin_amdgcn_ds_bpermute(lo_c << 2, (int)TB);
        uint32_t s;
        if (po == 0) s = ah + bh;
        else s = (inB ? bh : ah) - ((lo_l >= 0) ? (inB ? bl : al) : 0u);
        S = slot ? s : 0u;
        n = psz - ((p == 0) ? (uint32_t)pred_order : 0u);
        n = slot ? n : 1u;
    }
    __device__ __forceinline__ void params() {
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

// The general search (any sums, pmax <= 6): one order at a time in 64-bit arithmetic.  Rare (huge residuals, or the
// 64-partition orders of levels 6-8).
__device__ __forceinline__ uint64_t split_rice_search_slow(uint64_t ta, uint64_t tb, int pred_order, int pmax, int lane, int* best_po,
                                                           int* kbest) {
    const uint64_t TA = wave_incl_scan_u64(ta), TB = wave_incl_scan_u64(tb);
    uint64_t best = 0;
    bool have = false;
    int bpo = 0, kb = 0;
    for (int po = pmax; po >= 0; --po) {
        const int nparts = 1 << po;
        const uint32_t psz = 4096u >> po;
        const int cpp = 128 >> po;
        const int p = lane & (nparts - 1);
        const bool inB = (po > 0) && (p >= (nparts >> 1));
        const int ph = inB ? (p - (nparts >> 1)) : p;
        const int hi_l = (po == 0) ? 63 : (((ph + 1) * cpp - 1) & 63);
        const int lo_l = (po == 0) ? -1 : (ph * cpp - 1);
        const int lo_c = lo_l < 0 ? 0 : (lo_l & 63);
        const uint64_t ah = gather_u64(TA, hi_l), bh = gather_u64(TB, hi_l), al = gather_u64(TA, lo_c), bl = gather_u64(TB, lo_c);
        uint64_t S;
        if (po == 0) S = ah + bh;
        else S = (inB ? bh : ah) - ((lo_l >= 0) ? (inB ? bl : al) : 0);
        uint64_t pb = 0;
        int k = 0;
        if (lane < nparts) {
            const uint32_t n = psz - ((lane == 0) ? (uint32_t)pred_order : 0u);
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
    }
    *best_po = bpo;
    *kbest = kb;
    return best;
}

// The scanner: one wave.  pos = first frame whose offset is not out yet, prefix = bytes of all frames before it.
// Every step loads the next 256 size words, finds how many of them (from pos on, without a gap) are published,
// turns those into offsets with a wave scan and stores them.  It ends when all F offsets are out; if nothing moves
// for kLbSpinLimit polls it raises the error flag and leaves (the waiting frames time out by themselves).
__device__ __forceinline__ void fused_scanner(const FusedArgs& a, int lane) {
    const uint32_t F = (uint32_t)a.total_frames;
    const uint32_t nf = (uint32_t)a.nframes;
    uint32_t pos = 0;
    uint64_t prefix = 0;
    uint32_t idle = 0;
    while (pos < F) {
        uint32_t v[FA_F_SCANW];
#pragma unroll
        for (int k = 0; k < FA_F_SCANW; ++k) {
            const uint32_t i = pos + 64u * k + (uint32_t)lane;
            v[k] = (i < F) ? lb_load(a.size_pub + i) : 0u;
        }
        uint32_t adv = 0;
        bool open = true;  // no gap so far
#pragma unroll
        for (int k = 0; k < FA_F_SCANW; ++k) {
            const uint64_t ball = __ballot((v[k] >> 31) != 0);
            const uint32_t cnt = open ? ((ball == ~0ULL) ? 64u : (uint32_t)__builtin_ctzll(~ball)) : 0u;
            if (cnt > 0) {
                const uint32_t mine = ((uint32_t)lane < cnt) ? (v[k] & 0x7fffffffu) : 0u;
                const uint32_t incl = wave_incl_scan_u32(mine);
                const uint32_t i = pos + 64u * k + (uint32_t)lane;
                if ((uint32_t)lane < cnt) {
                    const uint64_t off = prefix + (incl - mine) + (uint64_t)(i / nf + 1u) * (uint64_t)a.hb;
                    __hip_atomic_store(a.off_pub + i, (unsigned long long)off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef FA_TIMELINE
                    if (a.info) a.info[i].porder = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();  // offset out
#endif
                }
                prefix += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                adv += cnt;
            }
            open = open && (cnt == 64u);
        }
        pos += adv;
        if (adv == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (++idle > kLbSpinLimit) {
                if (lane == 0) atomicOr(a.err, 4);
                return;
            }
        } else {
            idle = 0;
        }
    }
    if (lane == 0) *a.total = (int64_t)(prefix + (uint64_t)a.n_stream * (uint64_t)a.hb);
}

#if defined(FA_UNIT_FUSED) || !defined(FA_SPLIT_UNITS)
// ------------------------------------------------------------------------------------------
// K3F
// ------------------------------------------------------------------------------------------
// F32IN: the input is float32 and is quantised with the stream's offset / gain wherever a row is loaded (K1 fused
// into the staging load: float32_to_int32, utils.c:232-240).
template <int MLO, bool F32IN = false>
// (order 12 keeps 24 more doubles live in the lag and residual loops: its register budget is the 256 of two waves per SIMD)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MLO > 8 ? 2 : FA_F_WAVES, MLO > 8 ? 2 : FA_F_WAVES))) void encode_fused_kernel(FusedArgs a) {
    static_assert(MLO > 0, "levels 0-2 (fixed predictors, 1152-sample blocks) use the slot path");
    __shared__ __attribute__((aligned(16))) int32_t lds_all[kFWaves * kFWaveWords];
    __shared__ __attribute__((aligned(16))) uint16_t crc_s[kFCrcSlice];
    __shared__ uint32_t ticket_s;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kFCrcSlice / 2; i += 256) reinterpret_cast<uint32_t*>(crc_s)[i] = reinterpret_cast<const uint32_t*>(a.crc_tab)[i];
    if (tid == 0) ticket_s = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // (the ticket comes out of LDS, i.e. in a vector register: say that it is wave-uniform, so that the frame number, the
    // stream, the source pointer and everything else derived from it live in scalar registers)
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket_s);
    if (ticket == 0) {  // the scanner workgroup (the first one to start)
        if (wave == 0) fused_scanner(a, lane);
        return;
    }
    const int64_t g = (int64_t)(ticket - 1) * kFWaves + wave;
    if (g >= a.total_frames) return;

    int32_t* smp = lds_all + wave * kFWaveWords;
    uint32_t* ring = reinterpret_cast<uint32_t*>(smp + kFSmpWords);
    uint8_t* kpar = reinterpret_cast<uint8_t*>(smp + kFSmpWords + kFRingWords + 4);
    uint32_t* scr = ring;  // analysis scratch before the ring is zeroed for the writer

    const int64_t s = (int64_t)((uint32_t)g / (uint32_t)a.nframes);
    const int64_t f = g - s * a.nframes;
    if (f == a.nframes - 1 && a.tail_bs != kMaxBlock) return;  // a short last frame: encoded by the slot kernel
    constexpr int bs = kMaxBlock;
    const int32_t* src = a.data + (s * a.stream_size + f * (int64_t)bs);
    float q_off = 0.0f, q_gain = 0.0f;
    if constexpr (F32IN) { q_off = a.f_offsets[s]; q_gain = a.f_gains[s]; }
    auto load_row = [&](int j) __attribute__((always_inline)) {  // the lane's 4 samples of row j
        const int4 v = reinterpret_cast<const int4*>(src)[64 * j + lane];
        if constexpr (F32IN)
            return make_int4(quantise_f32(__int_as_float(v.x), q_off, q_gain), quantise_f32(__int_as_float(v.y), q_off, q_gain),
                             quantise_f32(__int_as_float(v.z), q_off, q_gain), quantise_f32(__int_as_float(v.w), q_off, q_gain));
        else
            return v;
    };
    const float* win = a.win;
    const uint4 fhe = a.hdr[f];
    const uint32_t fh_bits = 32u + ((fhe.w >> 8) & 0xFFu) + ((fhe.w >> 16) & 0xFFu) + (fhe.w >> 24) + 8u;

    // per-lane image addresses (words): K ^ (4 t) is unit t of the lane's own chunk A_l; rowbase + 256 j is the lane's
    // 4 samples of row j (row-major); hist7 / hist6 are the last two units of the chunk before A_l
    // They are re-derived from the lane number at the head of every phase (FA_IMAGE_ADDRS; the lane number through an
    // opaque asm, so that the compiler cannot keep one copy alive -- and spill it -- across the whole kernel).
#define FA_IMAGE_ADDRS                                                                  \
    const int ln_ = lane_id_opaque();                                                   \
    const int K = fsmp_unit(ln_ + 1, 0);                                                \
    const int rowbase = fsmp_unit((ln_ >> 3) + 1, ln_ & 7);                             \
    const int hist7 = fsmp_unit(ln_, 7), hist6 = fsmp_unit(ln_, 6), hist5 = fsmp_unit(ln_, 5); \
    (void)K; (void)rowbase; (void)hist7; (void)hist6; (void)hist5
    constexpr int tailA7 = 32 * 64 + 4 * (7 ^ (64 & 7)), tailA6 = 32 * 64 + 4 * (6 ^ (64 & 7)), tailA5 = 32 * 64 + 4 * (5 ^ (64 & 7));

    FA_STAMP_INIT;
#ifdef FA_TIMELINE
    const uint32_t tl_start_ = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
    // ---- P0: stage the frame, wasted bits, constant / narrow tests -----------------------------------
    int4 Bv[8];  // B_l: samples 2048 + 32 l + 4 t + {0,1,2,3}
    uint32_t orv = 0;
    int mn, mx;
    {
        FA_IMAGE_ADDRS;
        int4 hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = load_row(8 + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = load_row(j);
        if (lane < 32) smp[lane] = 0;  // zero history chunk
        mn = hi[0].x;
        mx = hi[0].x;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            orv |= (uint32_t)(hi[j].x | hi[j].y | hi[j].z | hi[j].w);
            mn = min(min(mn, hi[j].x), hi[j].y);
            mn = min(min(mn, hi[j].z), hi[j].w);
            mx = max(max(mx, hi[j].x), hi[j].y);
            mx = max(max(mx, hi[j].z), hi[j].w);
            *reinterpret_cast<int4*>(&smp[rowbase + 256 * j]) = hi[j];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) Bv[t] = *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            orv |= (uint32_t)(lo[j].x | lo[j].y | lo[j].z | lo[j].w);
            mn = min(min(mn, lo[j].x), lo[j].y);
            mn = min(min(mn, lo[j].z), lo[j].w);
            mx = max(max(mx, lo[j].x), lo[j].y);
            mx = max(max(mx, lo[j].z), lo[j].w);
            *reinterpret_cast<int4*>(&smp[rowbase + 256 * j]) = lo[j];
        }
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32(mx);
    const bool is_const = (mn == mx);
    orv = wave_or_u32(orv);
    const int wasted = orv ? (__ffs((int)orv) - 1) : 0;
    const int bps = 32 - wasted;
    if (__builtin_expect(wasted != 0, 0)) {
        FA_IMAGE_ADDRS;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Bv[t].x >>= wasted; Bv[t].y >>= wasted; Bv[t].z >>= wasted; Bv[t].w >>= wasted;
            int4* p = reinterpret_cast<int4*>(&smp[K ^ (4 * t)]);
            int4 v = *p;
            v.x >>= wasted; v.y >>= wasted; v.z >>= wasted; v.w >>= wasted;
            *p = v;
        }
        mn >>= wasted;
        mx >>= wasted;
    }
    const bool narrow = (mn >= -(1 << 24)) && (mx < (1 << 24));  // every |x| <= 2^24: fixed-predictor errors fit 32-bit ints
    lds_fence();

    FA_STAMP(0);
    const uint64_t verbatim_bits = 8 + (uint64_t)wasted + (uint64_t)bs * (uint64_t)bps;
    int type = 1;  // 0 const, 1 verbatim, 2 fixed, 3 lpc
    int order = 0, porder = 0, shift = 0, precision = 0;
    int kbest = 0;
    bool img_is_residual = false;  // the image holds (folded) residuals, warm-up samples excepted
    bool win_small_l = false, win_small_f = false;  // the LPC / FIXED candidate's lane sums are all below 2^24
    int fo = -1;
    int32_t qkeep[MLO];
#pragma unroll
    for (int j = 0; j < MLO; ++j) qkeep[j] = 0;
    const int pmax_geo = a.pmax_full;
    auto pmax_for = [&](int pred_order) __attribute__((always_inline)) {
        int pm = pmax_geo;
        while (pm > 0 && (bs >> pm) <= pred_order) pm--;
        return pm;
    };

    // history of the B half: the last samples of B_{l-1}, lane 0 takes them from A_63 (uniform LDS read)
    auto hist_b = [&](const int4& own, const int4& tail) __attribute__((always_inline)) {
        int4 h;
        h.x = dpp_wave_shr1(tail.x, own.x);
        h.y = dpp_wave_shr1(tail.y, own.y);
        h.z = dpp_wave_shr1(tail.z, own.z);
        h.w = dpp_wave_shr1(tail.w, own.w);
        return h;
    };

    if (__builtin_expect(is_const, 0)) {
        type = 0;
    } else {
        uint64_t best_bits = verbatim_bits;
        // ---- P2: fixed predictors 0..4, lane partial sums over A_l then B_l -------------------------------
        double tot0 = 0.0, tot1 = 0.0, tot2 = 0.0, tot3 = 0.0, tot4 = 0.0;
        double mx0 = 0.0, mx1 = 0.0, mx2 = 0.0, mx3 = 0.0, mx4 = 0.0;
        // per-half sums of the winner are needed for the partition search: keep both halves of every order
        uint64_t hA0 = 0, hA1 = 0, hA2 = 0, hA3 = 0, hA4 = 0;
        double dA0 = 0.0, dA1 = 0.0, dA2 = 0.0, dA3 = 0.0, dA4 = 0.0;
        if (__builtin_expect(narrow, 1)) {
            FA_IMAGE_ADDRS;
            const uint32_t BIAS = 0x80000000u;
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
            int p1, pe1, pe2, pe3;
            auto seed = [&](const int4& hh) __attribute__((always_inline)) {
                p1 = hh.w;
                pe1 = hh.w - hh.z;
                pe2 = pe1 - (hh.z - hh.y);
                pe3 = pe2 - ((hh.z - hh.y) - (hh.y - hh.x));
            };
            auto group = [&](auto mask_tag, const int4& xv, int gi0) __attribute__((always_inline)) {
                constexpr bool MASK = decltype(mask_tag)::value;
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
                        const int gi = gi0 + e;
                        s0 = n0;
                        s1 = (gi >= 1) ? n1 : s1;
                        s2 = (gi >= 2) ? n2 : s2;
                        s3 = (gi >= 3) ? n3 : s3;
                        s4 = (gi >= 4) ? n4 : s4;
                    } else {
                        s0 = n0; s1 = n1; s2 = n2; s3 = n3; s4 = n4;
                    }
                    p1 = x; pe1 = e1; pe2 = e2; pe3 = e3;
                }
            };
            // |e_k| < 2^28: a 16-sample u32 partial sum cannot overflow; folded into 64 bits every four groups
            uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
            auto fold = [&]() __attribute__((always_inline)) {
                a0 += s0; a1 += s1; a2 += s2; a3 += s3; a4 += s4;
                s0 = s1 = s2 = s3 = s4 = 0;
            };
            seed(*reinterpret_cast<const int4*>(&smp[hist7]));
            group(std::true_type{}, *reinterpret_cast<const int4*>(&smp[K]), 32 * lane);  // the only group that can hold samples 0..3
#pragma unroll
            for (int t = 1; t < 4; ++t) group(std::false_type{}, *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]), 0);
            fold();
#pragma unroll
            for (int t = 4; t < 8; ++t) group(std::false_type{}, *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]), 0);
            fold();
            hA0 = a0; hA1 = a1; hA2 = a2; hA3 = a3; hA4 = a4;
            seed(hist_b(Bv[7], *reinterpret_cast<const int4*>(&smp[tailA7])));
#pragma unroll
            for (int t = 0; t < 4; ++t) group(std::false_type{}, Bv[t], 0);
            fold();
#pragma unroll
            for (int t = 4; t < 8; ++t) group(std::false_type{}, Bv[t], 0);
            fold();
            tot0 = (double)a0; tot1 = (double)a1; tot2 = (double)a2; tot3 = (double)a3; tot4 = (double)a4;
        } else {
            FA_IMAGE_ADDRS;
            double p1, pe1, pe2, pe3;
            auto seed = [&](const int4& h) __attribute__((always_inline)) {
                p1 = (double)h.w;
                pe1 = (double)h.w - (double)h.z;
                const double e1b = (double)h.z - (double)h.y;
                pe2 = pe1 - e1b;
                const double e2b = e1b - ((double)h.y - (double)h.x);
                pe3 = pe2 - e2b;
            };
            auto group = [&](const int4& xv, int gi0) __attribute__((always_inline)) {
                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gi = gi0 + e;
                    const double xd = (double)xs[e];
                    const double e1 = xd - p1;
                    const double e2 = e1 - pe1;
                    const double e3 = e2 - pe2;
                    const double e4 = e3 - pe3;
                    const double a0 = fa_fabs(xd);
                    const double a1 = (gi >= 1) ? fa_fabs(e1) : 0.0;
                    const double a2 = (gi >= 2) ? fa_fabs(e2) : 0.0;
                    const double a3 = (gi >= 3) ? fa_fabs(e3) : 0.0;
                    const double a4 = (gi >= 4) ? fa_fabs(e4) : 0.0;
                    tot0 += a0; tot1 += a1; tot2 += a2; tot3 += a3; tot4 += a4;
                    mx0 = __builtin_fmax(mx0, a0);
                    mx1 = __builtin_fmax(mx1, a1);
                    mx2 = __builtin_fmax(mx2, a2);
                    mx3 = __builtin_fmax(mx3, a3);
                    mx4 = __builtin_fmax(mx4, a4);
                    p1 = xd; pe1 = e1; pe2 = e2; pe3 = e3;
                }
            };
            seed(*reinterpret_cast<const int4*>(&smp[hist7]));
#pragma unroll 1
            for (int t = 0; t < 8; ++t) group(*reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]), 32 * lane + 4 * t);
            dA0 = tot0; dA1 = tot1; dA2 = tot2; dA3 = tot3; dA4 = tot4;
            seed(hist_b(Bv[7], *reinterpret_cast<const int4*>(&smp[tailA7])));
#pragma unroll
            for (int t = 0; t < 8; ++t) group(Bv[t], 4096);
        }
        FA_STAMP(1);
        {
            double tsum[5] = {tot0, tot1, tot2, tot3, tot4};
            wave_sum_butterfly_n<5>(tsum);
            const double T0 = tsum[0], T1 = tsum[1], T2 = tsum[2], T3 = tsum[3], T4 = tsum[4];
            double M0 = 0.0, M1 = 0.0, M2 = 0.0, M3 = 0.0, M4 = 0.0;
            if (!narrow) {
                M0 = wave_max_f64(mx0); M1 = wave_max_f64(mx1); M2 = wave_max_f64(mx2); M3 = wave_max_f64(mx3);
                M4 = wave_max_f64(mx4);
            }
            const double lim = 2147483647.0;
            double smallest = 1.8446744073709552e19;
            if (M0 <= lim && T0 < smallest) { fo = 0; smallest = T0; }
            if (M1 <= lim && T1 < smallest) { fo = 1; smallest = T1; }
            if (M2 <= lim && T2 < smallest) { fo = 2; smallest = T2; }
            if (M3 <= lim && T3 < smallest) { fo = 3; smallest = T3; }
            if (M4 <= lim && T4 < smallest) { fo = 4; smallest = T4; }
        }
        // the winner's per-half magnitude sums (exact integers either way)
        uint64_t fixA, fixB;
        if (narrow) {
            const uint64_t tA = (fo == 0) ? hA0 : (fo == 1) ? hA1 : (fo == 2) ? hA2 : (fo == 3) ? hA3 : hA4;
            const double tt = (fo == 0) ? tot0 : (fo == 1) ? tot1 : (fo == 2) ? tot2 : (fo == 3) ? tot3 : tot4;
            fixA = tA;
            fixB = (uint64_t)tt - tA;
        } else {
            const double tA = (fo == 0) ? dA0 : (fo == 1) ? dA1 : (fo == 2) ? dA2 : (fo == 3) ? dA3 : dA4;
            const double tt = (fo == 0) ? tot0 : (fo == 1) ? tot1 : (fo == 2) ? tot2 : (fo == 3) ? tot3 : tot4;
            fixA = (uint64_t)tA;
            fixB = (uint64_t)(tt - tA);
        }
        int po_fix = 0, k_fix = 0;
        const int pmax_fix = pmax_for(fo < 0 ? 0 : fo);
        const bool small_fix = __all((fixA < (1u << 24)) && (fixB < (1u << 24)));
        win_small_f = small_fix;
        const bool fuse_search = fo >= 0 && pmax_fix <= 5 && a.max_lpc_order > 0 && small_fix;
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
        if (__builtin_expect(fo >= 0 && !fuse_search, 0)) {
            uint64_t rb;
            if (pmax_fix <= 5 && small_fix) {
                SplitRiceSearch fs;
                fs.start((uint32_t)fixA, (uint32_t)fixB, fo, pmax_fix, lane);
                fs.gather(); fs.params(); fs.totals();
#pragma unroll
                for (int o = 5; o >= 0; --o) fs.order(o);
                po_fix = fs.bpo; k_fix = fs.kb; rb = fs.best;
            } else {
                rb = split_rice_search_slow(fixA, fixB, fo, pmax_fix, lane, &po_fix, &k_fix);
            }
            est_fix = 8 + (uint64_t)wasted + (uint64_t)fo * (uint64_t)bps + rb;
            apply_fixed();
        }

        FA_STAMP(2);
        // ---- P3: LPC analysis ---------------------------------------------------------------------------
        int mlo = a.max_lpc_order;
        if (__builtin_expect(mlo > 0, 1)) {
            FA_IMAGE_ADDRS;
            double acc[MLO + 1];
#pragma unroll
            for (int j = 0; j <= MLO; ++j) acc[j] = 0.0;
            SplitRiceSearch fs;
            if (fuse_search) fs.start((uint32_t)fixA, (uint32_t)fixB, fo, pmax_fix, lane);
            double hist[MLO];
            // (a table of doubles would save the conversion of the window value, one instruction per sample of twelve, but
            // the doubled registers of the values in flight cost more in spills than that returns: scratch 32 -> 170 B)
            struct WinV { double w[4]; };
            auto load_win = [&](int i0) __attribute__((always_inline)) {
                const float4 f = *reinterpret_cast<const float4*>(win + i0);
                WinV v;
                v.w[0] = (double)f.x; v.w[1] = (double)f.y; v.w[2] = (double)f.z; v.w[3] = (double)f.w;
                return v;
            };
            auto lag_group = [&](const int4& xv, const WinV& wv) __attribute__((always_inline)) {
                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = (double)xs[e] * wv.w[e];
                    acc[0] = __builtin_fma(d, d, acc[0]);
#pragma unroll
                    for (int j = 0; j < MLO; ++j) acc[j + 1] = __builtin_fma(d, hist[j], acc[j + 1]);
#pragma unroll
                    for (int j = MLO - 1; j > 0; --j) hist[j] = hist[j - 1];
                    hist[0] = d;
                }
            };
            auto search_stage = [&](int tt) __attribute__((always_inline)) {  // tt: group 0..15 of the lane (constant after unrolling)
                if (!fuse_search) return;
                if (tt == 1) fs.gather();
                if (tt == 3) fs.params();
                if (tt == 5) fs.totals();
                if (tt >= 7 && tt <= 12) fs.order(12 - tt);
            };
#if FA_F_PARK
            // Half of the register image (units 4..7 of B_l: 16 registers) waits in the idle bit ring while the lag products of
            // the first half run -- the one place where the kernel's register demand peaks (nine accumulators, eight history
            // values and the window values in doubles beside the 32 registers of B_l).  Left to itself the compiler
            // spills exactly these 16 registers to scratch there, i.e. to memory; the ring is LDS and idle until the writer.
            {
                int4* const pk = reinterpret_cast<int4*>(ring) + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) pk[64 * t] = Bv[4 + t];
            }
#endif
            {   // half A: samples 32 l + ..., history = the MLO samples before (zero for lane 0)
                const int g0 = 32 * lane;
                const int4 h7 = *reinterpret_cast<const int4*>(&smp[hist7]), h6 = *reinterpret_cast<const int4*>(&smp[hist6]);
                const int4 h5 = *reinterpret_cast<const int4*>(&smp[hist5]);
                const int hs[12] = {h7.w, h7.z, h7.y, h7.x, h6.w, h6.z, h6.y, h6.x, h5.w, h5.z, h5.y, h5.x};
#pragma unroll
                for (int j = 0; j < MLO; ++j) {
                    const float wh = (g0 - 1 - j >= 0) ? win[g0 - 1 - j] : 0.0f;
                    hist[j] = (double)hs[j] * (double)wh;
                }
                // window values four groups at a time (16 registers in flight, not 32: the kernel lives at 168)
#pragma unroll
                for (int tb = 0; tb < 8; tb += FA_F_WBATCH) {
                    WinV wv[FA_F_WBATCH];
#pragma unroll
                    for (int t = 0; t < FA_F_WBATCH; ++t) wv[t] = load_win(g0 + 4 * (tb + t));
#pragma unroll
                    for (int t = 0; t < FA_F_WBATCH; ++t) {
                        lag_group(*reinterpret_cast<const int4*>(&smp[K ^ (4 * (tb + t))]), wv[t]);
                        search_stage(tb + t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if FA_F_PARK
            {   // (the address passes through an asm that also takes the last accumulator: the reads cannot move above half A)
                uint32_t pin = 16u * (uint32_t)lane;
                asm volatile("" : "+v"(pin) : "v"(__double2loint(acc[MLO])));
                const int4* const pk = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(ring) + pin);
#pragma unroll
                for (int t = 0; t < 4; ++t) Bv[4 + t] = pk[64 * t];
            }
#endif
            {   // half B
                const int g0 = 2048 + 32 * lane;
                const int4 h7 = hist_b(Bv[7], *reinterpret_cast<const int4*>(&smp[tailA7]));
                const int4 h6 = hist_b(Bv[6], *reinterpret_cast<const int4*>(&smp[tailA6]));
                int4 h5 = make_int4(0, 0, 0, 0);
                if constexpr (MLO > 8) h5 = hist_b(Bv[5], *reinterpret_cast<const int4*>(&smp[tailA5]));
                const int hs[12] = {h7.w, h7.z, h7.y, h7.x, h6.w, h6.z, h6.y, h6.x, h5.w, h5.z, h5.y, h5.x};
#pragma unroll
                for (int j = 0; j < MLO; ++j) hist[j] = (double)hs[j] * (double)win[g0 - 1 - j];
#pragma unroll
                for (int tb = 0; tb < 8; tb += FA_F_WBATCH) {
                    WinV wv[FA_F_WBATCH];
#pragma unroll
                    for (int t = 0; t < FA_F_WBATCH; ++t) wv[t] = load_win(g0 + 4 * (tb + t));
#pragma unroll
                    for (int t = 0; t < FA_F_WBATCH; ++t) {
                        lag_group(Bv[tb + t], wv[t]);
                        search_stage(8 + tb + t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FA_STAMP(4);
            if (fuse_search) {
                po_fix = fs.bpo;
                k_fix = fs.kb;
                est_fix = 8 + (uint64_t)wasted + (uint64_t)fo * (uint64_t)bps + (uint64_t)fs.best;
                apply_fixed();
            }
            double autoc[MLO + 1];
#if FA_F_BFLY_N
#pragma unroll
            for (int j = 0; j <= MLO; ++j) autoc[j] = acc[j];
            wave_sum_butterfly_n<MLO + 1>(autoc);
#else
#pragma unroll
            for (int j = 0; j <= MLO; ++j) autoc[j] = wave_sum_butterfly(acc[j]);
#endif

            FA_STAMP(5);
            if (__builtin_expect(autoc[0] != 0.0, 1)) {
                float* coef = reinterpret_cast<float*>(scr);         // MLO*MLO floats
                double* err = reinterpret_cast<double*>(scr + 160);  // MLO doubles
                int* meta = reinterpret_cast<int*>(scr + 220);       // usable order
                if (lane == 0) meta[0] = levinson<MLO>(autoc, mlo, coef, err);
                lds_fence();
                const int usable = meta[0];
                int prec = a.precision;
                int lo;
                {
                    double mybits = 1e300;
                    if (lane < usable) {
                        const double e = err[lane];
                        const double error_scale = a.escale_full;
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
                // (wave-uniform values read from LDS: say so, and they live in scalar registers from here to the preamble)
                ok = __builtin_amdgcn_readfirstlane(ok);
                sh = __builtin_amdgcn_readfirstlane(sh);
#pragma unroll
                for (int j = 0; j < MLO; ++j) qreg[j] = __builtin_amdgcn_readfirstlane(qreg[j]);
                FA_STAMP(6);
                if (__builtin_expect(ok != 0, 1)) {
                    FA_IMAGE_ADDRS;
                    const double scale = bitsd((uint64_t)(1023 - sh) << 52);  // 2^-sh (exact pre-scaling, see K3)
                    double qd[MLO];
#pragma unroll
                    for (int j = 0; j < MLO; ++j) qd[j] = (double)qreg[j] * scale;
                    // ---- P4: LPC residual in place (A in LDS, B in registers) + magnitude sums ----
                    double tlA = 0.0, tlB = 0.0;
                    double hx[MLO];
                    // both histories are fetched before either half is overwritten
                    const int4 a7 = *reinterpret_cast<const int4*>(&smp[hist7]), a6 = *reinterpret_cast<const int4*>(&smp[hist6]);
                    const int4 a5 = *reinterpret_cast<const int4*>(&smp[hist5]);
                    const int4 b7 = hist_b(Bv[7], *reinterpret_cast<const int4*>(&smp[tailA7]));
                    const int4 b6 = hist_b(Bv[6], *reinterpret_cast<const int4*>(&smp[tailA6]));
                    int4 b5 = make_int4(0, 0, 0, 0);
                    if constexpr (MLO > 8) b5 = hist_b(Bv[5], *reinterpret_cast<const int4*>(&smp[tailA5]));
                    lds_fence();
                    // Per sample: MLO fma, floor, subtract, |r| into the lane sum, and the zig-zag fold taken in the double
                    // domain -- trunc |2 r + 0.5| is 2 r for r >= 0 and -2 r - 1 for r < 0 (one fma and one conversion
                    // instead of a conversion and three integer instructions).  A residual outside int32 (|r| >= 2^31,
                    // which disqualifies the predictor) comes out as 0xffffffff, a value no valid residual folds to:
                    // that is how the rare frame with such residuals is recognised below, without a per-sample maximum.
                    auto fold_f64 = [&](double r) __attribute__((always_inline)) {
                        return (int)(uint32_t)__builtin_fabs(__builtin_fma(r, 2.0, 0.5));
                    };
                    auto res_group = [&](auto mask_tag, const int4& xv, int gi0, double& tl) __attribute__((always_inline)) {
                        constexpr bool MASK = decltype(mask_tag)::value;
                        const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                        int rs[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double xd = (double)xs[e];
#if FA_F_CEIL
                            // r = x - floor(sum of q x / 2^sh) = ceil(x - sum): the sample seeds the chain (every partial sum is
                            // an exact multiple of 2^-sh below 2^50, so the order of the terms is free)
                            double t = xd;
#pragma unroll
                            for (int j = 0; j < MLO; ++j) t = __builtin_fma(-qd[j], hx[j], t);
                            const double r = fa_ceil(t);
#else
                            double sum = 0.0;
#pragma unroll
                            for (int j = 0; j < MLO; ++j) sum = __builtin_fma(qd[j], hx[j], sum);
                            const double pred = fa_floor(sum);
                            const double r = xd - pred;
#endif
                            if constexpr (MASK) {
                                const bool v = (gi0 + e >= lo);
                                tl += v ? fa_fabs(r) : 0.0;
                                rs[e] = v ? fold_f64(r) : xs[e];
                            } else {
                                tl += fa_fabs(r);
                                rs[e] = fold_f64(r);
                            }
#pragma unroll
                            for (int j = MLO - 1; j > 0; --j) hx[j] = hx[j - 1];
                            hx[0] = xd;
                        }
                        return make_int4(rs[0], rs[1], rs[2], rs[3]);
                    };
                    constexpr int kWarmGroups = (MLO + 3) / 4;
                    {
                        const int hs[12] = {a7.w, a7.z, a7.y, a7.x, a6.w, a6.z, a6.y, a6.x, a5.w, a5.z, a5.y, a5.x};
#pragma unroll
                        for (int j = 0; j < MLO; ++j) hx[j] = (double)hs[j];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            int4* px = reinterpret_cast<int4*>(&smp[K ^ (4 * t)]);
                            if (t < kWarmGroups) *px = res_group(std::true_type{}, *px, 32 * lane + 4 * t, tlA);
                            else *px = res_group(std::false_type{}, *px, 0, tlA);
                        }
                    }
                    {
                        const int hs[12] = {b7.w, b7.z, b7.y, b7.x, b6.w, b6.z, b6.y, b6.x, b5.w, b5.z, b5.y, b5.x};
#pragma unroll
                        for (int j = 0; j < MLO; ++j) hx[j] = (double)hs[j];
#pragma unroll
                        for (int t = 0; t < 8; ++t) Bv[t] = res_group(std::false_type{}, Bv[t], 0, tlB);
                    }
                    FA_STAMP(7);
                    img_is_residual = true;
                    const int pmax = pmax_for(lo);
                    int po_l = 0, k_l = 0;
                    uint64_t rbits;
                    const bool small_l = __all((tlA < 16777216.0) && (tlB < 16777216.0));
                    // every |r| is below its lane's sum: with small sums no residual leaves int32.  Otherwise (wide samples)
                    // look for the mark of an out-of-range residual among the folded values (warm-up samples are not residuals)
                    bool lpc_in_range = true;
                    if (__builtin_expect(!small_l, 0)) {
                        bool hit = false;
#pragma unroll 1
                        for (int t = 0; t < 8; ++t) {
                            const int4 ra = *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]);
                            const int ras[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) hit = hit || ((32 * lane + 4 * t + e >= lo) && (uint32_t)ras[e] == 0xffffffffu);
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            hit = hit || (uint32_t)Bv[t].x == 0xffffffffu || (uint32_t)Bv[t].y == 0xffffffffu || (uint32_t)Bv[t].z == 0xffffffffu ||
                                  (uint32_t)Bv[t].w == 0xffffffffu;
                        lpc_in_range = !__any(hit);
                    }
                    win_small_l = small_l;
                    if (pmax <= 5 && small_l) {
                        SplitRiceSearch ls;
                        ls.start((uint32_t)tlA, (uint32_t)tlB, lo, pmax, lane);
                        ls.gather(); ls.params(); ls.totals();
#pragma unroll
                        for (int o = 5; o >= 0; --o) ls.order(o);
                        po_l = ls.bpo; k_l = ls.kb; rbits = ls.best;
                    } else {
                        rbits = split_rice_search_slow((uint64_t)tlA, (uint64_t)tlB, lo, pmax, lane, &po_l, &k_l);
                    }
                    if (lpc_in_range) {
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

    FA_STAMP(8);
    // (the lane number of the writer half of the kernel is a fresh copy: the one taken at the top would otherwise stay
    // alive across the register peak of the lag loops, and the compiler keeps it in scratch memory there)
    const int lane_w = lane_id_opaque();
    // ---- materialise the winner's residual: LPC is in place; FIXED is recomputed from the samples -------
    auto reload_image = [&]() __attribute__((always_inline)) {
        FA_IMAGE_ADDRS;
#pragma unroll 1
        for (int half = 1; half >= 0; --half) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int4 v = load_row(8 * half + j);
                v.x >>= wasted; v.y >>= wasted; v.z >>= wasted; v.w >>= wasted;
                *reinterpret_cast<int4*>(&smp[rowbase + 256 * j]) = v;
            }
            if (half == 1) {
#pragma unroll
                for (int t = 0; t < 8; ++t) Bv[t] = *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]);
            }
        }
        lds_fence();
    };
    if (__builtin_expect(type == 2, 0)) {
        if (img_is_residual) {
            reload_image();
            img_is_residual = false;
        }
        {   // (order 0: the residual is the sample itself, folded like any other)
            FA_IMAGE_ADDRS;
            const int4 ha = *reinterpret_cast<const int4*>(&smp[hist7]);
            const int4 hb2 = hist_b(Bv[7], *reinterpret_cast<const int4*>(&smp[tailA7]));
            lds_fence();
            int64_t x1, x2, x3, x4;
            auto fgroup = [&](const int4& xv, int gi0) __attribute__((always_inline)) {
                const int xs[4] = {xv.x, xv.y, xv.z, xv.w};
                int rs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t x0 = xs[e];
                    int64_t r;
                    if (order == 0) r = x0;
                    else if (order == 1) r = x0 - x1;
                    else if (order == 2) r = x0 - 2 * x1 + x2;
                    else if (order == 3) r = x0 - 3 * x1 + 3 * x2 - x3;
                    else r = x0 - 4 * x1 + 6 * x2 - 4 * x3 + x4;
                    rs[e] = (gi0 + e >= order) ? rice_fold((int)r) : xs[e];
                    x4 = x3; x3 = x2; x2 = x1; x1 = x0;
                }
                return make_int4(rs[0], rs[1], rs[2], rs[3]);
            };
            x1 = ha.w; x2 = ha.z; x3 = ha.y; x4 = ha.x;
#pragma unroll 1
            for (int t = 0; t < 8; ++t) {
                int4* px = reinterpret_cast<int4*>(&smp[K ^ (4 * t)]);
                *px = fgroup(*px, 32 * lane_w + 4 * t);
            }
            x1 = hb2.w; x2 = hb2.z; x3 = hb2.y; x4 = hb2.x;
#pragma unroll
            for (int t = 0; t < 8; ++t) Bv[t] = fgroup(Bv[t], 4096);
            img_is_residual = true;
        }
        lds_fence();
    }

    // ---- exact size of the winner (Rice parameters are uniform over a lane_w's chunk), VERBATIM fallback ----
    bool rice2 = false;
    if (type >= 2) rice2 = __any((lane_w < (1 << porder)) && (kbest >= 15));
    const int plen = rice2 ? 5 : 4;
    uint32_t sub_bits;  // bits of the subframe
    if (type >= 2) {
        FA_IMAGE_ADDRS;
        const int pA = (lane_w << porder) >> 7, pB = ((64 + lane_w) << porder) >> 7;
        const uint32_t kA = (uint32_t)__builtin_amdgcn_ds_bpermute(pA << 2, kbest);
        const uint32_t kB = (uint32_t)__builtin_amdgcn_ds_bpermute(pB << 2, kbest);
        const uint32_t cpp = 128u >> porder;  // chunks per partition
        uint32_t bitsA = 0, bitsB = 0;
        // quotient lengths.  A lane_w's magnitude sums below 2^24 (the usual case, known from the partition search) bound
        // its 64 quotients by 2^25 in total: plain adds.  Otherwise every quotient is clamped (a code that long
        // overflows its row anyway: the totals stay small and the row test below sends the frame to VERBATIM).
        auto size_group = [&](auto mask_tag, auto clamp_tag, const int4& rv, uint32_t k, int gi0, uint32_t& acc) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(mask_tag)::value;
            constexpr bool CLAMP = decltype(clamp_tag)::value;
            const int rs[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t u = (uint32_t)rs[e];  // (folded where the residual was computed)
                uint32_t q = u >> k;
                if constexpr (CLAMP) q = min(q, 16384u);
                if constexpr (MASK) q = (gi0 + e >= order) ? (q + k + 1u) : 0u;
                acc += q;
            }
        };
        constexpr int kWarmGroupsS = (MLO + 3) / 4 > 1 ? (MLO + 3) / 4 : 1;  // (fixed orders <= 4 fit the first group)
        auto size_lane = [&](auto clamp_tag) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int4 rv = *reinterpret_cast<const int4*>(&smp[K ^ (4 * t)]);
                if (t < kWarmGroupsS) size_group(std::true_type{}, clamp_tag, rv, kA, 32 * lane_w + 4 * t, bitsA);
                else size_group(std::false_type{}, clamp_tag, rv, kA, 0, bitsA);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) size_group(std::false_type{}, clamp_tag, Bv[t], kB, 0, bitsB);
        };
        if (__builtin_expect((type == 3) ? win_small_l : win_small_f, 1)) size_lane(std::false_type{});
        else size_lane(std::true_type{});
        bitsA += (8 - kWarmGroupsS) * 4 * (kA + 1u);
        bitsB += 32u * (kB + 1u);
        // partition parameters: the chunk that opens a partition carries them (partition 0 opens in chunk A_0)
        if (((uint32_t)lane_w & (cpp - 1u)) == 0u || cpp > 64u) {
            if (cpp <= 64u) { bitsA += (uint32_t)plen; bitsB += (uint32_t)plen; }
            else if (lane_w == 0) bitsA += (uint32_t)plen;
        }
        // row totals (8 lanes per row) against the row cap, frame total
        uint32_t rA = bitsA, rB = bitsB;
        rA += (uint32_t)xchg_i32<0>((int)rA); rB += (uint32_t)xchg_i32<0>((int)rB);
        rA += (uint32_t)xchg_i32<1>((int)rA); rB += (uint32_t)xchg_i32<1>((int)rB);
        rA += (uint32_t)xchg_i32<2>((int)rA); rB += (uint32_t)xchg_i32<2>((int)rB);
        const bool row_over = __any((rA > (uint32_t)kRowCapBits) || (rB > (uint32_t)kRowCapBits));
        const uint64_t exact = 6 + wave_sum_u64((uint64_t)bitsA + bitsB) + 8 + (uint64_t)wasted + (uint64_t)order * (uint64_t)bps +
                               ((type == 3) ? (4 + 5 + (uint64_t)order * (uint64_t)precision) : 0);
        if (row_over || exact > verbatim_bits) {
            type = 1;
            order = 0;
            porder = 0;
        }
        sub_bits = (uint32_t)exact;
    }
    if (type == 0) sub_bits = 8u + (uint32_t)wasted + (uint32_t)bps;
    if (type == 1) sub_bits = (uint32_t)verbatim_bits;
    const uint32_t total_bytes = ((fh_bits + sub_bits + 7u) >> 3) + 2u;
    const uint32_t L = total_bytes - 2u;  // bytes covered by the CRC-16

    FA_STAMP(9);
    // ---- publish the size ------------------------------------------------------------------------------
    const uint32_t F = (uint32_t)a.total_frames;
    const uint32_t gu = (uint32_t)g;
#ifdef FA_TIMELINE
    // diagnostic build: the optional FrameInfo record carries 100 MHz timestamps instead of the decisions
    // (wasted = start, shift = size published, porder = offset out (written by the scanner), precision = offset asked
    // for, blocksize = offset seen)
    if (lane_w == 0 && a.info) {
        a.info[g].wasted = (int32_t)tl_start_;
        a.info[g].shift = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0);
    }
#endif
    if (lane_w == 0) {
        __hip_atomic_store(a.size_pub + gu, 0x80000000u | total_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.frame_bytes[g] = total_bytes;
    }
    (void)F;
#ifndef FA_TIMELINE
    if (lane_w == 0 && a.info) {
        FrameInfo fi;
        fi.type = type;
        fi.order = (type >= 2) ? order : 0;
        fi.porder = (type >= 2) ? porder : 0;
        fi.wasted = wasted;
        fi.shift = (type == 3) ? shift : 0;
        fi.precision = (type == 3) ? precision : 0;
        fi.nbytes = (int32_t)total_bytes;
        fi.blocksize = bs;
        a.info[g] = fi;
    }
#endif

#ifdef FA_F_ANALYSIS_ONLY  // timing experiment: what the analysis (everything up to the published size) costs on its own
    // (no frame is written; frame_abs gets the frame's SLOT position, so that the stream-header kernel that follows
    // stays inside the capacity buffer -- an unset frame_abs would send it out of bounds)
    if (lane_w == 0) a.frame_abs[g] = g * (int64_t)kSlotBytes + (s + 1) * a.hb;
    return;
#endif
    // ---- writer state ------------------------------------------------------------------------------------
    kpar[lane_w] = (uint8_t)kbest;
    for (int i = lane_w; i < kFRingWords; i += 64) ring[i] = 0;
    if (lane_w == 0) ring[kFRingWords] = 0;
    lds_fence();
    uint32_t pos = 0;
    uint32_t blocks_flushed = 0;
    uint8_t* dst = nullptr;  // final position of the frame, known after the look-back
    bool dropped = false, have_dst = false;
    uint32_t crc_t = 0;  // this lane_w's running CRC state (transformed domain, see crc tables)

    auto put_bits = [&](uint32_t P, uint32_t val, uint32_t nb) __attribute__((always_inline)) {
        const uint32_t off = P & 31u;
        const uint64_t X = (uint64_t)val << (64u - nb - off);
        const uint32_t a0 = (P >> 3) & (uint32_t)(4 * kFRingWords - 4);
        atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ring) + a0), (uint32_t)(X >> 32));
        atomicOr(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ring) + a0 + 4), (uint32_t)X);  // may be the mirror word
    };
    // One Rice code whose stop bit lands at stream bit P: the stop bit and the k low bits of u, left-aligned in a word
    // (shl = 31 - k; whatever u holds above bit k is shifted out or falls on the stop bit), then funnel-shifted to
    // P mod 32 -- v_alignbit takes the shift modulo 32, so there is no mask, no 64-bit shift and no subtraction.
    auto put_code = [&](uint32_t P, uint32_t u, uint32_t shl) __attribute__((always_inline)) {
        const uint32_t vL = (u << shl) | 0x80000000u;
        uint32_t wi;  // (asm: the compiler rewrites the bit-field extract into shift + mask and then needs a separate add)
        asm("v_bfe_u32 %0, %1, 5, %2" : "=v"(wi) : "v"(P), "n"(__builtin_ctz((unsigned)kFRingWords)));
        uint32_t* const w = ring + wi;
        atomicOr(w, __builtin_amdgcn_alignbit(0u, vL, P));
        atomicOr(w + 1, __builtin_amdgcn_alignbit(vL, 0u, P));  // may be the mirror word
    };
    // byte offset of the frame: the scanner stores it in off_pub[g] once every frame before it has published.
    // lb_issue reads the word once (usually still zero); lb_resolve, where the first block is about to leave the
    // ring, polls it until it is there.
    unsigned long long off_word = 0;
    auto lb_issue = [&]() __attribute__((always_inline)) {
#ifndef FA_F_NOLB
        off_word = lb_load(a.off_pub + gu);
#endif
    };
    auto lb_resolve = [&]() __attribute__((always_inline)) {
        uint32_t spins = 0;
        bool fail = false;
#ifdef FA_TIMELINE
        const uint32_t tl_req_ = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef FA_F_NOLB  // timing experiment only (frames land at slot positions): what the wait costs
        off_word = (unsigned long long)(g * (int64_t)kSlotBytes + (s + 1) * a.hb);
#else
        while (off_word == 0) {
            __builtin_amdgcn_s_sleep(FA_F_SLEEP);
            if (++spins > kLbSpinLimit) { fail = true; break; }
            off_word = lb_load(a.off_pub + gu);
        }
#endif
#ifdef FA_TIMELINE
        if (lane_w == 0 && a.info) {
            a.info[g].precision = (int32_t)tl_req_;
            a.info[g].blocksize = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();
        }
#endif
#ifdef FA_STAMPS
        st_[3] += spins;                 // polls
#endif
        // (every lane_w loaded the same word: say so, the destination then lives in scalar registers)
        const int64_t off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off_word >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off_word));
        fail = __builtin_amdgcn_readfirstlane((int)fail) != 0;
        if (fail || off < 0 || off + (int64_t)total_bytes > a.capacity) {
            if (lane_w == 0) atomicOr(a.err, fail ? 2 : 1);
            dropped = true;  // the frame is not written; the host reports the error
        } else {
            dst = a.blob + off;
            if (lane_w == 0) a.frame_abs[g] = off;
        }
        have_dst = true;
    };
    // CRC-16 over interleaved words: every lane_w folds its word of each 256-byte block; the old state enters through
    // the top half of the word (tables pre-multiplied by x^2016, so that equals advancing it by 256 bytes)
    auto crc_word = [&](uint32_t wv) __attribute__((always_inline)) {
        const uint32_t w = wv ^ (crc_t << 16);
        const uint32_t t0 = crc_s[w >> 24], t1 = crc_s[256 + ((w >> 16) & 255u)], t2 = crc_s[512 + ((w >> 8) & 255u)], t3 = crc_s[768 + (w & 255u)];
        // (the four table entries are zero-extended 16-bit loads: full-width xors keep the state clean without the
        // 16-bit operation + mask the compiler otherwise emits)
        uint32_t x01;
        asm("v_xor_b32 %0, %1, %2" : "=v"(x01) : "v"(t0), "v"(t1));
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(crc_t) : "v"(x01), "v"(t2), "v"(t3));
    };
    // Stores are destination-ALIGNED dwords (an unaligned dword store becomes partial-line writes: the first version
    // wrote 2.6x the blob's bytes to the fabric, profiles/r02h_traffic.json).  The frame starts at dst, sh = dst & 3
    // bytes past a dword boundary; aligned dword m (address dst - sh + 4 m) holds stream bytes [4 m - sh, 4 m - sh + 4),
    // i.e. the last sh bytes of ring word m - 1 and the first 4 - sh of ring word m (big-endian words: one v_alignbit).
    // Lane l takes word m - 1 from lane_w l - 1 (DPP), lane_w 0 from `carry`, the last word of the previous block.  Dwords
    // that reach outside the frame's bytes [0, total_bytes) -- its first and last -- are written byte by byte: the
    // neighbouring frames own the rest of them.
    uint32_t carry = 0;
    auto emit_word = [&](uint32_t m, uint32_t Q) __attribute__((always_inline)) {  // all lanes: ring word m = m0 + lane_w (0 beyond the frame)
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
        const uint32_t P = (uint32_t)dpp_wave_shr1((int)carry, (int)Q);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)Q, 63);
        const uint32_t V = __builtin_amdgcn_alignbit(P, Q, 8u * sh);  // big-endian value of stream bytes [lo, lo + 4)
        const int lo = (int)(4u * m) - (int)sh;
        if (lo >= 0 && (uint32_t)lo + 4u <= total_bytes) {
            *reinterpret_cast<uint32_t*>(dst + lo) = __builtin_bswap32(V);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = lo + i;
                if (b >= 0 && (uint32_t)b < total_bytes) dst[b] = (uint8_t)(V >> (24 - 8 * i));
            }
        }
    };
    // A complete 256-byte block behind the frame's first lies inside [0, total_bytes) whatever the alignment: its 64
    // aligned dwords need no range tests (emit_word's byte-wise path serves the frame's first and last dwords only).
    auto emit_block = [&](uint32_t blk, uint32_t Q) __attribute__((always_inline)) {
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
        const uint32_t P = (uint32_t)dpp_wave_shr1((int)carry, (int)Q);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)Q, 63);
        const uint32_t V = __builtin_amdgcn_alignbit(P, Q, 8u * sh);
        *reinterpret_cast<uint32_t*>(dst - sh + 256u * blk + 4u * (uint32_t)lane_w) = __builtin_bswap32(V);
    };
    // Called BEFORE a row is written, with the row's exact length (the scan has it before the first code is placed):
    // completed blocks leave the ring only when the row would not fit beside them -- the first bit after the row must stay
    // less than a ring away from the oldest block still held (which also keeps the mirror word to one straddling code at
    // a time).
    //
    // Where they go depends on whether the frame's byte offset is known yet.  It usually is not when the ring first
    // fills up: the offset needs every frame before this one to have published its size, and among the ~3000 frames in
    // flight some straggler is always ~8 us behind ("the wait").  Instead of waiting there, the frame moves
    // its completed blocks -- CRC folded on the way -- into REGISTERS: one block is one register across the wave, the
    // analysis is over and a third of the register file is idle, so 32 blocks (8 KB; with the ring 12 KB, more than an
    // average frame) cost no memory traffic and no LDS.  The offset word is polled without waiting: a load issued at
    // one call is looked at by the next.  Once it is there the parked blocks are stored first, in order (the carry of
    // the destination-aligned stores runs through them), and the rest goes straight from the ring as before.  Only a
    // frame that has run out of registers as well, or has nothing left to do, spins.
#if FA_F_PARKBLK
    uint32_t park[FA_F_PARKN];
#endif
    uint32_t n_parked = 0;
    auto take_block = [&](uint32_t blk) __attribute__((always_inline)) {  // ring block blk -> word per lane_w, slot cleared
        const uint32_t wi = (blk * 64 + lane_w) & kFRingMask;
        uint32_t wv = ring[wi];
        ring[wi] = 0;
        if (__builtin_expect((blk & (uint32_t)(kFRingBlocks - 1)) == 0, 0)) {
            if (lane_w == 0) { wv |= ring[kFRingWords]; ring[kFRingWords] = 0; }
        }
        return wv;
    };
    auto flush_blocks = [&](uint32_t row_bits, bool final_call) __attribute__((always_inline)) {
        const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pos >> 11));
        blocks_flushed = (uint32_t)__builtin_amdgcn_readfirstlane((int)blocks_flushed);
        const uint32_t hi_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)((pos + row_bits) >> 11));
        if (!final_call && hi_blk - blocks_flushed < (uint32_t)kFRingBlocks) return;
#ifdef FA_STAMPS
        const unsigned long long tq0_ = fa_memtime();
#endif
        if (!have_dst) {
#if FA_F_PARKBLK
            if (!final_call && off_word == 0) {  // (the load issued at the previous call, or at publish time)
                off_word = lb_load(a.off_pub + gu);  // looked at by the next call
                n_parked = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_parked);
                while (blocks_flushed < done && n_parked < (uint32_t)FA_F_PARKN) {
                    const uint32_t wv = take_block(blocks_flushed);
                    crc_word(wv);
                    park[n_parked] = wv;
                    n_parked++;
                    blocks_flushed++;
                }
                if (hi_blk - blocks_flushed < (uint32_t)kFRingBlocks) {
#ifdef FA_STAMPS
                    st_[14] += fa_memtime() - tq0_;
#endif
                    return;  // the row fits now
                }
            }
#endif
            lb_resolve();
        }
#ifdef FA_STAMPS
        st_[14] += fa_memtime() - tq0_;  // (part of the flush calls: waiting for the frame's offset)
#endif
#if FA_F_PARKBLK
        // the parked blocks first: blocks 0 .. n_parked - 1 of the frame
        n_parked = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_parked);
        if (__builtin_expect(n_parked != 0, 0)) {
            if (!dropped) {
                emit_word((uint32_t)lane_w, park[0]);  // (the frame's first dword may be partial: byte-wise edges)
                for (uint32_t i = 1; i < n_parked; ++i) emit_block(i, park[i]);
            }
            n_parked = 0;
        }
#endif
        // the frame's first block, once: its first dword may be partial (emit_word's byte-wise edges).  Kept out of the
        // loop below, whose preheader would otherwise rebuild emit_word's lane_w masks and byte addresses at every call.
        if (__builtin_expect(blocks_flushed == 0 && done > 0, 0)) {
            const uint32_t wv = take_block(0);
            crc_word(wv);
            if (!dropped) emit_word((uint32_t)lane_w, wv);
            blocks_flushed = 1;
            __builtin_amdgcn_sched_barrier(0);
        }
        while (blocks_flushed < done) {
            const uint32_t wv = take_block(blocks_flushed);
            crc_word(wv);
            if (!dropped) emit_block(blocks_flushed, wv);
            blocks_flushed++;
        }
    };

#if FA_F_PARKBLK
    // (the registers of `park` begin their life HERE, with whatever they hold: left undefined, the compiler carries the
    // "value" of the array from the top of the kernel and spills part of it around the analysis)
#pragma unroll
    for (int i = 0; i < FA_F_PARKN; ++i) asm volatile("" : "=v"(park[i]));
#endif
    lb_issue();
    // ---- preamble: frame header, subframe header, warm-up, LPC fields, residual header ------------------
    {
        const int tc = (type == 0) ? 0x00 : (type == 1) ? 0x01 : (type == 2) ? (0x08 | order) : (0x20 | (order - 1));
        const uint32_t smask = (bps == 32) ? 0xffffffffu : ((1u << bps) - 1u);
        const int nwarm = (type == 0) ? 1 : (type >= 2) ? order : 0;
        constexpr int kWarmLanes = (MLO > 4) ? MLO : 4;
        constexpr int kL_warm = 6, kL_lpc = kL_warm + kWarmLanes, kL_coef = kL_lpc + 1, kL_rice = kL_coef + MLO;
        static_assert(kL_rice < 64, "preamble fields must fit the wave");
        uint32_t fv = 0, fnb = 0;
        if (lane_w == 0) { fv = fhe.x; fnb = 32; }
        else if (lane_w == 1) { fv = fhe.y; fnb = (fhe.w >> 8) & 0xFFu; }
        else if (lane_w == 2) { fv = fhe.z & 0xFFFFu; fnb = (fhe.w >> 16) & 0xFFu; }
        else if (lane_w == 3) { fv = fhe.z >> 16; fnb = fhe.w >> 24; }
        else if (lane_w == 4) { fv = ((fhe.w & 0xFFu) << 8) | (uint32_t)((tc << 1) | (wasted ? 1 : 0)); fnb = 16; }
        else if (lane_w == 5) { if (wasted) { fv = 1; fnb = (uint32_t)wasted; } }
        else if (lane_w < kL_lpc) {
            if (lane_w - kL_warm < nwarm) {
                // warm-up samples are original samples: the residual passes leave them in place; a constant frame's
                // sample comes from the image as staged
                fv = (uint32_t)smp[fsmp_idx(lane_w - kL_warm)] & smask;
                fnb = (uint32_t)bps;
            }
        }
        else if (lane_w == kL_lpc) { if (type == 3) { fv = ((uint32_t)(precision - 1) << 5) | (uint32_t)shift; fnb = 9; } }
        else if (lane_w < kL_rice) {
            if (type == 3 && lane_w - kL_coef < order) {
                int32_t q = 0;
#pragma unroll
                for (int j = 0; j < MLO; ++j) q = (lane_w - kL_coef == j) ? qkeep[j] : q;
                fv = (uint32_t)q & ((1u << precision) - 1u);
                fnb = (uint32_t)precision;
            }
        }
        else if (lane_w == kL_rice) { if (type >= 2) { fv = ((rice2 ? 1u : 0u) << 4) | (uint32_t)porder; fnb = 6; } }
        const uint32_t incl = wave_incl_scan_u32(fnb);
        if (fnb) put_bits(incl - fnb, fv, fnb);
        pos = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    lds_fence();
    FA_STAMP(10);

    // ---- rows -------------------------------------------------------------------------------------------
    if (__builtin_expect(type == 1, 0)) {
        // VERBATIM: rows straight from global memory (the image may hold a residual)
        const uint32_t mask = (bps == 32) ? 0xffffffffu : ((1u << bps) - 1u);
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
            const int4 rv = load_row(j);
            const int rs[4] = {rv.x >> wasted, rv.y >> wasted, rv.z >> wasted, rv.w >> wasted};
            flush_blocks((uint32_t)(256 * bps), false);
            uint32_t p = pos + (uint32_t)(4 * bps) * (uint32_t)lane_w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                put_bits(p, (uint32_t)rs[e] & mask, (uint32_t)bps);
                p += (uint32_t)bps;
            }
            pos += (uint32_t)(256 * bps);
        }
    } else if (type >= 2) {
        FA_IMAGE_ADDRS;
        const uint32_t ps = (uint32_t)(bs >> porder);
        const int l2ps = 12 - porder;
        // row j (0..15) is read row-major from the image: rows 0..7 hold the first half, then the second half is
        // written over them (store_b) and rows 8..15 read the same addresses
        auto store_b = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<int4*>(&smp[K ^ (4 * t)]) = Bv[t];
        };
        // Rows of 512 samples, 8 consecutive samples per lane_w (two 16-byte units of one chunk): the per-row work -- the
        // scan of the lane_w lengths, the parameter lookup, the flush test -- is paid 8 times per frame instead of 16.
        // (The 12288-bit cap of the specification stays a property of 256-sample rows; the sizing pass checked it.)
        {
            const int cb = (ln_ >> 2) + 1;
            const int r8a = 32 * cb + 4 * ((2 * (ln_ & 3)) ^ (cb & 7)), r8b = 32 * cb + 4 * ((2 * (ln_ & 3) + 1) ^ (cb & 7));
            auto row8 = [&](auto first_tag, int j) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;  // row 0: warm-up samples carry no code
                const uint32_t gb = (uint32_t)(512 * j + 8 * lane_w);
                const int4 va = *reinterpret_cast<const int4*>(&smp[r8a + 512 * (j & 3)]);
                const int4 vb = *reinterpret_cast<const int4*>(&smp[r8b + 512 * (j & 3)]);
                const uint32_t us[8] = {(uint32_t)va.x, (uint32_t)va.y, (uint32_t)va.z, (uint32_t)va.w,
                                        (uint32_t)vb.x, (uint32_t)vb.y, (uint32_t)vb.z, (uint32_t)vb.w};
                const uint32_t pidx = gb >> l2ps;
                const uint32_t k = kpar[pidx], kp1 = k + 1u;
                // partition 0 opens at sample `order`, the others at multiples of the partition size (>= 64)
                const uint32_t pstart = (pidx == 0u) ? (uint32_t)order : (pidx << l2ps);
                const bool newp = FIRST ? (gb <= pstart && pstart < gb + 8u) : ((gb & (ps - 1u)) == 0u);
                uint32_t q[8];
                uint32_t len = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    uint32_t qq = us[e] >> k;
                    if constexpr (FIRST) {
                        const uint32_t gi = gb + (uint32_t)e;
                        const bool valid = gi >= (uint32_t)order;
                        qq += (gi == pstart) ? (uint32_t)plen : 0u;
                        q[e] = valid ? qq : 0xffffffffu;  // marks "no code"
                        len += valid ? (qq + kp1) : 0u;
                    } else {
                        if (e == 0) qq += newp ? (uint32_t)plen : 0u;
                        q[e] = qq;
                        len += qq;
                    }
                }
                const uint32_t lane_len = FIRST ? len : (len + 8u * kp1);
                const uint32_t incl = wave_incl_scan_u32(lane_len);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#ifdef FA_STAMPS
                const unsigned long long tf0_ = fa_memtime();
#endif
                flush_blocks(total, false);
#ifdef FA_STAMPS
                st_[15] += fa_memtime() - tf0_;  // (part of "rows": the flush calls)
#endif
                const uint32_t onek = 1u << k, mask = onek - 1u;
                const uint32_t p0 = pos + incl - lane_len;
                uint32_t p = p0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (FIRST) {
                        if (q[e] != 0xffffffffu) {
                            if (gb + (uint32_t)e == pstart) put_bits(p, k, (uint32_t)plen);
                            put_bits(p + q[e], onek | (us[e] & mask), kp1);
                            p += q[e] + kp1;
                        }
                    } else {
                        put_code(p + q[e], us[e], 31u - k);
                        p += q[e] + kp1;
                    }
                }
                if constexpr (!FIRST) {
                    if (newp) put_bits(p0, k, (uint32_t)plen);
                }
                pos += total;
            };
            row8(std::true_type{}, 0);
#pragma unroll 1
            for (int j = 1; j < 8; ++j) {
                if (j == 4) store_b();  // rows 0..3 (the first half) have been read: the second half takes their place
                row8(std::false_type{}, j);
            }
        }
    }

    FA_STAMP(11);
    // ---- tail: byte align, CRC-16, final words -----------------------------------------------------------
    flush_blocks(0u, true);  // the offset is resolved (waiting if need be), parked blocks and every complete block go out
    if (!have_dst) lb_resolve();
    FA_STAMP(13);
    {
        // words not flushed yet: [64 * blocks_flushed, nwords); the CRC covers bytes [0, L)
        const uint32_t nwords = (total_bytes + 3u) >> 2;
        uint32_t last_end = 256u * blocks_flushed - 256u + 4u * (uint32_t)lane_w + 4u;  // end of this lane_w's last folded word (if any block was flushed)
        bool any = blocks_flushed > 0;
        for (uint32_t w0 = blocks_flushed * 64; w0 < nwords; w0 += 64) {
            const uint32_t wl = w0 + lane_w;
            if (4u * wl < L) {
                uint32_t wv = ring[wl & kFRingMask];
                if ((wl & kFRingMask) == 0) wv |= ring[kFRingWords];
                crc_word(wv);
                last_end = 4u * wl + 4u;
                any = true;
            }
        }
        // lane_w states -> CRC: state * x^(8 (L - last_end) - 2016); xpow[i] = x^(8 (i - 255))
        uint32_t contrib = 0;
        if (any) {
            const int after = (int)L - (int)last_end;  // -3 .. 511
            contrib = crc16_mulmod((uint16_t)crc_t, a.crc_tab[kFCrcSlice + after + 3]);
        }
        contrib ^= (uint32_t)xchg_i32<0>((int)contrib);
        contrib ^= (uint32_t)xchg_i32<1>((int)contrib);
        contrib ^= (uint32_t)xchg_i32<2>((int)contrib);
        contrib ^= (uint32_t)xchg_i32<3>((int)contrib);
        contrib ^= (uint32_t)xchg_i32<4>((int)contrib);
        const uint32_t crc = ((uint32_t)__builtin_amdgcn_readlane((int)contrib, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)contrib, 32)) & 0xFFFFu;
        lds_fence();
        if (lane_w == 0) put_bits(8u * L, crc, 16);
        lds_fence();
        if (!dropped) {
            // the remaining words, and one word of zeros after them: the aligned dword that holds the frame's last
            // (up to 3) bytes starts in the last ring word
            for (uint32_t w0 = blocks_flushed * 64; w0 <= nwords; w0 += 64) {
                const uint32_t wl = w0 + lane_w;
                uint32_t wv = 0;
                if (wl < nwords) {
                    wv = ring[wl & kFRingMask];
                    if ((wl & kFRingMask) == 0) wv |= ring[kFRingWords];
                }
                emit_word(wl, wv);
            }
        }
    }
    FA_STAMP(12);
#ifdef FA_STAMPS
    if (lane_w == 0 && a.stamps && (blockIdx.x % 61u) == 0) {  // per XCD: frames, offset wait, lifetime (slots 32 + 4 x)
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        unsigned long long life = 0;
        for (int i_ = 0; i_ < 13; ++i_) life += st_[i_];
        atomicAdd(&a.stamps[32 + 4 * xcc], 1ULL);
        atomicAdd(&a.stamps[33 + 4 * xcc], st_[14]);
        atomicAdd(&a.stamps[34 + 4 * xcc], life);
        unsigned long long prep = 0;
        for (int i_ = 0; i_ < 10; ++i_) prep += st_[i_];
        atomicAdd(&a.stamps[35 + 4 * xcc], prep);  // start -> publish
    }
#endif
    FA_STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------
// after K3F: per-stream starts / nbytes, total, stream headers (fLaC, STREAMINFO, SEEKTABLE)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fused_finish_kernel(uint8_t* __restrict__ out, const int64_t* __restrict__ frame_abs,
                                                           const uint32_t* __restrict__ frame_bytes, int64_t n_stream, int64_t nframes,
                                                           int64_t stream_size, int32_t B, int32_t tail_bs, int32_t nch, int64_t hb,
                                                           int64_t* __restrict__ starts, int64_t* __restrict__ nbytes,
                                                           int64_t* __restrict__ total) {
    const int64_t s = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t first_abs = frame_abs[s * nframes];
    const int64_t st = first_abs - hb;
    const int64_t end = (s + 1 < n_stream) ? (frame_abs[(s + 1) * nframes] - hb)
                                          : (frame_abs[n_stream * nframes - 1] + (int64_t)frame_bytes[n_stream * nframes - 1]);
    uint8_t* h = out + st;
    if (tid == 0) {
        starts[s] = st;
        nbytes[s] = end - st;
        if (s + 1 == n_stream) *total = end;
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
        const uint64_t off = (uint64_t)(frame_abs[s * nframes + f] - first_abs);
        const int bsz = (f == nframes - 1) ? tail_bs : B;
        for (int i = 0; i < 8; ++i) { p[i] = (uint8_t)(sn >> (56 - 8 * i)); p[8 + i] = (uint8_t)(off >> (56 - 8 * i)); }
        p[16] = (uint8_t)(bsz >> 8);
        p[17] = (uint8_t)bsz;
    }
}

// short last frames (stream lengths that are not a multiple of 4096): sizes in before K3F, bytes moved after it.
// tail_publish: size_pub of every stream's last frame from the slot encoder's frame_bytes.
// tail_prep: after K3F every frame has its offset -- arguments for compact_frames_kernel, called with "one frame per
// stream" geometry (its header size is then 64 bytes, folded into the offsets), and frame_abs for the stream headers.
FA_GLOBAL __global__ __launch_bounds__(256) void fused_tail_publish_kernel(const uint32_t* __restrict__ frame_bytes, uint32_t* __restrict__ size_pub,
                                                                           int64_t n_stream, int64_t nframes) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_stream) return;
    const int64_t g = s * nframes + nframes - 1;
    __hip_atomic_store(size_pub + g, 0x80000000u | frame_bytes[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FA_GLOBAL __global__ __launch_bounds__(256) void fused_tail_prep_kernel(const unsigned long long* __restrict__ off_pub,
                                                                        const uint32_t* __restrict__ frame_bytes, int64_t n_stream, int64_t nframes,
                                                                        int64_t* __restrict__ frame_abs, uint32_t* __restrict__ tail_bytes,
                                                                        int64_t* __restrict__ tail_off, int64_t* __restrict__ zeros) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_stream) return;
    const int64_t g = s * nframes + nframes - 1;
    const int64_t off = (int64_t)off_pub[g];
    frame_abs[g] = off;
    tail_bytes[s] = frame_bytes[g];
    tail_off[s] = off - stream_header_bytes(1);
    zeros[s] = 0;
}

#endif  // the kernels

// host-side launchers (defined in the unit that holds the kernels: csrc/fused_unit.hip in the shipped build, compiled
// with the default scheduling strategy -- max-ILP, which the slot encoder and the decoder like, costs this kernel
// registers it does not have)
void launch_fused_encode(hipStream_t st, const FusedArgs& a, bool f32)
#if defined(FA_UNIT_FUSED) || !defined(FA_SPLIT_UNITS)
{
    const dim3 grid((unsigned)((a.total_frames + kFWaves - 1) / kFWaves) + 1u);  // + the scanner's workgroup
    const dim3 block(64 * kFWaves);
#ifdef FA_DEV_MINIMAL  // diagnostic builds: the level 3-5 kernels only
    if (f32) hipLaunchKernelGGL((encode_fused_kernel<8, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((encode_fused_kernel<8, false>), grid, block, 0, st, a);
#else
    if (f32) {
        switch (a.max_lpc_order) {
            case 6: hipLaunchKernelGGL((encode_fused_kernel<6, true>), grid, block, 0, st, a); break;
            case 8: hipLaunchKernelGGL((encode_fused_kernel<8, true>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((encode_fused_kernel<12, true>), grid, block, 0, st, a); break;
        }
    } else {
        switch (a.max_lpc_order) {
            case 6: hipLaunchKernelGGL((encode_fused_kernel<6, false>), grid, block, 0, st, a); break;
            case 8: hipLaunchKernelGGL((encode_fused_kernel<8, false>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((encode_fused_kernel<12, false>), grid, block, 0, st, a); break;
        }
    }
#endif
}
#else
;
#endif
void launch_fused_finish(hipStream_t st, uint8_t* out, const int64_t* frame_abs, const uint32_t* frame_bytes, int64_t n_stream,
                         int64_t nframes, int64_t stream_size, int32_t B, int32_t tail_bs, int32_t nch, int64_t hb, int64_t* starts,
                         int64_t* nbytes, int64_t* total)
#if defined(FA_UNIT_FUSED) || !defined(FA_SPLIT_UNITS)
{
    hipLaunchKernelGGL(fused_finish_kernel, dim3((unsigned)n_stream), dim3(256), 0, st, out, frame_abs, frame_bytes, n_stream, nframes,
                       stream_size, B, tail_bs, nch, hb, starts, nbytes, total);
}
#else
;
#endif
void launch_fused_tail_publish(hipStream_t st, const uint32_t* frame_bytes, uint32_t* size_pub, int64_t n_stream, int64_t nframes)
#if defined(FA_UNIT_FUSED) || !defined(FA_SPLIT_UNITS)
{
    hipLaunchKernelGGL(fused_tail_publish_kernel, dim3((unsigned)((n_stream + 255) / 256)), dim3(256), 0, st, frame_bytes, size_pub, n_stream, nframes);
}
#else
;
#endif
void launch_fused_tail_prep(hipStream_t st, const unsigned long long* off_pub, const uint32_t* frame_bytes, int64_t n_stream, int64_t nframes,
                            int64_t* frame_abs, uint32_t* tail_bytes, int64_t* tail_off, int64_t* zeros)
#if defined(FA_UNIT_FUSED) || !defined(FA_SPLIT_UNITS)
{
    hipLaunchKernelGGL(fused_tail_prep_kernel, dim3((unsigned)((n_stream + 255) / 256)), dim3(256), 0, st, off_pub, frame_bytes, n_stream, nframes,
                       frame_abs, tail_bytes, tail_off, zeros);
}
#else
;
#endif

}  // namespace fa
