// encode_placed.hpp -- K3G: single-pass FLAC encode for every geometry K3F does not take (gfx950, wave64).
//
// K3F (encode_fused.hpp) covers full 4096-sample mono frames of levels 3-8.  Everything else the reference's arrays
// produce -- streams shorter than two frames ((12, 1000), (1, 10000): src/flacarray/tests/bindings.py:165-230), the
// 1152-sample blocks of levels 0-2, streams whose length is not a multiple of 4, rows that are not 16-byte aligned and
// the two-channel frames of int64 / float64 arrays (compress.c:482-511) -- used to take the slot sequence: K3 packs
// every frame into a worst-case slot (16.6 KB per channel and frame: 17 GB for the 4096 x 2^20 workload), K4 scans the
// sizes in two launches, a device-to-host copy sizes the blob, K5 re-reads every slot from HBM and moves it.
//
// Here the wave that packed a frame also places it.  The grid is persistent (kPlacedGrid workgroups: what the chip
// holds); every wave owns TWO slots and loops:
//     ticket  -> frame number g (a device-wide counter, so frames are handed out in start order)
//     K3's frame body (encode_frame_body, unchanged: same decisions, same bits) packs frame g into one slot
//     size_pub[g] <- bytes; the scanner wave (ticket 0, K3F's fused_scanner) turns published sizes into byte offsets
//     the frame packed ONE TRIP AGO, waiting in the other slot: off_pub[g'] -> its absolute offset (it has had a whole
//                    frame body's time to arrive: the scanner's latency and the spread of the frames before it are
//                    hidden), and the wave moves it there (K5's per-frame copy: byte shift + CRC-16).
// Progress: let u be the smallest frame not yet published; the wave that holds its ticket is either packing it (and will
// publish without waiting for anybody) or -- the ticket is drawn one step early -- waiting for the offset of the frame it
// packed before, which is smaller than u, so everything before THAT is published and the scanner delivers.  No dependence on dispatch order; time-outs raise the error word and every wave terminates.
// The wave that places a frame also writes the stream index and header fields that frame owns (its seek point; the
// stream's fixed header and starts[] with the first frame, nbytes[] with the last): the whole encode is ONE launch after
// the memset of the publish words, and the host waits once, for the error word and the total.
// Slots: 2 x grid x 16.6 KB (x 2 for two channels) = 68-136 MB instead of one per frame; no K4, no K5, no size
// read-back before the blob is written.  The bytes are those of the slot sequence (tests pin one to the other).
#pragma once
#include "encode_fused.hpp"

namespace fa {

// The persistent grid: 8 workgroups (20 KB of LDS each) on each of 256 CUs.  Not asked of the occupancy API, which answers
// 3-4 per CU for this kernel on this stack (it reckons with 64 KB of LDS) -- a grid that size ran the frame bodies at half
// the chip.  A grid larger than what is resident is harmless: a workgroup draws its first ticket when it starts, so the
// ones that wait for a free CU hold no frame anybody could be waiting for.
constexpr int kPlacedGrid = 2048;
#ifndef FA_PG_GROUP
#define FA_PG_GROUP kPlacedGroup
#endif
constexpr int kPlacedGroup = 16;  // 256-byte blocks the placement copy keeps in flight (the frame body's registers are free by then)

// byte i (< 46) of a stream's fixed header: "fLaC", STREAMINFO (RFC 9639 8.2: block size B twice, frame sizes unknown,
// 44100 Hz, nch channels of 32 bits, stream_size samples, no MD5), and the SEEKTABLE block header of nf 18-byte points --
// the bytes write_headers_kernel / fused_finish_kernel write
__device__ __forceinline__ uint8_t stream_header_byte(int i, int B, int nch, int64_t stream_size, int64_t nf) {
    const uint64_t ts = ((uint64_t)stream_size < (1ULL << 36)) ? (uint64_t)stream_size : 0;
    const uint64_t packed = ((uint64_t)44100 << 44) | ((uint64_t)(nch - 1) << 41) | ((uint64_t)31 << 36) | ts;
    const uint32_t stl = (uint32_t)(18 * nf);
    if (i < 4) return (uint8_t)(0x43614C66u >> (8 * i));  // "fLaC"
    if (i < 8) return (i == 7) ? 34 : 0;
    const int j = i - 8;  // STREAMINFO byte
    if (j < 4) return (uint8_t)((j & 1) ? B : (B >> 8));
    if (j < 10) return 0;
    if (j < 18) return (uint8_t)(packed >> (56 - 8 * (j - 10)));
    if (j < 34) return 0;
    const int k = i - 42;  // SEEKTABLE block header (last metadata block)
    return (k == 0) ? 0x83 : (uint8_t)(stl >> (8 * (3 - k)));
}

#if defined(FA_UNIT_PLACED) || !defined(FA_SPLIT_UNITS)
// p: the placement half of K3F's argument block (total_frames, nframes, n_stream, hb, blob, capacity, ticket, size_pub,
// off_pub, total, err, info, starts, nbytes); p.crc_tab holds K5's tables here (kCrcTabWords entries).
// Register budget: the two-channel kernels need the two-waves-per-SIMD bound spelled out (left alone they take more than
// 256 registers: one wave per SIMD, 23 instead of 16 ms on 1024 x 2^20 int64); the one-channel kernels stay at ~182 on
// their own and run 5 % faster without it (8.45 against 8.85 ms on 1024 x (2^20 - 3) int32) -- except the fixed-predictor
// kernels of levels 0-2, which like the bound (13.2 against 13.7 ms on 1024 x 2^20 at level 1).
#ifndef FA_PG_ATTR
#define FA_PG_ATTR __attribute__((amdgpu_waves_per_eu((NCH == 2 || MLO == 0) ? 2 : 1, (NCH == 2 || MLO == 0) ? 2 : 8)))
#endif
template <int MLO, int NCH>
__global__ __launch_bounds__(64) FA_PG_ATTR void encode_placed_kernel(EncodeArgs a, FusedArgs p) {
    __shared__ __attribute__((aligned(16))) int32_t lds[k3_lds_words<NCH>()];
    static_assert(kCrcTabWords / 2 <= kSmpWords, "the CRC tables overlay the frame image");
    const int lane = threadIdx.x;
    // two slots per workgroup: the frame packed last waits in one for its offset while the next is packed into the other
    uint8_t* const slot0 = a.slots + (size_t)blockIdx.x * 2 * (size_t)a.slot_stride;
    // K5's tables take the frame image's place once the frame is packed -- except in the kernels of levels 0-2, whose
    // 1152-sample image (staged in 5 rows of 256: 21 chunks of 68 words) ends at word 1428 of the 4420: there the tables
    // sit behind it for the whole launch (12.45 -> 11.92 ms on 1024 x 2^20 at level 1, A/B in one call; the same change
    // measured while the loop still waited for offsets in line looked like a loss).
#ifndef FA_PG_TABRES
#define FA_PG_TABRES 1
#endif
    constexpr bool kTabResident = FA_PG_TABRES && (MLO == 0);
    static_assert(!kTabResident || (1536 + kCrcTabWords / 2 <= kSmpWords && 1536 >= ((1152 + kRow - 1) / kRow * kRow / kChunk + 1) * kChunkStride), "the resident tables lie behind the short image");
    uint16_t* const tab = reinterpret_cast<uint16_t*>(lds + (kTabResident ? 1536 : 0));
    if (kTabResident) {
        for (int i = lane; i < kCrcTabWords / 2; i += 64) reinterpret_cast<uint32_t*>(tab)[i] = reinterpret_cast<const uint32_t*>(p.crc_tab)[i];
    }
#ifdef FA_STAMPS  // diagnostic build: cycles per phase of the loop, summed over the frames of every 64th workgroup (stamps[20..25])
    unsigned long long pg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_ = fa_memtime();
    if (lane == 0 && p.stamps) {  // when did this workgroup start?  (100 MHz ticks after the first one: histogram of 2.5 ms bins in stamps[32..39])
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();
        const unsigned long long old_ = atomicCAS(&p.stamps[28], 0ULL, now_);
        const unsigned long long rel_ = old_ ? now_ - old_ : 0ULL;
        const int b_ = (int)(rel_ / 250000ULL);
        atomicAdd(&p.stamps[32 + (b_ > 7 ? 7 : b_)], 1ULL);
    }
#define FA_PG_STAMP(k) do { const unsigned long long n_ = fa_memtime(); pg_[k] += n_ - pt_; pt_ = n_; } while (0)
#define FA_PG_FLUSH do { if (lane == 0 && p.stamps) { if ((blockIdx.x & 63) == 1) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&p.stamps[20 + i_], pg_[i_]); if (pg_[5]) atomicAdd(&p.stamps[30], 1ULL); atomicMax(&p.stamps[31], pg_[5]); } } while (0)
#else
#define FA_PG_STAMP(k) do { } while (0)
#define FA_PG_FLUSH do { } while (0)
#endif
    int64_t g_wait = -1;  // the frame that waits in the other slot (published, not placed yet)
    uint32_t n_wait = 0;
    int cur = 0;
    // One ticket = one frame.  (Four consecutive frames per ticket -- a quarter of the atomics on the one counter -- ran
    // 400 x slower: a wave waits for an offset between two frames of its ticket, so publishing frame 4k+3 came to depend
    // on frame 4k-1 being published, a serial chain through every ticket.)  The next ticket is drawn inside the frame body
    // (below) and looked at after the placement step, so the counter's latency is hidden.
    // (the kernels of levels 0-2 only: 13.2 -> 12.3 ms on 1024 x 2^20 at level 1, where a frame body is 20 us and the draw's
    // latency a quarter of it; 4096-sample frames lose 2-3 % by it, 8.47 -> 8.63-8.72 ms on 1024 x (2^20 - 3))
    const bool early_draw = (MLO == 0) && p.total_frames > 2 * (int64_t)kPlacedGrid;
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t == 0) {  // the first wave to start is the scanner; when it is done every frame has its offset
            fused_scanner(p, lane);
            return;
        }
        const int64_t g = (int64_t)t - 1;
        const bool have = g < p.total_frames;
        FA_PG_STAMP(0);  // ticket
        uint32_t n = 0;
        if (have) {
            // (the lane number is recomputed per frame behind an asm the optimiser cannot see through: taken from threadIdx.x,
            // every lane-dependent address and mask of the frame body is invariant in this loop, gets hoisted out of it and
            // lives in registers across the whole body -- 257 instead of 169 registers, i.e. one wave per SIMD instead of two)
            const int lane_f = lane_id_opaque();
            __builtin_assume(lane_f >= 0 && lane_f < 64);
            // The next trip's ticket is drawn in the middle of the frame body, right after the staging loads: vector memory
            // operations return in order, so drawn anywhere near other loads (the tables, the slot) its microseconds on the
            // one counter every wave of the chip draws from stall whoever waits next; behind it here lie the fixed-predictor
            // passes, LDS and registers only.
            // (Arrays of a frame or two per wave draw it after the body instead: their waves run in step, a second burst of
            // draws in the middle of the body stalls every one of them: 0.141 against 0.113 ms at 1024 frames.)
            n = encode_frame_body<MLO, NCH>(a, g, slot0 + (size_t)cur * (size_t)a.slot_stride, lds, lane_f, [&]() __attribute__((always_inline)) {
                if (early_draw && lane == 0) t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
            // a frame is published as soon as it is packed, before this wave waits for anything: whoever holds a ticket
            // publishes without depending on anybody, so every wait below ends
            if (lane == 0) __hip_atomic_store(p.size_pub + g, 0x80000000u | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FA_PG_STAMP(1);  // frame body
        if (!early_draw && have && lane == 0) t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g_wait >= 0) {
            // ---- place the frame packed one trip ago: its offset has had a whole frame body's time to arrive ----
            const uint32_t s = (uint32_t)g_wait / (uint32_t)a.nframes;
            const uint32_t f = (uint32_t)g_wait - s * (uint32_t)a.nframes;
            // the two offsets this step needs -- the frame's and that of its stream's first frame -- are asked for before the
            // tables are fetched: one round trip to memory for all three instead of three in a row
            unsigned long long off = lb_load(p.off_pub + g_wait);
            unsigned long long first_abs = lb_load(p.off_pub + (size_t)s * (size_t)a.nframes);
            lds_fence();  // (the writer's last LDS reads are done: the image may go)
            if (!kTabResident) {
                for (int i = lane; i < kCrcTabWords / 2; i += 64) reinterpret_cast<uint32_t*>(tab)[i] = reinterpret_cast<const uint32_t*>(p.crc_tab)[i];
            }
            FA_PG_STAMP(2);  // tables
            for (uint32_t spins = 0;; ++spins) {
                if (spins) off = lb_load(p.off_pub + g_wait);
                off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
                if (off) break;
                if (spins > kLbSpinLimit) break;
                if ((spins & 1023u) == 1023u && (lb_load(p.err) & 6)) return;  // somebody timed out: nothing behind it will get an offset
                __builtin_amdgcn_s_sleep(FA_F_SLEEP);
            }
            FA_PG_STAMP(3);  // wait for the offset
            if (!off) {
                if (lane == 0) atomicOr(p.err, 2);
                return;
            }
            if ((int64_t)off + (int64_t)n_wait > p.capacity) {  // the caller's buffer is too small: sized, not written
                if (lane == 0) atomicOr(p.err, 1);
            } else {
                // The slot's stores before its loads: both are this wave's and go through this CU's write-through L1, which
                // keeps a CU coherent with itself, so ordering is all it takes (workgroup scope = s_waitcnt).  Agent scope
                // here would write back and invalidate the whole XCD's L2 once per frame and wave (measured: 2x the run time).
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                compact_one_frame<FA_PG_GROUP>(lane, reinterpret_cast<const uint32_t*>(slot0 + (size_t)(cur ^ 1) * (size_t)a.slot_stride), n_wait,
                                                p.blob + off, tab);
                FA_PG_STAMP(4);  // placement: the copy
                // ---- the stream's index entries and header fields this frame owns (what K5a / the finish kernel of K3F
                //      write in a launch of their own): its seek point; with the stream's first frame the 46 fixed bytes
                //      and starts[s]; with its last frame nbytes[s] ----
                // (the first frame's offset is out, or about to be: the scanner passes the frames in order, but two of its
                // stores may land in either order)
                if (f == 0) first_abs = off;
                for (uint32_t spins = 0; first_abs == 0 && spins < kLbSpinLimit; ++spins) first_abs = lb_load(p.off_pub + (size_t)s * (size_t)a.nframes);
                first_abs = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(first_abs >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first_abs);
                if (first_abs == 0) {
                    if (lane == 0) atomicOr(p.err, 2);
                    return;
                }
                uint8_t* const h = p.blob + (first_abs - (unsigned long long)p.hb);
                if (lane < 18) {
                    const uint64_t sn = (uint64_t)f * (uint64_t)a.B, fo = off - first_abs;
                    const uint32_t bsz = (f == (uint32_t)a.nframes - 1) ? (uint32_t)a.tail_bs : (uint32_t)a.B;
                    h[46 + 18 * (size_t)f + lane] = (lane < 8) ? (uint8_t)(sn >> (56 - 8 * lane))
                                                  : (lane < 16) ? (uint8_t)(fo >> (56 - 8 * (lane - 8)))
                                                                : (uint8_t)(bsz >> (8 * (17 - lane)));
                }
                if (f == 0) {
                    if (lane < 46) h[lane] = stream_header_byte(lane, a.B, NCH, a.stream_size, a.nframes);
                    if (lane == 0) p.starts[s] = (int64_t)(first_abs - (unsigned long long)p.hb);
                }
                if (f == (uint32_t)a.nframes - 1 && lane == 0) p.nbytes[s] = (int64_t)(off + n_wait - (first_abs - (unsigned long long)p.hb));
            }
            lds_fence();  // (the table reads are done before the next frame's image overwrites them)
            FA_PG_STAMP(6);  // placement: seek point, header, index
#ifdef FA_STAMPS
            pg_[5] += 1;
#endif
        }
        if (!have) { FA_PG_FLUSH; return; }
        g_wait = g;
        n_wait = n;
        cur ^= 1;
    }
}
#endif

void launch_encode_placed(hipStream_t st, const EncodeArgs& a, const FusedArgs& p, int nch, int64_t grid)
#if defined(FA_UNIT_PLACED) || !defined(FA_SPLIT_UNITS)
{
    const dim3 g((unsigned)grid), b(64);
#ifdef FA_DEV_MINIMAL
    hipLaunchKernelGGL((encode_placed_kernel<8, 1>), g, b, 0, st, a, p);
#else
    if (nch == 1) {
        switch (a.max_lpc_order) {
            case 0: hipLaunchKernelGGL((encode_placed_kernel<0, 1>), g, b, 0, st, a, p); break;
            case 6: hipLaunchKernelGGL((encode_placed_kernel<6, 1>), g, b, 0, st, a, p); break;
            case 8: hipLaunchKernelGGL((encode_placed_kernel<8, 1>), g, b, 0, st, a, p); break;
            default: hipLaunchKernelGGL((encode_placed_kernel<12, 1>), g, b, 0, st, a, p); break;
        }
    } else {
        switch (a.max_lpc_order) {
            case 0: hipLaunchKernelGGL((encode_placed_kernel<0, 2>), g, b, 0, st, a, p); break;
            case 6: hipLaunchKernelGGL((encode_placed_kernel<6, 2>), g, b, 0, st, a, p); break;
            case 8: hipLaunchKernelGGL((encode_placed_kernel<8, 2>), g, b, 0, st, a, p); break;
            default: hipLaunchKernelGGL((encode_placed_kernel<12, 2>), g, b, 0, st, a, p); break;
        }
    }
#endif
}
#else
;
#endif

}  // namespace fa
