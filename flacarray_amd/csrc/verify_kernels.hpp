// verify_kernels.hpp -- optional integrity pass of the decoder: the CRC-16 of every decoded frame.
//
// libFLAC verifies the frame CRC-16 while decoding and reports a mismatch through its error callback
// (FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH), which the reference prints
// (src/flacarray/libflacarray/decompress.c:104-121).  K7 checks every header's CRC-8 but not the frame CRC-16 -- a
// flipped residual bit decodes to wrong samples with return code 0.  With fa_set_decode_verify(1) this kernel runs
// after K7 over the same tasks and turns a mismatch into ERROR_DECODE_PROCESS.  Off by default: it re-reads the
// compressed bytes (c per sample).
//
// One wavefront per frame; the lanes fold interleaved 32-bit words with the slicing tables of the single-pass
// encoder (encode_fused.hpp: the running state enters through the top half of the next word, 256 bytes further on).
#pragma once
#include "decode_kernels.hpp"
#include "encode_fused.hpp"

namespace fa {

__global__ __launch_bounds__(256) void verify_crc16_kernel(DecodeArgs a, const uint16_t* __restrict__ crc_tab) {
    __shared__ __attribute__((aligned(16))) uint16_t crc_s[kFCrcSlice];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kFCrcSlice / 2; i += 256) reinterpret_cast<uint32_t*>(crc_s)[i] = reinterpret_cast<const uint32_t*>(crc_tab)[i];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 4 + (tid >> 6);
    if (t >= a.n_tasks) return;
    int64_t s, f;
    if (a.task_stream) { s = a.task_stream[t]; f = a.task_frame[t]; }
    else { s = t / a.nfr; f = a.f0 + (t - s * a.nfr); }
    const int64_t start = a.ftab[s * a.nf + f];
    if (start < 0) return;  // (K6 / K7 have flagged the stream already)
    const int64_t end = (f + 1 < a.nf) ? a.ftab[s * a.nf + f + 1] : a.meta[s].end_abs;
    if (end - start < 4 || end > a.blob_bytes) {
        if (lane == 0) atomicOr(a.err, kErrDecodeProcess);
        return;
    }
    const uint8_t* p = a.blob + start;
    const uint32_t L = (uint32_t)(end - start - 2);  // bytes covered by the CRC
    uint32_t crc_t = 0, last_end = 0;
    bool any = false;
    // Eight blocks of 256 bytes per trip: the eight loads of a lane are in flight together and the (serial, LDS-bound)
    // folds follow -- one load per trip made the kernel wait for memory 37 times per frame (3.8 ms at cfg 2; K7 itself
    // takes 7).  A trip whose 2048 bytes lie inside the frame's CRC range and inside the blob (wave-uniform test) runs
    // without per-word masks or address arithmetic (one address, immediate offsets: the kernel must stay within 48 registers,
    // what a SIMD has left beside two waves of K7); the frame's last bytes go word by word.
#ifndef FA_K9_BATCH
#define FA_K9_BATCH 8
#endif
    constexpr int kBatch = FA_K9_BATCH;
    const uint32_t Lin = (uint32_t)(((int64_t)L < a.blob_bytes - start ? (int64_t)L : a.blob_bytes - start) & ~(int64_t)3);  // whole words inside both
    uint32_t base = 0;
    for (; base + 256u * kBatch <= Lin; base += 256u * kBatch) {
        const uint8_t* q = p + base + 4u * (uint32_t)lane;
        uint32_t raw[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) __builtin_memcpy(&raw[k], q + 256 * k, 4);  // (any byte alignment)
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint32_t w = __builtin_bswap32(raw[k]) ^ (crc_t << 16);
            crc_t = (uint32_t)crc_s[w >> 24] ^ (uint32_t)crc_s[256 + ((w >> 16) & 255u)] ^ (uint32_t)crc_s[512 + ((w >> 8) & 255u)] ^
                    (uint32_t)crc_s[768 + (w & 255u)];
        }
        last_end = base + 256u * (kBatch - 1) + 4u * (uint32_t)lane + 4u;
        any = true;
    }
    for (uint32_t o = base + 4u * (uint32_t)lane; o < L; o += 256u) {
        uint32_t w;
        if (start + o + 4 <= a.blob_bytes) {
            uint32_t raw;
            __builtin_memcpy(&raw, p + o, 4);
            w = __builtin_bswap32(raw);
        } else {  // the last bytes of the blob: byte by byte
            w = 0;
            for (uint32_t b2 = 0; b2 < 4 && start + o + b2 < a.blob_bytes; ++b2) w |= (uint32_t)p[o + b2] << (24 - 8 * b2);
        }
        if (o + 4u > L) w &= ~0u << (8u * (4u - (L - o)));  // bytes at and after L (the CRC itself) do not count
        w ^= crc_t << 16;
        crc_t = (uint32_t)crc_s[w >> 24] ^ (uint32_t)crc_s[256 + ((w >> 16) & 255u)] ^ (uint32_t)crc_s[512 + ((w >> 8) & 255u)] ^
                (uint32_t)crc_s[768 + (w & 255u)];
        last_end = o + 4u;
        any = true;
    }
    uint32_t contrib = 0;
    if (any) contrib = crc16_mulmod((uint16_t)crc_t, crc_tab[kFCrcSlice + ((int)L - (int)last_end) + 3]);
    contrib ^= (uint32_t)xchg_i32<0>((int)contrib);
    contrib ^= (uint32_t)xchg_i32<1>((int)contrib);
    contrib ^= (uint32_t)xchg_i32<2>((int)contrib);
    contrib ^= (uint32_t)xchg_i32<3>((int)contrib);
    contrib ^= (uint32_t)xchg_i32<4>((int)contrib);
    const uint32_t crc = ((uint32_t)__builtin_amdgcn_readlane((int)contrib, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)contrib, 32)) & 0xFFFFu;
    const uint32_t stored = ((uint32_t)p[L] << 8) | (uint32_t)p[L + 1];
    if (lane == 0 && crc != stored) atomicOr(a.err, kErrDecodeProcess);
}

}  // namespace fa
