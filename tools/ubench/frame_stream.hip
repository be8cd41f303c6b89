// Micro-benchmark: the memory ceiling of the decoder's ACCESS SHAPE, without any decoding.  Diagnostic tool.
//
// K7 (decode_kernels.hpp) gives every lane one frame: the lane reads its frame's compressed bytes in RC-byte chunks
// (64 distinct cache lines per load instruction) and the wave stores the decoded samples as WP-byte row pieces (one
// piece per frame and tile).  This kernel does exactly those loads and stores -- same strides, same number of
// frames in flight (waves per CU through a dummy LDS array) -- and nothing else, so its time is what the memory
// system charges for the pattern.  `frame_in` bytes of input and 16384 bytes of output per frame.
//   hipcc -O3 --offload-arch=gfx950 -o frame_stream frame_stream.hip && ./frame_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// RC: bytes a lane reads per load burst (64, 128, 256); WP: bytes of a row piece (128, 256, 512); NT: nontemporal stores
template <int RC, int WP, int LDSB, bool NT, int MODE = 0>
__global__ __launch_bounds__(64) void frame_stream(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n_frames,
                                                    int frame_in, uint32_t* __restrict__ sink, int64_t ostride) {
    __shared__ uint32_t occupy[LDSB / 4];
    const int lane = threadIdx.x;
    const int64_t f = (int64_t)blockIdx.x * 64 + lane;
    if (f >= n_frames) return;
    occupy[lane] = lane;
    const uint8_t* src = in + f * (int64_t)frame_in;
    const int64_t frame0 = (int64_t)blockIdx.x * 64;
    uint32_t acc = occupy[(lane * 7) & 63];
    constexpr int SAMP_PER_TILE = WP / 4;           // samples of a row piece
    constexpr int LANES_PER_ROW = WP / 16;          // lanes that store one piece
    constexpr int ROWS_PER_PASS = 64 / LANES_PER_ROW;
    constexpr int PASSES = 64 / ROWS_PER_PASS;      // store instructions per tile
    int in_pos = 0;
    // the input is consumed at frame_in / 4096 bytes per sample
    for (int t = 0; t < 4096; t += SAMP_PER_TILE) {
        const int want = (int)(((int64_t)(t + SAMP_PER_TILE) * frame_in) >> 12);
        while (MODE != 2 && (in_pos + RC <= want || (t + SAMP_PER_TILE == 4096 && in_pos < frame_in))) {
#pragma unroll
            for (int v = 0; v < RC / 16; ++v) {
                if (in_pos + 16 * v + 16 <= frame_in) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(src + in_pos + 16 * v);
                    acc ^= d.x ^ d.y ^ d.z ^ d.w;
                }
            }
            in_pos += RC;
        }
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int r = it * ROWS_PER_PASS + lane / LANES_PER_ROW;
            const int cg = lane % LANES_PER_ROW;
            u32x4 v;
            v.x = acc; v.y = acc + 1; v.z = (uint32_t)t; v.w = (uint32_t)r;
            u32x4* p = reinterpret_cast<u32x4*>(out + (frame0 + r) * ostride + (int64_t)t * 4 + cg * 16);
            if (MODE != 1 && (frame0 + r) < n_frames) {
                if constexpr (NT) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        }
    }
    if (acc == 0x12345) sink[0] = acc;
}

template <int RC, int WP, int LDSB, bool NT, int MODE = 0>
void run(const char* name, const uint8_t* in, uint8_t* out, int64_t nf, int frame_in, uint32_t* sink, int64_t ostride = 16384) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const dim3 grid((unsigned)((nf + 63) / 64)), block(64);
    hipLaunchKernelGGL((frame_stream<RC, WP, LDSB, NT, MODE>), grid, block, 0, 0, in, out, nf, frame_in, sink, ostride);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((frame_stream<RC, WP, LDSB, NT, MODE>), grid, block, 0, 0, in, out, nf, frame_in, sink, ostride);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double bytes = (double)nf * ((MODE == 2 ? 0 : frame_in) + (MODE == 1 ? 0.0 : 16384.0));
    printf("%-58s %7.3f ms  %5.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    fflush(stdout);
}

// Cooperative reads: G consecutive lanes fetch 16 G contiguous bytes of one frame, one instruction serves 64 / G frames,
// G instructions serve the wave's 64 frames with 16 G bytes each (the data would go to the frames' LDS rings).
template <int G, int LDSB>
__global__ __launch_bounds__(64) void frame_read_coop(const uint8_t* __restrict__ in, int64_t n_frames, int frame_in, uint32_t* __restrict__ sink) {
    __shared__ uint32_t occupy[LDSB / 4];
    const int lane = threadIdx.x;
    occupy[lane] = lane;
    const int64_t frame0 = (int64_t)blockIdx.x * 64;
    uint32_t acc = occupy[(lane * 7) & 63];
    constexpr int RC = 16 * G;
    constexpr int SAMP = 32;
    int in_pos = 0;
    for (int t = 0; t < 4096; t += SAMP) {
        const int want = (int)(((int64_t)(t + SAMP) * frame_in) >> 12);
        while (in_pos + RC <= want || (t + SAMP == 4096 && in_pos < frame_in)) {
#pragma unroll
            for (int it = 0; it < G; ++it) {
                const int64_t f = frame0 + it * (64 / G) + lane / G;
                const int o = in_pos + 16 * (lane % G);
                if (f < n_frames && o + 16 <= frame_in) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(in + f * (int64_t)frame_in + o);
                    acc ^= d.x ^ d.y ^ d.z ^ d.w;
                }
            }
            in_pos += RC;
        }
    }
    if (acc == 0x12345) sink[0] = acc;
}
template <int G, int LDSB>
void run_coop(const char* name, const uint8_t* in, int64_t nf, int frame_in, uint32_t* sink) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const dim3 grid((unsigned)((nf + 63) / 64)), block(64);
    hipLaunchKernelGGL((frame_read_coop<G, LDSB>), grid, block, 0, 0, in, nf, frame_in, sink);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((frame_read_coop<G, LDSB>), grid, block, 0, 0, in, nf, frame_in, sink);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-58s %7.3f ms  %5.2f TB/s\n", name, best, (double)nf * frame_in / (best * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const int64_t nf = 2048 * 256;  // frames of 2048 channels x 2^20 samples
    const int frame_in = 9392;      // 2.293 bytes per sample, a multiple of 16
    uint8_t *in, *out;
    uint32_t* sink;
    if (hipMalloc(&in, nf * (int64_t)frame_in + 4096) != hipSuccess || hipMalloc(&out, nf * 20480) != hipSuccess ||
        hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 5, nf * (int64_t)frame_in);
    (void)hipMemset(out, 0, nf * 16384);
    printf("frames %lld, %d B in + 16384 B out per frame (%.2f GB)\n", (long long)nf, frame_in, nf * (frame_in + 16384.0) / 1e9);
    // LDS per workgroup sets the frames in flight: 17408 B -> 9 waves per CU (K7 today), 26112 -> 6, 13056 -> 12
    run<64, 128, 17408, false>("read 64 B, write 128 B pieces, 9 waves/CU (K7 today)", in, out, nf, frame_in, sink);
    run<64, 128, 17408, true>("read 64 B, write 128 B pieces, 9 waves/CU, nontemporal", in, out, nf, frame_in, sink);
    run<128, 128, 17408, false>("read 128 B, write 128 B, 9 waves/CU", in, out, nf, frame_in, sink);
    run<64, 256, 17408, false>("read 64 B, write 256 B, 9 waves/CU", in, out, nf, frame_in, sink);
    run<128, 256, 17408, false>("read 128 B, write 256 B, 9 waves/CU", in, out, nf, frame_in, sink);
    run<128, 512, 17408, false>("read 128 B, write 512 B, 9 waves/CU", in, out, nf, frame_in, sink);
    run<256, 512, 17408, false>("read 256 B, write 512 B, 9 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 26112, false>("read 64 B, write 128 B, 6 waves/CU", in, out, nf, frame_in, sink);
    run<128, 256, 26112, false>("read 128 B, write 256 B, 6 waves/CU", in, out, nf, frame_in, sink);
    run<128, 512, 26112, false>("read 128 B, write 512 B, 6 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 13056, false>("read 64 B, write 128 B, 12 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 8704, false>("read 64 B, write 128 B, 18 waves/CU", in, out, nf, frame_in, sink);
    run<128, 256, 40960, false>("read 128 B, write 256 B, 4 waves/CU", in, out, nf, frame_in, sink);
    run<128, 512, 40960, false>("read 128 B, write 512 B, 4 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 17408, false, 1>("reads only (64 B chunks), 9 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 17408, false, 2>("writes only (128 B pieces), 9 waves/CU", in, out, nf, frame_in, sink);
    run<64, 512, 17408, false, 2>("writes only (512 B pieces), 9 waves/CU", in, out, nf, frame_in, sink);
    run<64, 128, 17408, false, 2>("writes only (128 B), frame stride 16384+128", in, out, nf, frame_in, sink, 16384 + 128);
    run<64, 128, 17408, false, 2>("writes only (128 B), frame stride 16384+256", in, out, nf, frame_in, sink, 16384 + 256);
    run<64, 128, 17408, false, 2>("writes only (128 B), frame stride 16384+1024", in, out, nf, frame_in, sink, 16384 + 1024);
    run<64, 128, 17408, false, 2>("writes only (128 B), frame stride 16384+4096", in, out, nf, frame_in, sink, 16384 + 4096);
    run<64, 128, 17408, false, 0>("both, frame stride 16384+256", in, out, nf, frame_in, sink, 16384 + 256);
    run_coop<1, 17408>("coop reads: 1 lane x 16 B per frame and instruction", in, nf, frame_in, sink);
    run_coop<4, 17408>("coop reads: 4 lanes x 16 B = 64 B per frame", in, nf, frame_in, sink);
    run_coop<8, 17408>("coop reads: 8 lanes = 128 B per frame", in, nf, frame_in, sink);
    run_coop<16, 17408>("coop reads: 16 lanes = 256 B per frame", in, nf, frame_in, sink);
    run_coop<4, 26112>("coop reads: 64 B per frame, 6 waves/CU", in, nf, frame_in, sink);
    run_coop<8, 26112>("coop reads: 128 B per frame, 6 waves/CU", in, nf, frame_in, sink);
    return 0;
}
