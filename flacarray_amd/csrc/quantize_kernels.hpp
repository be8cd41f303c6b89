// quantize_kernels.hpp -- float32 <-> int32 quantisation on gfx950.
//
// K1 float32_to_int32_kernel : float32_to_int32(), src/flacarray/libflacarray/utils.c:160-243
// K2 int32_to_float32_kernel : int32_to_float32(), src/flacarray/libflacarray/utils.c:350-368
//
// The arithmetic follows the reference operation by operation: float subtract and float
// multiply, then a DOUBLE add of +-0.5 and truncation (utils.c:232-240); offset snapped to a
// whole number of quanta in double (utils.c:221-222).  No contraction: every float op is an
// explicit __f*_rn intrinsic.  Out-of-range casts reproduce what the reference does on x86-64
// (cvttsd2si returns INT_MIN).  A NaN anywhere in the input sets bit 0 of *flags, which the
// Python layer turns into the reference's RuntimeError (utils.py:268-269).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

__device__ __forceinline__ int32_t x86_cvtt_i32(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}
__device__ __forceinline__ int64_t x86_cvtt_i64(double v) {
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)v;
}

// One 1024-thread block per stream: pass 1 min/max (HBM read), pass 2 convert (re-read is
// served by L2 / Infinity Cache for streams up to tens of MB).
FA_GLOBAL __global__ __launch_bounds__(1024) void float32_to_int32_kernel(const float* __restrict__ input, int64_t stream_size,
                                                                const float* __restrict__ quanta, int32_t* __restrict__ output,
                                                                float* __restrict__ offsets, float* __restrict__ gains,
                                                                int* __restrict__ flags) {
    __shared__ float s_min[16], s_max[16];
    __shared__ float s_off, s_gain;
    __shared__ int s_nan;
    const int64_t is = blockIdx.x;
    const float* in = input + is * stream_size;
    int32_t* out = output + is * stream_size;
    const int tid = threadIdx.x;
    if (tid == 0) s_nan = 0;
    float mn = in[0], mx = in[0];
    bool nan = false;
    const bool vec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const int64_t n4 = vec ? (stream_size >> 2) : 0;
    for (int64_t i = tid; i < n4; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        nan = nan || (v.x != v.x) || (v.y != v.y) || (v.z != v.z) || (v.w != v.w);
        mn = (v.x < mn) ? v.x : mn; mx = (v.x > mx) ? v.x : mx;
        mn = (v.y < mn) ? v.y : mn; mx = (v.y > mx) ? v.y : mx;
        mn = (v.z < mn) ? v.z : mn; mx = (v.z > mx) ? v.z : mx;
        mn = (v.w < mn) ? v.w : mn; mx = (v.w > mx) ? v.w : mx;
    }
    for (int64_t i = 4 * n4 + tid; i < stream_size; i += 1024) {
        const float v = in[i];
        nan = nan || (v != v);
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = (a < mn) ? a : mn;
        mx = (b > mx) ? b : mx;
    }
    __syncthreads();
    if (nan) s_nan = 1;
    if ((tid & 63) == 0) { s_min[tid >> 6] = mn; s_max[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        float smin = s_min[0], smax = s_max[0];
        for (int w = 1; w < 16; ++w) {
            smin = (s_min[w] < smin) ? s_min[w] : smin;
            smax = (s_max[w] > smax) ? s_max[w] : smax;
        }
        if (s_nan) atomicOr(flags, 1);
        float off = (float)(0.5 * (double)__fadd_rn(smin, smax));                    // utils.c:194
        const float d1 = __fsub_rn(smin, off), d2 = __fsub_rn(smax, off);
        const float amp = (d1 > d2) ? (float)(1.01 * (double)d1) : (float)(1.01 * (double)d2);  // :198-202
        const float min_quanta = __fdiv_rn(amp, 2147483648.0f);                      // :203 (float)2147483647 == 2^31
        const float sq = quanta ? quanta[is] : min_quanta;
        const int64_t nquant = x86_cvtt_i64((double)off / (double)sq);                // :221
        off = (float)((double)sq * (double)nquant);                                   // :222
        const float gain = (sq == 0.0f) ? 1.0f : (float)(1.0 / (double)sq);           // :224-230
        offsets[is] = off;
        gains[is] = gain;
        s_off = off;
        s_gain = gain;
    }
    __syncthreads();
    const float off = s_off, gain = s_gain;
    const bool ovec = vec && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int64_t m4 = ovec ? (stream_size >> 2) : 0;
    for (int64_t i = tid; i < m4; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        const float vs[4] = {v.x, v.y, v.z, v.w};
        int r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float st = __fsub_rn(vs[e], off);
            const float pr = __fmul_rn(gain, st);
            const double dv = (st >= 0.0f) ? (double)pr + 0.5 : (double)pr - 0.5;    // :236-239
            r[e] = x86_cvtt_i32(dv);
        }
        reinterpret_cast<int4*>(out)[i] = make_int4(r[0], r[1], r[2], r[3]);
    }
    for (int64_t i = 4 * m4 + tid; i < stream_size; i += 1024) {
        const float st = __fsub_rn(in[i], off);
        const float pr = __fmul_rn(gain, st);
        const double dv = (st >= 0.0f) ? (double)pr + 0.5 : (double)pr - 0.5;
        out[i] = x86_cvtt_i32(dv);
    }
}

// ---- float32 input fused into the encoder (encode_fused.hpp, F32IN): only the per-stream range is needed up front ----
// K1a: grid = n_stream x chunks, one 256-thread block per chunk of up to kRangeChunk samples: min / max / NaN of the chunk.
// K1b: one thread per stream folds the chunk results into offset and gain, with exactly the arithmetic of K1 / utils.c:194-230.
// The samples are then quantised where the encoder loads them (quantise_f32): 4 B per sample read once more, nothing written.
constexpr int kRangeChunk = 65536;
FA_GLOBAL __global__ __launch_bounds__(256) void float32_range_kernel(const float* __restrict__ input, int64_t stream_size, int64_t chunks_per_stream,
                                                            float* __restrict__ part_min, float* __restrict__ part_max,
                                                            int* __restrict__ flags) {
    __shared__ float s_min[4], s_max[4];
    const int64_t is = blockIdx.x / chunks_per_stream;
    const int64_t ck = blockIdx.x - is * chunks_per_stream;
    const int64_t lo = ck * kRangeChunk;
    int64_t hi = lo + kRangeChunk;
    if (hi > stream_size) hi = stream_size;
    const float* in = input + is * stream_size + lo;
    const int tid = threadIdx.x;
    float mn = in[0], mx = in[0];
    bool nan = false;
    const bool vec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const int64_t n = hi - lo;
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t i = tid; i < n4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        nan = nan || (v.x != v.x) || (v.y != v.y) || (v.z != v.z) || (v.w != v.w);
        mn = (v.x < mn) ? v.x : mn; mx = (v.x > mx) ? v.x : mx;
        mn = (v.y < mn) ? v.y : mn; mx = (v.y > mx) ? v.y : mx;
        mn = (v.z < mn) ? v.z : mn; mx = (v.z > mx) ? v.z : mx;
        mn = (v.w < mn) ? v.w : mn; mx = (v.w > mx) ? v.w : mx;
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += 256) {
        const float v = in[i];
        nan = nan || (v != v);
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = (a < mn) ? a : mn;
        mx = (b > mx) ? b : mx;
    }
    if (__any(nan) && (tid & 63) == 0) atomicOr(flags, 1);
    if ((tid & 63) == 0) { s_min[tid >> 6] = mn; s_max[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = (s_min[w] < mn) ? s_min[w] : mn;
            mx = (s_max[w] > mx) ? s_max[w] : mx;
        }
        part_min[blockIdx.x] = mn;
        part_max[blockIdx.x] = mx;
    }
}

FA_GLOBAL __global__ __launch_bounds__(256) void float32_params_kernel(const float* __restrict__ part_min, const float* __restrict__ part_max,
                                                             int64_t n_stream, int64_t chunks_per_stream, const float* __restrict__ quanta,
                                                             float* __restrict__ offsets, float* __restrict__ gains) {
    const int64_t is = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (is >= n_stream) return;
    float smin = part_min[is * chunks_per_stream], smax = part_max[is * chunks_per_stream];
    for (int64_t c = 1; c < chunks_per_stream; ++c) {
        const float a = part_min[is * chunks_per_stream + c], b = part_max[is * chunks_per_stream + c];
        smin = (a < smin) ? a : smin;
        smax = (b > smax) ? b : smax;
    }
    float off = (float)(0.5 * (double)__fadd_rn(smin, smax));                    // utils.c:194
    const float d1 = __fsub_rn(smin, off), d2 = __fsub_rn(smax, off);
    const float amp = (d1 > d2) ? (float)(1.01 * (double)d1) : (float)(1.01 * (double)d2);  // :198-202
    const float min_quanta = __fdiv_rn(amp, 2147483648.0f);                      // :203
    const float sq = quanta ? quanta[is] : min_quanta;
    const int64_t nquant = x86_cvtt_i64((double)off / (double)sq);                // :221
    off = (float)((double)sq * (double)nquant);                                   // :222
    offsets[is] = off;
    gains[is] = (sq == 0.0f) ? 1.0f : (float)(1.0 / (double)sq);                  // :224-230
}

// one sample, utils.c:232-240: float subtract, float multiply, double +-0.5, truncation (x86 overflow behaviour)
__device__ __forceinline__ int32_t quantise_f32(float x, float off, float gain) {
    const float st = __fsub_rn(x, off);
    const float pr = __fmul_rn(gain, st);
    const double dv = (st >= 0.0f) ? (double)pr + 0.5 : (double)pr - 0.5;
    return x86_cvtt_i32(dv);
}

// grid = (chunks per stream, n_stream folded into x); each block converts up to 16384 samples
constexpr int kDequantChunk = 16384;
FA_GLOBAL __global__ __launch_bounds__(256) void int32_to_float32_kernel(const int32_t* __restrict__ input, int64_t stream_size,
                                                               int64_t chunks_per_stream, const float* __restrict__ offsets,
                                                               const float* __restrict__ gains, float* __restrict__ output) {
    const int64_t is = blockIdx.x / chunks_per_stream;
    const int64_t ck = blockIdx.x - is * chunks_per_stream;
    const float off = offsets[is];
    const float coeff = (float)(1.0 / (double)gains[is]);                             // utils.c:361
    const int64_t lo = ck * kDequantChunk;
    int64_t hi = lo + kDequantChunk;
    if (hi > stream_size) hi = stream_size;
    const int32_t* in = input + is * stream_size;
    float* out = output + is * stream_size;
    const bool vec = ((reinterpret_cast<uintptr_t>(in + lo) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out + lo) & 15) == 0);
    const int64_t n4 = vec ? ((hi - lo) >> 2) : 0;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
        const int4 v = reinterpret_cast<const int4*>(in + lo)[i];
        float4 o;
        o.x = __fadd_rn(off, __fmul_rn(coeff, (float)v.x));                           // utils.c:364
        o.y = __fadd_rn(off, __fmul_rn(coeff, (float)v.y));
        o.z = __fadd_rn(off, __fmul_rn(coeff, (float)v.z));
        o.w = __fadd_rn(off, __fmul_rn(coeff, (float)v.w));
        reinterpret_cast<float4*>(out + lo)[i] = o;
    }
    for (int64_t i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256) out[i] = __fadd_rn(off, __fmul_rn(coeff, (float)in[i]));
}

// ---- float64 <-> int64 (utils.c:245-348): every operation is a double operation ------------------
FA_GLOBAL __global__ __launch_bounds__(1024) void float64_to_int64_kernel(const double* __restrict__ input, int64_t stream_size,
                                                                const double* __restrict__ quanta, int64_t* __restrict__ output,
                                                                double* __restrict__ offsets, double* __restrict__ gains,
                                                                int* __restrict__ flags) {
    __shared__ double s_min[16], s_max[16];
    __shared__ double s_off, s_gain;
    __shared__ int s_nan;
    const int64_t is = blockIdx.x;
    const double* in = input + is * stream_size;
    int64_t* out = output + is * stream_size;
    const int tid = threadIdx.x;
    if (tid == 0) s_nan = 0;
    double mn = in[0], mx = in[0];
    bool nan = false;
    const bool vec = ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const int64_t n2 = vec ? (stream_size >> 1) : 0;
    for (int64_t i = tid; i < n2; i += 1024) {
        const double2 v = reinterpret_cast<const double2*>(in)[i];
        nan = nan || (v.x != v.x) || (v.y != v.y);
        mn = (v.x < mn) ? v.x : mn; mx = (v.x > mx) ? v.x : mx;
        mn = (v.y < mn) ? v.y : mn; mx = (v.y > mx) ? v.y : mx;
    }
    for (int64_t i = 2 * n2 + tid; i < stream_size; i += 1024) {
        const double v = in[i];
        nan = nan || (v != v);
        mn = (v < mn) ? v : mn;
        mx = (v > mx) ? v : mx;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = (a < mn) ? a : mn;
        mx = (b > mx) ? b : mx;
    }
    __syncthreads();
    if (nan) s_nan = 1;
    if ((tid & 63) == 0) { s_min[tid >> 6] = mn; s_max[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        double smin = s_min[0], smax = s_max[0];
        for (int w = 1; w < 16; ++w) {
            smin = (s_min[w] < smin) ? s_min[w] : smin;
            smax = (s_max[w] > smax) ? s_max[w] : smax;
        }
        if (s_nan) atomicOr(flags, 1);
        double off = 0.5 * (smin + smax);                                             // utils.c:278
        const double amp = ((smin - off) > (smax - off)) ? 1.01 * (smin - off) : 1.01 * (smax - off);  // :282-286
        const double min_quanta = amp / 9223372036854775808.0;                        // :287 (double)(2^63 - 1) == 2^63
        const double sq = quanta ? quanta[is] : min_quanta;
        const int64_t nquant = x86_cvtt_i64(off / sq);                                // :305
        off = sq * (double)nquant;                                                    // :306
        const double gain = (sq == 0.0) ? 1.0 : 1.0 / sq;                             // :308-314
        offsets[is] = off;
        gains[is] = gain;
        s_off = off;
        s_gain = gain;
    }
    __syncthreads();
    const double off = s_off, gain = s_gain;
    for (int64_t i = tid; i < stream_size; i += 1024) {
        const double t = in[i] - off;
        out[i] = (t >= 0.0) ? x86_cvtt_i64(gain * t + 0.5) : x86_cvtt_i64(gain * t - 0.5);  // :316-323
    }
}

FA_GLOBAL __global__ __launch_bounds__(256) void int64_to_float64_kernel(const int64_t* __restrict__ input, int64_t stream_size,
                                                               int64_t chunks_per_stream, const double* __restrict__ offsets,
                                                               const double* __restrict__ gains, double* __restrict__ output) {
    const int64_t is = blockIdx.x / chunks_per_stream;
    const int64_t ck = blockIdx.x - is * chunks_per_stream;
    const double off = offsets[is];
    const double coeff = 1.0 / gains[is];                                             // utils.c:340
    const int64_t lo = ck * kDequantChunk;
    int64_t hi = lo + kDequantChunk;
    if (hi > stream_size) hi = stream_size;
    const int64_t* in = input + is * stream_size;
    double* out = output + is * stream_size;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) out[i] = off + coeff * (double)in[i];  // :343
}

}  // namespace fa
