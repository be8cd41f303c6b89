// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU / DS instructions the codec
// kernels are made of, at 1 and 2 waves per SIMD on gfx950.  Diagnostic tool, not part of the library.
//   hipcc -O2 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)



template <int OP>
__global__ void k(unsigned long long* out, int iters, int seed) {
    double a[8], b[8], c[8];
    for (int j = 0; j < 8; ++j) { a[j] = 1.0 + j * 1e-9 + seed; b[j] = 1.0 + 1e-12 * (j + threadIdx.x); c[j] = 1e-3 * (j + 1); }
    float f[8]; int i[8];
    for (int j = 0; j < 8; ++j) { f[j] = 0.5f + j; i[j] = j + threadIdx.x; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define OPX(j) \
            if constexpr (OP == 0) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[j]) : "v"(b[(j + 1) & 7])); } \
            else if constexpr (OP == 1) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b[j]), "v"(c[j])); } \
            else if constexpr (OP == 2) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b[0]), "v"(c[j])); } \
            else if constexpr (OP == 3) { asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[j]) : "v"(b[0]), "v"(c[j])); } \
            else if constexpr (OP == 4) { asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a[j]) : "v"(b[j])); } \
            else if constexpr (OP == 5) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[j]) : "v"(b[j]), "v"(c[j])); } \
            else if constexpr (OP == 6) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j]) : "v"(f[(j + 1) & 7]), "v"(f[(j + 2) & 7])); } \
            else if constexpr (OP == 7) { asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(i[j]) : "v"(i[(j + 1) & 7]), "v"(i[(j + 2) & 7])); } \
            else if constexpr (OP == 8) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b[0]), "s"(1.5)); } \
            ;
            REP8(OPX)
#undef OPX
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int j = 0; j < 8; ++j) s += a[j] + b[j] + c[j] + f[j] + i[j];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    if (s == 77.125) out[0] = 1;
}
template <int OP>
void run(const char* name) {
    unsigned long long* d;
    const int iters = 64;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps, blocks = 256;
        const int nw = threads / 64 * blocks;
        hipMalloc(&d, nw * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, r);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[nw / 2] / (iters * 32.0);
        if (wps == 1) printf("%-28s", name);
        printf("  %dw/SIMD: %6.2f cyc/instr/wave (SIMD %6.2f)", wps, cyc, cyc / wps);
        hipFree(d);
    }
    printf("\n");
}

int main() {
    run<0>("v_fma_f64 acc*b+acc (2 distinct srcs)");
    run<1>("v_fma_f64 acc += b[j]*c[j] (3 srcs)");
    run<2>("v_fma_f64 acc += b0*c[j] (shared b)");
    run<3>("v_fmac_f64 acc += b0*c[j]");
    run<4>("v_fma_f64 acc += b*b");
    run<5>("v_mul_f64 a = b*c");
    run<6>("v_fma_f32 3 srcs");
    run<7>("v_add3_u32 3 srcs");
    run<8>("v_fma_f64 acc += b0*sgpr");
    return 0;
}
