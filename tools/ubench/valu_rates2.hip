// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU / DS instructions the codec
// kernels are made of, at 1 and 2 waves per SIMD on gfx950.  Diagnostic tool, not part of the library.
//   hipcc -O2 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)


typedef int int4v __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void k(unsigned long long* out, int iters, int seed) {
    int i[8], i2[8]; int4v v4[2];
    for (int j = 0; j < 8; ++j) { i[j] = j * 77 + seed + threadIdx.x; i2[j] = j * 31 + seed * 3 + threadIdx.x; }
    v4[0] = v4[1] = int4v{seed, seed, seed, seed};
    unsigned long long msk = 0x5555aaaa3333ccccULL ^ seed, macc = 0;
    __shared__ int lds[4096];
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) lds[t] = seed;
    __syncthreads();
    const int lofs = (threadIdx.x & 1023) * 4, lofs2 = ((threadIdx.x & 1023) >> 1) * 4, lofs4 = (threadIdx.x & 255) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define OPX(j) \
            if constexpr (OP == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 1) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 2) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 3) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 4) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 5) { asm volatile("v_mov_b32 %0, %1" : "=v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 6) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(i[j])); } \
            else if constexpr (OP == 7) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 8) { asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(i[j])); } \
            else if constexpr (OP == 9) { asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 10) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 11) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 12) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 13) { asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 14) { asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 15) { asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 16) { asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 17) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 18) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 19) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 20) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[j]) : "v"(i[(j+1)&7])); } \
            else if constexpr (OP == 21) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "s"(msk)); } \
            else if constexpr (OP == 22) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i[j]) : "v"(i2[j]), "v"(i2[(j+1)&7])); } \
            else if constexpr (OP == 23) { asm volatile("v_cmp_lt_i32 vcc, %0, %1" :: "v"(i[j]), "v"(i[(j+1)&7]) : "vcc"); } \
            else if constexpr (OP == 24) { { unsigned long long m_; asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m_) : "v"(i[j]), "v"(i[(j+1)&7])); macc ^= m_; } } \
            else if constexpr (OP == 25) { asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(i[j]) : "v"(i2[j]), "v"(i2[(j+1)&7]) : "vcc"); } \
            else if constexpr (OP == 26) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(i[j]), "+v"(i2[j]) : "v"(i[(j+1)&7]), "v"(i2[(j+1)&7]) : "vcc"); } \
            else if constexpr (OP == 27) { asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(i[j]) :: "a0"); } \
            else if constexpr (OP == 28) { asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i[j]) : "v"(i[(j+1)&7]), "v"(i[(j+2)&7])); } \
            else if constexpr (OP == 29) { asm volatile("v_not_b32 %0, %0" : "+v"(i[j])); } \
            else if constexpr (OP == 30) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(i[0])); } \
            else if constexpr (OP == 31) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[0]) : "v"(i[1])); } \
            else if constexpr (OP == 32) { asm volatile("ds_or_b32 %0, %1" :: "v"(lofs), "v"(i[(j+1)&7]) : "memory"); } \
            else if constexpr (OP == 33) { asm volatile("ds_or_b32 %0, %1" :: "v"(lofs2), "v"(i[(j+1)&7]) : "memory"); } \
            else if constexpr (OP == 34) { asm volatile("ds_write_b32 %0, %1" :: "v"(lofs), "v"(i[(j+1)&7]) : "memory"); } \
            else if constexpr (OP == 35) { asm volatile("ds_read_b32 %0, %1" : "=v"(i[j]) : "v"(lofs) : "memory"); } \
            else if constexpr (OP == 36) { asm volatile("ds_read_b128 %0, %1" : "=v"(v4[j&1]) : "v"(lofs4) : "memory"); } \
            ;
            REP8(OPX)
#undef OPX
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int is = (int)macc + v4[0].x + v4[1].y;
    for (int j = 0; j < 8; ++j) is += i[j] + i2[j];
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    if (is == 77) out[0] = 1;
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
    run<0>("v_add_u32");
    run<1>("v_sub_u32");
    run<2>("v_and_b32");
    run<3>("v_or_b32");
    run<4>("v_xor_b32");
    run<5>("v_mov_b32");
    run<6>("v_lshlrev_b32 const");
    run<7>("v_lshrrev_b32 vgpr");
    run<8>("v_lshrrev_b32 const");
    run<9>("v_ashrrev_i32 vgpr");
    run<10>("v_max_i32");
    run<11>("v_min_u32");
    run<12>("v_add3_u32");
    run<13>("v_lshl_add_u32");
    run<14>("v_add_lshl_u32");
    run<15>("v_xad_u32");
    run<16>("v_bfi_b32");
    run<17>("v_perm_b32");
    run<18>("v_mad_u32_u24");
    run<19>("v_mul_u32_u24");
    run<20>("v_cndmask vcc (VOP2)");
    run<21>("v_cndmask e64 sgpr mask");
    run<22>("v_cndmask distinct dst");
    run<23>("v_cmp_lt_i32 vcc");
    run<24>("v_cmp_lt_i32 e64 sgpr");
    run<25>("v_cmp + v_cndmask pair");
    run<26>("v_add_co+v_addc (64b add)");
    run<27>("v_accvgpr_write+read");
    run<28>("v_med3_i32");
    run<29>("v_sub_u32 sdwa-free not");
    run<30>("v_lshlrev_b32 dep chain");
    run<31>("v_add_u32 dep chain");
    run<32>("ds_or_b32 distinct words");
    run<33>("ds_or_b32 2 lanes per word");
    run<34>("ds_write_b32");
    run<35>("ds_read_b32");
    run<36>("ds_read_b128");
    return 0;
}
