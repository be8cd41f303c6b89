// Micro-benchmark: what read+write rate a plain device copy reaches on this GPU, by access width, grid shape and
// cache policy -- the ceiling the compaction kernel (K5) is measured against.  Diagnostic tool.
//   hipcc -O3 --offload-arch=gfx950 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T, bool NT>
__global__ void copy_stride(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else dst[i] = src[i];
    }
}
template <typename T, bool NT, int U>
__global__ void copy_tile(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    // each block copies U consecutive tiles of blockDim.x elements: U loads in flight per thread
    const size_t base = ((size_t)blockIdx.x * U) * blockDim.x + threadIdx.x;
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * blockDim.x;
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * blockDim.x;
        if (i < n) { if constexpr (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename F>
void timeit(const char* name, size_t bytes, F&& launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %5.2f TB/s read+write\n", name, best, 2.0 * bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 9800000000ull / 16 * 16;
    void *s, *d;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(s, 3, bytes);
    (void)hipMemset(d, 0, bytes);
    timeit("hipMemcpyAsync D2D", bytes, [&] { (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    const size_t n4 = bytes / 4, n16 = bytes / 16;
    timeit("dword, grid-stride 8192x256", bytes, [&] { hipLaunchKernelGGL((copy_stride<uint32_t, false>), dim3(8192), dim3(256), 0, 0, (const uint32_t*)s, (uint32_t*)d, n4); });
    timeit("dword nt, grid-stride 8192x256", bytes, [&] { hipLaunchKernelGGL((copy_stride<uint32_t, true>), dim3(8192), dim3(256), 0, 0, (const uint32_t*)s, (uint32_t*)d, n4); });
    timeit("dwordx4, grid-stride 8192x256", bytes, [&] { hipLaunchKernelGGL((copy_stride<u32x4, false>), dim3(8192), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dwordx4 nt, grid-stride 8192x256", bytes, [&] { hipLaunchKernelGGL((copy_stride<u32x4, true>), dim3(8192), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dwordx4, grid-stride 2048x256", bytes, [&] { hipLaunchKernelGGL((copy_stride<u32x4, false>), dim3(2048), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dwordx4, tiles of 4, one shot", bytes, [&] { hipLaunchKernelGGL((copy_tile<u32x4, false, 4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dwordx4 nt, tiles of 4, one shot", bytes, [&] { hipLaunchKernelGGL((copy_tile<u32x4, true, 4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dwordx4, tiles of 8, one shot", bytes, [&] { hipLaunchKernelGGL((copy_tile<u32x4, false, 8>), dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n16); });
    timeit("dword, tiles of 8, one shot", bytes, [&] { hipLaunchKernelGGL((copy_tile<uint32_t, false, 8>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, 0, (const uint32_t*)s, (uint32_t*)d, n4); });
    return 0;
}
