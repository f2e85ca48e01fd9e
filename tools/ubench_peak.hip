// ubench_peak.hip -- what a plain device copy reaches on this GPU, the yardstick the decode is priced against (not part
// of the library).  Variants: U independent 16-byte loads in flight per lane before the first store (1, 2, 4, 8), one-shot
// grids (every lane copies U pieces and exits) against grid-stride grids sized to the chip (256 CUs x waves), plain and
// non-temporal loads / stores, and the pure read and pure write rates beside them.  Bytes reported are read + written.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld(const u32x4* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st(u32x4* p, u32x4 v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// one-shot: block b copies pieces [b * 256 * U, (b + 1) * 256 * U), lane-contiguous within each of the U rounds
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_once(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + 256 * u < n) v[u] = ld(src + base + 256 * u, NTL);
#pragma unroll
    for (int u = 0; u < U; u++) if (base + 256 * u < n) st(dst + base + 256 * u, v[u], NTS);
}
// grid-stride: `gridDim.x` blocks walk the buffer, U loads in flight per lane and step
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_stride(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t step = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += step) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (base + 256 * u < n) v[u] = ld(src + base + 256 * u, NTL);
#pragma unroll
        for (int u = 0; u < U; u++) if (base + 256 * u < n) st(dst + base + 256 * u, v[u], NTS);
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_only(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t step = (size_t)gridDim.x * 256 * U;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += step) {
#pragma unroll
        for (int u = 0; u < U; u++) if (base + 256 * u < n) acc ^= src[base + 256 * u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[threadIdx.x] = acc;
}
template <int U, bool NTS>
__global__ __launch_bounds__(256) void write_only(u32x4* __restrict__ dst, size_t n) {
    const size_t step = (size_t)gridDim.x * 256 * U;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += step) {
#pragma unroll
        for (int u = 0; u < U; u++) if (base + 256 * u < n) st(dst + base + 256 * u, v, NTS);
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoi(argv[1]) : 4ull) << 30, n = bytes / 16;
    u32x4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto fn) {
        fn(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int it = 0; it < 7; it++) {
            (void)hipEventRecord(e0); fn(); (void)hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s failed\n", name); exit(1); }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-72s %8.3f ms  %8.1f GB/s\n", name, best, moved / 1e9 / (best / 1e3));
    };
    char name[160];
#define ONCE(U, NTL, NTS) do { snprintf(name, sizeof name, "copy, one-shot grid, %d loads in flight per lane%s%s", U, NTL ? ", nt loads" : "", NTS ? ", nt stores" : ""); \
    timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_once<U, NTL, NTS>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, a, b, n); }); } while (0)
#define STRIDE(U, G, NTL, NTS) do { snprintf(name, sizeof name, "copy, grid-stride %5d blocks, %d loads in flight per lane%s%s", G, U, NTL ? ", nt loads" : "", NTS ? ", nt stores" : ""); \
    timeit(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_stride<U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n); }); } while (0)
    ONCE(1, false, false); ONCE(2, false, false); ONCE(4, false, false); ONCE(8, false, false);
    ONCE(4, false, true); ONCE(4, true, true); ONCE(8, true, true);
    STRIDE(1, 2048, false, false); STRIDE(4, 1024, false, false); STRIDE(4, 2048, false, false); STRIDE(4, 4096, false, false);
    STRIDE(4, 8192, false, false); STRIDE(8, 2048, false, false); STRIDE(8, 4096, false, false);
    STRIDE(4, 2048, false, true); STRIDE(4, 2048, true, true); STRIDE(8, 2048, true, true); STRIDE(4, 4096, true, true);
    timeit("read only, grid-stride 4096 blocks, 4 loads in flight", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_only<4>), dim3(4096), dim3(256), 0, 0, a, b, n); });
    timeit("read only, grid-stride 4096 blocks, 8 loads in flight", 1.0 * bytes, [&] { hipLaunchKernelGGL((read_only<8>), dim3(4096), dim3(256), 0, 0, a, b, n); });
    timeit("write only, grid-stride 4096 blocks, plain stores", 1.0 * bytes, [&] { hipLaunchKernelGGL((write_only<4, false>), dim3(4096), dim3(256), 0, 0, b, n); });
    timeit("write only, grid-stride 4096 blocks, nt stores", 1.0 * bytes, [&] { hipLaunchKernelGGL((write_only<4, true>), dim3(4096), dim3(256), 0, 0, b, n); });
    timeit("hipMemcpyAsync device to device", 2.0 * bytes, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
