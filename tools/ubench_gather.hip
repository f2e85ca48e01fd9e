// Where do gathered rows come from fastest?  Every block writes 64 KB * kTiles of output made of 48-byte rows picked at
// random (positions hashed in registers: no index is read from memory) out of "its" dictionary of `dict_bytes` bytes;
// `share` consecutive blocks use the same dictionary, dictionaries lie `dict_stride` apart in one big source buffer.
//   global: rows gathered straight from global memory (L2 when the dictionary is resident there, HBM when not)
//   lds:    the block first streams its dictionary into LDS (coalesced 16-byte loads), then gathers from LDS
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gather.hip -o tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void st_nt(uint4* p, uint4 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const uint8_t* __restrict__ src, uint4* __restrict__ out, uint32_t vec_per_block,
                                             uint32_t dict_bytes, uint64_t dict_stride, uint32_t share, uint32_t n_dicts) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t d = (blockIdx.x / share) % n_dicts;
    const uint8_t* dict = src + d * dict_stride;
    if (LDS) {
        for (uint32_t i = threadIdx.x * 16; i < dict_bytes; i += THREADS * 16) *(uint4*)(lds + i) = *(const uint4*)(dict + i);
        __syncthreads();
    }
    uint4* o = out + (size_t)blockIdx.x * vec_per_block;
    constexpr int U = 4;
    for (uint32_t i0 = threadIdx.x; i0 < vec_per_block; i0 += THREADS * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + THREADS * u;
            const uint32_t g = blockIdx.x * vec_per_block + i;
            const uint32_t p = hash32(g / 3) % (dict_bytes - 64) + (g % 3) * 16;  // < dict_bytes - 16
            if (LDS) __builtin_memcpy(&v[u], lds + p, 16); else __builtin_memcpy(&v[u], dict + p, 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = i0 + THREADS * u;
            if (i < vec_per_block) st_nt(o + i, v[u]);
        }
    }
}

int main(int argc, char** argv) {
    const size_t out_bytes = (argc > 1 ? (size_t)atoi(argv[1]) : 1ull) << 30;
    const uint32_t dict_bytes = 58 * 1024;
    const uint64_t dict_stride = 128 * 1024;
    const uint32_t max_dicts = 4096;
    uint4* out; CK(hipMalloc(&out, out_bytes)); CK(hipMemset(out, 1, out_bytes));
    uint8_t* src; CK(hipMalloc(&src, max_dicts * dict_stride)); CK(hipMemset(src, 3, max_dicts * dict_stride));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0)); fn(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-72s %8.3f ms  %8.1f GB/s\n", name, best, out_bytes / 1e9 / (best / 1e3));
    };
    CK(hipFuncSetAttribute((const void*)k<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    char name[160];
    // global gathers: 256 threads, 64 KB of output per block, 16 blocks per dictionary (one pack = 4 MB of rows)
    for (uint32_t nd : {1u, 8u, 64u, 256u, 1024u, 4096u}) {
        const uint32_t vpb = 4096, nblocks = (uint32_t)(out_bytes / 16 / vpb);
        snprintf(name, sizeof name, "global gathers, %4u dictionaries of 58 KB (%6.1f MB), 16 blocks each", nd, nd * 58 / 1024.0);
        timeit(name, [&] { hipLaunchKernelGGL((k<false, 256>), dim3(nblocks), dim3(256), 0, 0, src, out, vpb, dict_bytes, dict_stride, 16u, nd); });
    }
    {
        const uint32_t vpb = 4096, nblocks = (uint32_t)(out_bytes / 16 / vpb);
        timeit("global gathers, 4096 dictionaries, 1 block each (every dictionary read once)", [&] { hipLaunchKernelGGL((k<false, 256>), dim3(nblocks), dim3(256), 0, 0, src, out, vpb, dict_bytes, dict_stride, 1u, 4096u); });
    }
    // LDS gathers: T threads, T/64 tiles of 64 KB per block, lds_kb of LDS requested (sets the blocks per compute unit)
    auto lds_run = [&](auto kern, uint32_t threads, uint32_t lds_kb) {
        const uint32_t vpb = 4096 * (threads / 64), nblocks = (uint32_t)(out_bytes / 16 / vpb);
        snprintf(name, sizeof name, "LDS gathers, %4u threads, %3u KB of LDS per block, %u blocks per dictionary", threads, lds_kb, 64 / (threads / 64));
        timeit(name, [&] { hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), lds_kb * 1024, 0, src, out, vpb, dict_bytes, dict_stride, 64 / (threads / 64), 1024u); });
    };
    lds_run(k<true, 256>, 256, 58);
    lds_run(k<true, 256>, 256, 91);
    lds_run(k<true, 512>, 512, 58);
    lds_run(k<true, 512>, 512, 124);
    lds_run(k<true, 1024>, 1024, 58);
    lds_run(k<true, 1024>, 1024, 160);
    return 0;
}
