// ubench.hip -- memory-pipeline calibration microbenchmarks for MI355X (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void w16_stride(uint4* out, size_t nvec) {   // grid-stride coalesced 16B stores
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nvec; i += gridDim.x * 256ull) out[i] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void w16_tile(uint4* out, size_t vec_per_block) {  // each block writes its own contiguous tile
    uint4* o = out + blockIdx.x * vec_per_block;
    for (size_t i = threadIdx.x; i < vec_per_block; i += 256) o[i] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void copy16(const uint4* in, uint4* out, size_t nvec) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nvec; i += gridDim.x * 256ull) out[i] = in[i];
}
// gather: each lane reads an UNALIGNED 16 bytes at src + rowbase + k (rows of ~50 B, random row per 3 chunks) and stores aligned
__global__ __launch_bounds__(256) void gather16u(const uint8_t* src, const uint32_t* rowpos, uint4* out, size_t vec_per_block, uint32_t dict_bytes) {
    uint4* o = out + blockIdx.x * vec_per_block;
    for (size_t i = threadIdx.x; i < vec_per_block; i += 256) {
        size_t g = blockIdx.x * vec_per_block + i;
        uint32_t p = rowpos[g / 3] + (uint32_t)(g % 3) * 16;  // chunk g of a 48-byte row
        uint4 v; __builtin_memcpy(&v, src + p, 16);
        o[i] = v;
    }
}
// same with two ALIGNED loads + byte funnel
__global__ __launch_bounds__(256) void gather16a(const uint8_t* src, const uint32_t* rowpos, uint4* out, size_t vec_per_block, uint32_t dict_bytes) {
    uint4* o = out + blockIdx.x * vec_per_block;
    for (size_t i = threadIdx.x; i < vec_per_block; i += 256) {
        size_t g = blockIdx.x * vec_per_block + i;
        uint32_t p = rowpos[g / 3] + (uint32_t)(g % 3) * 16;
        const uint4* q = (const uint4*)(src + (p & ~15u));
        uint4 a = q[0], b = q[1];
        uint32_t sh = p & 15u;
        uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t d = sh >> 2, bs = (sh & 3) * 8;
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t lo = d == 0 ? w[k] : d == 1 ? w[k + 1] : d == 2 ? w[k + 2] : w[k + 3];
            uint32_t hi = d == 0 ? w[k + 1] : d == 1 ? w[k + 2] : d == 2 ? w[k + 3] : w[k + 4];
            r[k] = bs ? (lo >> bs) | (hi << (32 - bs)) : lo;
        }
        o[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}
int main() {
    const size_t bytes = 4ull << 30, nvec = bytes / 16;
    uint4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const uint32_t dict_bytes = 900 * 1024;  // one pack's dictionary
    const int ndict = 512;
    uint8_t* dict; CK(hipMalloc(&dict, (size_t)dict_bytes * ndict + 64));
    CK(hipMemset(dict, 3, (size_t)dict_bytes * ndict + 64));
    const size_t vec_per_block = 736;  // ~11.5 KB per block like a 256-row tile of 45-byte strings
    const size_t nblocks = nvec / vec_per_block;
    std::vector<uint32_t> rp(nvec / 3 + 16);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < rp.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; rp[i] = (uint32_t)(x % (dict_bytes - 64)); }
    // rows of block blk read the dictionary (blk / 256) % ndict  -> add base per block on device side via rowpos itself
    for (size_t g = 0; g < rp.size(); g++) { size_t blk = (g * 3) / vec_per_block; rp[g] += (uint32_t)(((blk / 256) % ndict) * (size_t)dict_bytes % 0xFFFFFFFFull); }
    uint32_t* drp; CK(hipMalloc(&drp, rp.size() * 4)); CK(hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn, double gb) {
        fn(); hipDeviceSynchronize();
        float best = 1e9;
        for (int it = 0; it < 5; it++) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-28s %8.3f ms  %8.1f GB/s\n", name, best, gb / (best / 1e3));
    };
    timeit("w16_stride grid 2048", [&] { hipLaunchKernelGGL(w16_stride, dim3(2048), dim3(256), 0, 0, a, nvec); }, 4.295);
    timeit("w16_stride grid 16384", [&] { hipLaunchKernelGGL(w16_stride, dim3(16384), dim3(256), 0, 0, a, nvec); }, 4.295);
    timeit("w16_tile 11.5KB/block", [&] { hipLaunchKernelGGL(w16_tile, dim3(nblocks), dim3(256), 0, 0, a, vec_per_block); }, nblocks * vec_per_block * 16 / 1e9);
    timeit("copy16 grid 2048", [&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, a, b, nvec); }, 2 * 4.295);
    timeit("copy16 grid 16384", [&] { hipLaunchKernelGGL(copy16, dim3(16384), dim3(256), 0, 0, a, b, nvec); }, 2 * 4.295);
    timeit("gather16 unaligned (out GB/s)", [&] { hipLaunchKernelGGL(gather16u, dim3(nblocks), dim3(256), 0, 0, dict, drp, a, vec_per_block, dict_bytes); }, nblocks * vec_per_block * 16 / 1e9);
    // the same gathers when every block reads ONE 900 KB dictionary (resident in every XCD's L2)
    std::vector<uint32_t> rp1(rp.size());
    for (size_t i = 0; i < rp1.size(); i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; rp1[i] = (uint32_t)(x % (dict_bytes - 64)); }
    uint32_t* drp1; CK(hipMalloc(&drp1, rp1.size() * 4)); CK(hipMemcpy(drp1, rp1.data(), rp1.size() * 4, hipMemcpyHostToDevice));
    timeit("gather16 unaligned, L2 dict", [&] { hipLaunchKernelGGL(gather16u, dim3(nblocks), dim3(256), 0, 0, dict, drp1, a, vec_per_block, dict_bytes); }, nblocks * vec_per_block * 16 / 1e9);
    // aligned variant of the same: rows at multiples of 16
    for (auto& v : rp1) v &= ~15u;
    CK(hipMemcpy(drp1, rp1.data(), rp1.size() * 4, hipMemcpyHostToDevice));
    timeit("gather16 aligned rows, L2 dict", [&] { hipLaunchKernelGGL(gather16u, dim3(nblocks), dim3(256), 0, 0, dict, drp1, a, vec_per_block, dict_bytes); }, nblocks * vec_per_block * 16 / 1e9);
    for (auto& v : rp) v &= ~15u;
    CK(hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    timeit("gather16 aligned rows, 512 dicts", [&] { hipLaunchKernelGGL(gather16u, dim3(nblocks), dim3(256), 0, 0, dict, drp, a, vec_per_block, dict_bytes); }, nblocks * vec_per_block * 16 / 1e9);
    timeit("gather16 2xaligned (out GB/s)", [&] { hipLaunchKernelGGL(gather16a, dim3(nblocks), dim3(256), 0, 0, dict, drp, a, vec_per_block, dict_bytes); }, nblocks * vec_per_block * 16 / 1e9);
    return 0;
}
