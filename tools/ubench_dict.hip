// ubench_dict.hip -- what a dictionary lookup out[i] = dict[id[i]] can reach on MI355X, apart from the decoder (not part of
// the library): ids already expanded to uint16 in memory, one 1024-row tile per wave as in k_dict_num, dictionaries of
// several sizes, gathered from global memory or from a copy staged in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kPackRows = 65536;

template <typename U, bool kLds, bool kXcd>
__global__ __launch_bounds__(256) void dict_gather(const uint16_t* __restrict__ ids, const U* __restrict__ dicts, U* __restrict__ out,
                                                   uint32_t ec, uint32_t ndict) {
    extern __shared__ __align__(16) uint8_t s_raw[];
    // all blocks of a pack on one XCD (workgroups are dealt round-robin over the 8 XCDs by linear id), as the decoder does
    const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x, full = gridDim.y & ~7u;
    uint32_t bx = blockIdx.x, pack = blockIdx.y;
    if (kXcd && L < full * gridDim.x) { pack = ((L >> 3) / gridDim.x) * 8u + (L & 7u); bx = (L >> 3) % gridDim.x; }
    const uint32_t tile = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const U* dict = dicts + (size_t)(pack % ndict) * ec;
    if (kLds) {
        U* sd = (U*)s_raw;
        for (uint32_t i = threadIdx.x; i < ec; i += 256) sd[i] = dict[i];
        __syncthreads();
        dict = sd;
    }
    const size_t row0 = (size_t)pack * kPackRows + (size_t)tile * 1024;
    const uint32_t* id2 = (const uint32_t*)(ids + row0);
    uint32_t two[8];
#pragma unroll
    for (int q = 0; q < 8; q++) two[q] = id2[q * 64 + lane];
    U v[16];
#pragma unroll
    for (int q = 0; q < 8; q++) { v[2 * q] = dict[two[q] & 0xFFFFu]; v[2 * q + 1] = dict[two[q] >> 16]; }
    U* dst = out + row0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t r = q * 128 + 2 * lane;
        if constexpr (sizeof(U) == 4) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 t = {(uint32_t)v[2 * q], (uint32_t)v[2 * q + 1]};
            __builtin_nontemporal_store(t, (u32x2*)(dst + r));
        } else {
            typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
            u64x2 t = {(uint64_t)v[2 * q], (uint64_t)v[2 * q + 1]};
            __builtin_nontemporal_store(t, (u64x2*)(dst + r));
        }
    }
}

template <typename U>
int run(const char* tname, int npacks, uint16_t* d_ids, std::vector<uint16_t>& h_ids, U* d_dict, U* d_out, hipEvent_t e0, hipEvent_t e1) {
    const uint32_t ecs[] = {1000, 4096, 10000, 16384, 36500, 65536};
    const size_t rows = (size_t)npacks * kPackRows;
    for (uint32_t ec : ecs) {
        for (int seq = 0; seq < 2; seq++) {
            if (seq && ec != 65536) continue;
            uint64_t x = 88172645463325252ull + ec;
            for (size_t i = 0; i < rows; i++) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                h_ids[i] = seq ? (uint16_t)(i % kPackRows) : (uint16_t)(x % ec);
            }
            CK(hipMemcpy(d_ids, h_ids.data(), rows * 2, hipMemcpyHostToDevice));
            for (int lds = 0; lds < 3; lds++) {
                const size_t dict_bytes = (size_t)ec * sizeof(U);
                if (lds == 2 && dict_bytes > 64 * 1024) continue;
                auto launch = [&] {
                    if (lds == 2) hipLaunchKernelGGL((dict_gather<U, true, true>), dim3(16, npacks), dim3(256), dict_bytes, 0, d_ids, d_dict, d_out, ec, 512u);
                    else if (lds == 1) hipLaunchKernelGGL((dict_gather<U, false, true>), dim3(16, npacks), dim3(256), 0, 0, d_ids, d_dict, d_out, ec, 512u);
                    else hipLaunchKernelGGL((dict_gather<U, false, false>), dim3(16, npacks), dim3(256), 0, 0, d_ids, d_dict, d_out, ec, 512u);
                };
                launch();
                CK(hipDeviceSynchronize());
                float best = 1e9;
                for (int it = 0; it < 5; it++) {
                    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("%s dict %6u entries (%4zu KB) %-10s %-20s %7.3f ms  %7.1f G rows/s  %7.1f GB/s out\n", tname, ec, dict_bytes / 1024,
                       seq ? "ids 0..n-1" : "random ids", lds == 2 ? "LDS" : (lds ? "global, pack per XCD" : "global"), best, rows / (best * 1e6), rows * sizeof(U) / (best * 1e6));
            }
        }
    }
    return 0;
}

int main() {
    const int npacks = 2048;  // 134 M rows
    const size_t rows = (size_t)npacks * kPackRows;
    uint16_t* d_ids; void *d_dict, *d_out;
    CK(hipMalloc(&d_ids, rows * 2));
    CK(hipMalloc(&d_dict, 512ull * 65536 * 8));
    CK(hipMalloc(&d_out, rows * 8));
    CK(hipMemset(d_dict, 5, 512ull * 65536 * 8));
    std::vector<uint16_t> h_ids(rows);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (run<uint32_t>("u32", npacks, d_ids, h_ids, (uint32_t*)d_dict, (uint32_t*)d_out, e0, e1)) return 1;
    if (run<uint64_t>("u64", npacks, d_ids, h_ids, (uint64_t*)d_dict, (uint64_t*)d_out, e0, e1)) return 1;
    return 0;
}
