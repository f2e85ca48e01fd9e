// ubench_ta.hip -- what a compute unit's vector memory path sustains for the copy shapes of the string decode, with enough
// independent work per lane that latency is not the limit (not part of the library).  Sources are small (L2 resident),
// destinations are streamed to HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(uint4* p, uint4 w) { u32x4 t = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(t, (u32x4*)p); }

// MODE 0: aligned coalesced loads from a 1 MB source; 1: unaligned, coalesced (source shifted by 5 bytes); 2: unaligned random 48-byte rows;
// 3: random aligned 16-byte pieces; 4: no loads (stores only); 5: loads only (sum kept, one store per block)
template <int MODE, bool NT, int U = 4>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ src, const uint32_t* __restrict__ rowpos, uint4* __restrict__ out,
                                        size_t vec_per_block, uint32_t src_bytes, size_t rp_len) {
    uint4* o = out + blockIdx.x * vec_per_block;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __shared__ uint32_t tab[257];
    if (MODE == 7 || MODE == 9) {
        for (uint32_t i = threadIdx.x; i <= 256; i += 256) tab[i] = i * 56u;  // offsets of 256 rows of 56 bytes
        __syncthreads();
    }
    for (size_t i0 = threadIdx.x; i0 < vec_per_block; i0 += 256 * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + 256 * u;
            const size_t g = blockIdx.x * vec_per_block + i;
            uint32_t p;
            if (MODE == 0) p = (uint32_t)((g * 16) % (src_bytes - 64));
            else if (MODE == 1) p = (uint32_t)((g * 16) % (src_bytes - 64)) + 5;
            else if (MODE == 2) p = rowpos[(g / 3) & 0xFFFFF] + (uint32_t)(g % 3) * 16;
            else if (MODE == 3) p = rowpos[g & 0xFFFFF] & ~15u;
            else if (MODE == 6) p = rowpos[(g / 3) % rp_len] + (uint32_t)(g % 3) * 16;                      // the index is streamed (4 bytes per 48)
            else if (MODE == 10) p = __builtin_nontemporal_load(rowpos + (g / 3) % rp_len) + (uint32_t)(g % 3) * 16;
            else if (MODE == 11) p = rowpos[(g / 3) % rp_len];
            else if (MODE == 12) p = (rowpos[(g / 12) % rp_len] + (uint32_t)(g % 12) * 16) % (src_bytes - 64);   // one streamed index word per 192 bytes
            else if (MODE == 13) p = (rowpos[(g / 48) % rp_len] + (uint32_t)(g % 48) * 16) % (src_bytes - 64);   // per 768 bytes
            else if (MODE == 14) p = (rowpos[(g / 192) % rp_len] + (uint32_t)(g % 192) * 16) % (src_bytes - 64); // per 3 KB
            else if (MODE == 15) {
                // the wave's 64 chunks (pieces g0 .. g0+63) read index words (g0/3) .. (g0+63)/3: a wave-uniform base, fetched
                // with scalar loads (the scalar cache's own path to L2), then picked per lane
                const size_t g0 = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((g - (threadIdx.x & 63)) & 0xFFFFFFFFu)) + ((g - (threadIdx.x & 63)) & ~0xFFFFFFFFull);
                const size_t w0 = (g0 / 3) % (rp_len - 24);
                typedef const uint32_t __attribute__((address_space(4))) * kptr;  // constant address space: uniform loads go to the scalar cache
                const uint64_t ba = (uint64_t)(rowpos + w0);
                const kptr base = (kptr)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ba >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ba));
                uint32_t t[24];
#pragma unroll
                for (int q = 0; q < 24; q++) t[q] = base[q];
                const uint32_t sel = (uint32_t)(g / 3 - g0 / 3);  // 0..21
                uint32_t x = t[0];
#pragma unroll
                for (int q = 1; q < 24; q++) x = sel == (uint32_t)q ? t[q] : x;
                p = x + (uint32_t)(g % 3) * 16;
            }
            else if (MODE == 8) p = (rowpos[(g / 3) % rp_len] ^ rowpos[(g / 3 + 1) % rp_len]) % (src_bytes - 64);   // two streamed index words per piece
            else if (MODE == 7 || MODE == 9) {
                // an 8-step search over 257 offsets in LDS, then the index word of the row it finds
                const uint32_t want = (uint32_t)((i * 16u) % (256u * 56u));
                uint32_t r = 0;
#pragma unroll
                for (uint32_t step = 128; step >= 1; step >>= 1) { const uint32_t c = r + step; if (c < 256 && tab[c] <= want) r = c; }
                p = MODE == 9 ? rowpos[(g / 3) % rp_len] + (want - tab[r]) % 48u : rowpos[((g / 3) & 0xFFF00) + r] + (want - tab[r]) % 40u;
            }
            else p = 0;
            if (MODE == 0) p &= ~15u;
            if (MODE == 11) v[u] = make_uint4(p, 1, 2, 3);
            else if (MODE != 4 && i < vec_per_block) __builtin_memcpy(&v[u], src + p, 16); else v[u] = make_uint4(i, 1, 2, 3);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = i0 + 256 * u;
            if (i < vec_per_block) {
                if (MODE == 5) { acc.x += v[u].x; acc.y ^= v[u].y; acc.z += v[u].z; acc.w ^= v[u].w; }
                else if (NT) st_nt(o + i, v[u]); else o[i] = v[u];
            }
        }
    }
    if (MODE == 5 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) o[threadIdx.x] = acc;
}
int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoi(argv[1]) : 4ull) << 30, nvec = bytes / 16;
    uint4* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 1, bytes));
    const uint32_t src_bytes = 1 << 20;
    uint8_t* src; CK(hipMalloc(&src, src_bytes + 256)); CK(hipMemset(src, 3, src_bytes + 256));
    std::vector<uint32_t> rp((4ull << 30) / 16 / 3 + 1024);  // 358 MB: beyond the memory-side cache
    uint64_t x = 88172645463325252ull;
    for (auto& v : rp) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)(x % (src_bytes - 64)); }
    uint32_t* drp; CK(hipMalloc(&drp, rp.size() * 4)); CK(hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    const size_t vpb = 4096;  // 64 KB of output per block
    const size_t nblocks = nvec / vpb;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
        fn(); hipDeviceSynchronize();
        float best = 1e9;
        for (int it = 0; it < 5; it++) { hipEventRecord(e0); fn(); hipError_t le = hipGetLastError(); hipEventRecord(e1); hipError_t se = hipEventSynchronize(e1); if (le != hipSuccess || se != hipSuccess) { printf("%s: launch %s, sync %s\n", name, hipGetErrorString(le), hipGetErrorString(se)); exit(1); } float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-58s %8.3f ms  %8.1f GB/s of pieces\n", name, best, nblocks * vpb * 16 / 1e9 / (best / 1e3));
    };
#define RUN(M, NT, name) timeit(name, [&] { hipLaunchKernelGGL((k<M, NT>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); })
    RUN(4, true, "stores only, nt");
    RUN(4, false, "stores only, plain");
    RUN(5, true, "loads only: aligned coalesced, L2 source");
    RUN(0, true, "aligned coalesced loads + nt stores");
    RUN(0, false, "aligned coalesced loads + plain stores");
    RUN(1, true, "unaligned coalesced loads + nt stores");
    RUN(2, true, "unaligned random 48-byte rows + nt stores");
    RUN(3, true, "aligned random 16-byte pieces + nt stores");
    RUN(6, true, "rows of 48 bytes, index streamed from HBM + nt stores");
#define RUNU(M, UU, name) timeit(name, [&] { hipLaunchKernelGGL((k<M, true, UU>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); })
    timeit("  same with plain stores", [&] { hipLaunchKernelGGL((k<6, false, 4>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); });
    timeit("  same, index loads non-temporal", [&] { hipLaunchKernelGGL((k<10, true, 4>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); });
    timeit("  same, index nt + plain stores", [&] { hipLaunchKernelGGL((k<10, false, 4>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); });
    timeit("  index streamed, NO gather (index value stored)", [&] { hipLaunchKernelGGL((k<11, true, 4>), dim3(nblocks), dim3(256), 0, 0, src, drp, a, vpb, src_bytes, rp.size() - 8); });
    RUNU(15, 4, "  index streamed through the SCALAR cache (s_load), 4 pieces");
    RUNU(15, 1, "  same, 1 piece");
    RUNU(12, 4, "  one streamed index word per 192 bytes of output");
    RUNU(13, 4, "  one streamed index word per 768 bytes of output");
    RUNU(14, 4, "  one streamed index word per 3 KB of output");
    RUNU(6, 1, "  same, 1 piece per lane in flight");
    RUNU(6, 2, "  same, 2 pieces");
    RUNU(6, 8, "  same, 8 pieces");
    RUNU(6, 16, "  same, 16 pieces");
    RUNU(2, 1, "index from the 4 MB table, 1 piece per lane in flight");
    RUNU(2, 2, "  same, 2 pieces");
    RUN(8, true, "two streamed index words per piece + nt stores");
    RUN(7, true, "8-step LDS search, index from a 4 MB table + nt stores");
    RUN(9, true, "8-step LDS search, index streamed + nt stores");
    return 0;
}
