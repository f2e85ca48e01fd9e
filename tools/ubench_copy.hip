// ubench_copy.hip -- which part of the chunk-organised string copy costs what: a family of kernels from a bare
// gather-and-store to the copy kernel's full structure, on synthetic rows of 32..80 bytes out of a 1 MB dictionary
// (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(uint8_t* p, uint4 w) { u32x4 t = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(t, (u32x4*)p); }
__device__ __forceinline__ uint4 ld16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint64_t low_bytes(int x) { x = max(x, 0); return x >= 8 ? ~0ull : ((1ull << (8 * x)) - 1ull); }

#ifndef ROWS_
#define ROWS_ 256
#endif
#ifndef EDGE_FULL
#define EDGE_FULL 0
#endif
constexpr int ROWS = ROWS_;  // rows per tile
// off[r]: output offset of row r (tile-relative offsets are off[r] - off[r0]); src[r]: position in the dictionary
// LEVEL 0: chunk -> row by arithmetic guess (no search), one gather, store; no LDS
// LEVEL 1: + tile's offsets staged in LDS behind a barrier (search still skipped: row from the guess, corrected by nothing)
// LEVEL 2: + the real 8-step binary search in LDS
// LEVEL 3: + straddle loop (real merge of following rows)
template <int LEVEL, int U>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ dict, const uint32_t* __restrict__ off, const uint32_t* __restrict__ src,
                                        uint8_t* __restrict__ out, uint32_t ntiles) {
    __shared__ uint32_t s_off[ROWS + 1];
    __shared__ uint32_t s_dlt[ROWS];
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
#ifdef REUSE_TILES
    const uint32_t r0 = (tile % REUSE_TILES) * ROWS;
    out += (size_t)(tile / REUSE_TILES) * (off[REUSE_TILES * ROWS] & ~15u);
#else
    const uint32_t r0 = tile * ROWS;
#endif
    const int tid = threadIdx.x;
    uint32_t b0, b1;
    if (LEVEL >= 1) {
        for (uint32_t i = tid; i <= ROWS; i += 256) s_off[i] = off[r0 + i];
        for (uint32_t i = tid; i < ROWS; i += 256) s_dlt[i] = src[r0 + i] - off[r0 + i];
        __syncthreads();
        b0 = s_off[0]; b1 = s_off[ROWS];
    } else {
        b0 = off[r0]; b1 = off[r0 + ROWS];
    }
    for (uint32_t c0 = (b0 & ~15u) + 16u * tid; c0 < b1; c0 += 16u * 256u * U) {
        uint4 w[U];
        uint32_t row[U], cp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t cpos = c0 + 16u * 256u * u;
            cp[u] = cpos;
            const uint32_t first = max(cpos, b0);
            uint32_t r = 0;
            if (LEVEL >= 2) {
#pragma unroll
                for (uint32_t step = ROWS / 2; step >= 1; step >>= 1) {
                    const uint32_t c = r + step;
                    const uint32_t o = s_off[min(c, (uint32_t)ROWS)];
                    if (c < ROWS && o <= first) r = c;
                }
            } else {
                r = min((uint32_t)ROWS - 1, (uint32_t)(((uint64_t)(first - b0) * ROWS) / (b1 - b0)));
            }
            row[u] = r;
            // source position: inside the dictionary whatever the row guess was (clamped), so that every level reads valid memory
            uint32_t sp;
            if (LEVEL >= 1) sp = s_dlt[r] + cpos; else sp = src[r0 + r] + (cpos - off[r0 + r]);
            if (LEVEL < 2) sp &= 0xFFFFFu;  // the guessed row may be off: keep the address inside the 1 MB dictionary
            if (cpos < b1) w[u] = ld16u(dict + 64 + (sp & 0x1FFFFFu)); else w[u] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t cpos = cp[u];
            if (cpos >= b1) continue;
            uint4 v = w[u];
            const uint32_t first = max(cpos, b0), cend = min(cpos + 16u, b1);
            if (LEVEL >= 3) {
                uint32_t r = row[u];
                uint32_t rend = s_off[r + 1];
                uint32_t pos = min(cend, rend);
                if (pos < cend) {
                    uint64_t wl = (uint64_t)v.x | ((uint64_t)v.y << 32), wh = (uint64_t)v.z | ((uint64_t)v.w << 32);
                    while (pos < cend) {
                        do { r++; rend = s_off[r + 1]; } while (rend <= pos && r + 1 < ROWS);
                        const uint32_t segend = min(cend, max(rend, pos + 1));
                        const uint4 x = ld16u(dict + 64 + ((s_dlt[r] + cpos) & 0x1FFFFFu));
                        const int a = (int)(pos - cpos), e = (int)(segend - cpos);
                        const uint64_t ml = low_bytes(e) & ~low_bytes(a), mh = low_bytes(e - 8) & ~low_bytes(a - 8);
                        wl = (wl & ~ml) | (((uint64_t)x.x | ((uint64_t)x.y << 32)) & ml);
                        wh = (wh & ~mh) | (((uint64_t)x.z | ((uint64_t)x.w << 32)) & mh);
                        pos = segend;
                    }
                    v = make_uint4((uint32_t)wl, (uint32_t)(wl >> 32), (uint32_t)wh, (uint32_t)(wh >> 32));
                }
            }
            if (EDGE_FULL || (first == cpos && cend == cpos + 16u)) st_nt(out + cpos, v);
            else {
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 16; k++) { const uint32_t q = cpos + k; if (q >= first && q < cend) out[q] = (uint8_t)(wv[k >> 2] >> (8 * (k & 3))); }
            }
        }
    }
}
int main() {
    const uint32_t ntiles = (1u << 26) / ROWS;        // 64 M rows
    const size_t nrows = (size_t)ntiles * ROWS;
    std::vector<uint32_t> off(nrows + 1), src(nrows);
    uint64_t x = 88172645463325252ull, pos = 0;
    for (size_t i = 0; i < nrows; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        off[i] = (uint32_t)pos;
        pos += 32 + (x % 49);                        // 32..80 bytes, mean 56
        src[i] = (uint32_t)((x >> 20) % ((1u << 20) - 128));
    }
    off[nrows] = (uint32_t)pos;
    if (pos >= (1ull << 32)) { printf("too many bytes\n"); return 1; }
    uint8_t *dict, *out; uint32_t *doff, *dsrc;
    CK(hipMalloc(&dict, (2u << 20) + 256)); CK(hipMemset(dict, 5, (2u << 20) + 256));
    CK(hipMalloc(&out, pos + 64));
    CK(hipMalloc(&doff, off.size() * 4)); CK(hipMalloc(&dsrc, src.size() * 4));
    CK(hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
        fn(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int it = 0; it < 4; it++) { (void)hipEventRecord(e0); fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-64s %8.3f ms  %8.1f GB/s of body\n", name, best, pos / 1e9 / (best / 1e3));
    };
#define RUN(L, U, name) timeit(name, [&] { hipLaunchKernelGGL((k<L, U>), dim3(ntiles), dim3(256), 0, 0, dict, doff, dsrc, out, ntiles); })
    RUN(0, 2, "0: guess row, gather, store; no LDS (2 chunks per lane)");
    RUN(0, 4, "0: same, 4 chunks per lane");
    RUN(1, 2, "1: + offsets staged in LDS behind a barrier (2 chunks)");
    RUN(1, 4, "1: same, 4 chunks per lane");
    RUN(2, 2, "2: + 8-step binary search in LDS (2 chunks)");
    RUN(2, 4, "2: same, 4 chunks per lane");
    RUN(3, 2, "3: + straddle loop = the copy kernel (2 chunks)");
    RUN(3, 4, "3: same, 4 chunks per lane");
    return 0;
}
