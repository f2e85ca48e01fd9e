// ubench_bridge.hip -- from "rows gathered out of L2 and stored at the rate of plain stores" to the chunk loop of the
// string kernel, one ingredient at a time (not part of the library).  Every variant writes the same kind of output -- rows
// of a dictionary laid end to end, produced as 16-byte aligned chunks with one 16-byte gather per chunk -- and is timed on
// 64 M rows; what differs is named by the flags:
//   WAVE_TILES   a wave owns a tile of 1024 rows (two passes of 512) instead of a workgroup owning 64 KB of output
//   LDS_DLT      the rows' source positions are read from a table in LDS (built per pass) instead of computed
//   SEARCH       the row of a chunk is found by a 9-step binary search in the LDS offsets instead of by division
//   VARLEN       rows of 20..80 bytes (mean 50) instead of 48: chunks straddle rows
//   STRADDLE     chunks that straddle rows gather the following rows too (the merge loop)
//   MANY_DICTS   every pack of 64 tiles has its own 548 KB dictionary and all blocks of a pack sit on one XCD,
//                instead of one 1 MB dictionary for everything
//   IDS          the source position of a row comes from doff[id] with the 16-bit ids streamed from memory (2 bytes a row)
//   OCC6         at most 6 waves per SIMD (the string kernel's launch bound)
// and candidate remedies:
//   ZIPF         the ids follow a Zipf(1.1) law over the dictionary's entries (hot entries scattered) instead of a uniform one
//   TRIM         the last step of a pass skips the chunk slots no lane needs (wave-uniform) instead of repeating slot 0
//   ALIGN128     the chunk grid starts at a multiple of 128 bytes: every store instruction of a wave covers whole lines
//   TILE256      a wave owns 256 rows (one pass): four times the waves per pack, a quarter of the packs in flight per XCD
//   NOREMAP      per-pack dictionaries, but the blocks of a pack are dealt over the XCDs as they come (no XCD relabelling)
//   REMAP1       one dictionary, but with the XCD relabelling
//   ONE_DOFF     per-pack dictionaries, one offset table for all packs
//   PREFL        the same warm-up with ONE dword load per 128-byte line (two or three loads per lane instead of nine of 16 bytes)
//   PREFW        line warm-up of FOUR shares per workgroup (its own and the three that follow): with 64 small workgroups per pack
//                a share is 1/64 of the dictionary, and the first workgroups of a pack gather before the others have arrived
//   PREFS        the same with the four shares spread over the dictionary (own, +1/4, +2/4, +3/4): a quarter of the pack's
//                workgroups warms all of it
//   PREFA1/2/3   a workgroup warms its share (1/64) of the dictionary of the pack its XCD takes one, two or three packs LATER:
//                with 256-row tiles only three packs are in flight per XCD, so the L2 has room for the dictionaries to come
//   PREFK        line warm-up through SCALAR loads (one dword per line, a wave takes a quarter of the workgroup's share): they
//                fill L2 on their way like any load, but return through the scalar cache, not through the vector L1's
//                in-order queue, so a miss among them holds back nobody's gathers.  PREFK4: four spread shares as PREFS
//   PREFST       instead of warming by loads, a workgroup STORES its share of the pack's dictionary (the same bytes it holds):
//                what a producer that writes the dictionary just before its use would do; stores return nothing
//   FEWDICTS     per-pack addressing, but only four dictionaries in all (pack & 3): always in L2 after the first touches
//   COLDWARM     with FEWDICTS and PREFL: the warm-up loads read the PER-PACK dictionary (cold lines from HBM, as the kernel's do)
//                while the gathers use the four resident dictionaries: the price of the cold loads alone
//   LATE         with PREFL: nothing waits for the warm-up loads on their own; their registers are only "used" at the end of the
//                kernel, so the first wait of a wave (for its ids) covers both: one trip to cold memory per wave instead of two
//   PREFV        before its tiles, a workgroup streams its share (1/16) of the pack's dictionary through coalesced 16-byte loads
//                whose data it throws away: the gathers that follow find the lines in the XCD's L2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(uint8_t* p, uint4 w) { u32x4 t = {w.x, w.y, w.z, w.w}; __builtin_nontemporal_store(t, (u32x4*)p); }
__device__ __forceinline__ uint4 ld16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint64_t low_bytes(int x) { x = max(x, 0); return x >= 8 ? ~0ull : ((1ull << (8 * x)) - 1ull); }
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <typename U> __device__ __forceinline__ U wave_incl_scan(U v) {
    for (int d = 1; d < 64; d <<= 1) { U t = __shfl_up(v, d); if ((int)(threadIdx.x & 63) >= d) v += t; }
    return v;
}
enum { WAVE_TILES = 1, LDS_DLT = 2, SEARCH = 4, VARLEN = 8, STRADDLE = 16, MANY_DICTS = 32, IDS = 64, OCC6 = 128,
       ZIPF = 256, TRIM = 512, ALIGN128 = 1024, TILE256 = 2048, PREFV = 4096, NOREMAP = 8192, REMAP1 = 16384, ONE_DOFF = 32768, PREFL = 65536, LDS40 = 131072, LDS80 = 262144, PREFL64 = 524288, PREFW = 1 << 20, PREFS = 1 << 21, PREFA2 = 1 << 22, PREFA3 = 1 << 23, PREFA1 = 1 << 24, PREFK = 1 << 25, PREFK4 = 1 << 26, PREFST = 1 << 27, FEWDICTS = 1 << 28, COLDWARM = 1 << 29, LATE = 1 << 30 };
constexpr uint32_t kDictBytes = 548 * 1024, kDictEntries = 11000;
__device__ uint32_t g_cold_spread = 0;  // COLDWARM with a region: the region's lines lie this many times 128 bytes apart (the same number of lines over a wider address span)
__device__ uint32_t g_cold_region = 0;  // COLDWARM: wrap the cold warm-up addresses into this many bytes (0 = the per-pack dictionaries as they lie)

// block-organised baseline: a workgroup writes 64 KB of 48-byte rows, chunk -> row by division, position hashed
__global__ __launch_bounds__(256) void k_block(const uint8_t* __restrict__ dict, uint8_t* __restrict__ out, uint32_t dict_bytes) {
    uint8_t* o = out + (size_t)blockIdx.x * 65536;
    for (uint32_t i0 = threadIdx.x; i0 < 4096; i0 += 1024) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = i0 + 256 * u, row = blockIdx.x * 1366 + c / 3;
            v[u] = ld16u(dict + hash32(row) % (dict_bytes - 128) + (c % 3) * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) st_nt(o + (size_t)(i0 + 256 * u) * 16, v[u]);
    }
}

template <int F>
__device__ __forceinline__ void tiles_body(uint32_t bx, uint32_t by, const uint32_t gx, const uint8_t* __restrict__ dicts, const uint32_t* __restrict__ doffs,
                                           const uint16_t* __restrict__ ids, uint8_t* __restrict__ out, uint32_t npacks, uint32_t pack_out_stride) {
    constexpr uint32_t kTileRows = (F & TILE256) ? 256 : 1024, kPassRows = (F & TILE256) ? 256 : 512;
    constexpr uint32_t kTilesPerPack = 65536 / kTileRows;
    // 4128 bytes per wave, as the string kernel has (offsets and deltas of 512 rows); LDS40 / LDS80 pad a workgroup to 40 / 80 KB:
    // 4 / 2 workgroups (16 / 8 waves) per compute unit, so fewer packs in flight per XCD
    constexpr uint32_t kWaveLds = (F & LDS80) ? 20480 : ((F & LDS40) ? 10240 : 4 * (512 + 4) * 2);
    __shared__ __align__(16) uint8_t smem[4][kWaveLds];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tile = bx * 4 + wave, pack = by;
    const uint8_t* dict = (F & MANY_DICTS) ? dicts + (size_t)((F & FEWDICTS) ? (pack & 3u) : pack) * (kDictBytes + 256) : dicts;
    const uint32_t dict_span = (F & MANY_DICTS) ? kDictBytes : (1u << 20);
    const uint32_t* doff = doffs + (((F & MANY_DICTS) && !(F & ONE_DOFF)) ? (size_t)pack * (kDictEntries + 8) : 0);
    if (F & PREFV) {
        const uint32_t share = dict_span / gx & ~15u;  // bytes of the dictionary this workgroup warms up
        const uint8_t* p0 = dict + 64 + (size_t)bx * share;
        // all loads are issued before any is waited for (nine cover a sixteenth of 548 KB at 256 lanes x 16 bytes)
        uint4 v[9];
#pragma unroll
        for (uint32_t k = 0; k < 9; k++) {
            const uint32_t o = threadIdx.x * 16u + k * 4096u;
            v[k] = *(const uint4*)(p0 + min(o, share - 16u));
        }
#pragma unroll
        for (uint32_t k = 0; k < 9; k++) asm volatile("" ::"v"(v[k].x), "v"(v[k].y), "v"(v[k].z), "v"(v[k].w));  // the data is not used
    }
    if (F & PREFL64) {  // one dword per 64 bytes: six loads per lane
        const uint32_t share = (dict_span / gx + 127u) & ~127u;
        const uint8_t* p0 = dict + 64 + (size_t)bx * share;
        uint32_t v[6];
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) {
            const uint32_t o = (threadIdx.x + k * 256u) * 64u;
            v[k] = *(const uint32_t*)(p0 + min(o, share - 64u));
        }
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) asm volatile("" ::"v"(v[k]));
    }
    uint32_t lv[3] = {0, 0, 0};
    if (F & PREFL) {
        const uint32_t share = (dict_span / gx + 127u) & ~127u;
        const uint8_t* p0 = dict + 64 + (size_t)bx * share;
        if (F & COLDWARM) {
            size_t o = (size_t)pack * (kDictBytes + 256) + (size_t)bx * share;
            const uint32_t region = g_cold_region;
            if (region) o = (o % region) & ~(size_t)127;  // the share (<= 35 KB) stays inside the allocation: region <= 512 MB, 2 MB spare at its end
            p0 = dicts + 64 + o;
        }
        const uint32_t spread = (F & COLDWARM) ? g_cold_spread : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t o = (threadIdx.x + k * 256u) * 128u;
            if (spread) {  // line number within the region, times the spread: region * spread <= 512 MB
                const size_t line = ((size_t)(p0 - (dicts + 64)) + min(o, share - 128u)) >> 7;
                lv[k] = *(const uint32_t*)(dicts + 64 + line * 128u * spread);
            } else {
                lv[k] = *(const uint32_t*)(p0 + min(o, share - 128u));
            }
        }
        if (!(F & LATE)) {
#pragma unroll
            for (uint32_t k = 0; k < 3; k++) asm volatile("" ::"v"(lv[k]));
        }
    }
    if (F & (PREFA1 | PREFA2 | PREFA3)) {
        // the pack this XCD takes D packs from now (the relabelling above, inverted): none for the last D groups of eight
        const uint32_t D = (F & PREFA1) ? 1u : ((F & PREFA2) ? 2u : 3u);
        const uint32_t Lb = blockIdx.y * gx + blockIdx.x, full = npacks & ~7u;
        if (Lb < full * gx) {
            const uint32_t xcd = Lb & 7u, grp = (Lb >> 3) / gx + D;
            if (grp * 8u + 8u <= full) {
                const uint32_t ahead = grp * 8u + ((xcd + grp) & 7u);  // < full <= npacks
                const uint8_t* ad = dicts + (size_t)ahead * (kDictBytes + 256);
                const uint32_t lines = (dict_span + 127u) >> 7, per = (lines + gx - 1u) / gx;
                uint32_t v[2];
#pragma unroll
                for (uint32_t k = 0; k < 2; k++) {
                    const uint32_t idx = threadIdx.x + k * 256u, line = bx * per + idx;
                    v[k] = *(const uint32_t*)(ad + 64 + 128ull * ((idx < per && line < lines) ? line : 0u));
                }
#pragma unroll
                for (uint32_t k = 0; k < 2; k++) asm volatile("" ::"v"(v[k]));
            }
        }
    }
    if (F & PREFST) {
        const uint32_t lines = (dict_span + 127u) >> 7, per = (lines + gx - 1u) / gx;
        uint8_t* wd = const_cast<uint8_t*>(dict) + 64 + (size_t)bx * per * 128u;
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {
            const uint32_t idx = threadIdx.x + k * 256u;  // 16-byte piece of the share
            if (idx < per * 8u && bx * per * 128u + idx * 16u + 16u <= dict_span) *(uint4*)(wd + idx * 16u) = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
        }
    }
    uint32_t kv[18];
    if (F & (PREFK | PREFK4)) {
        typedef const uint32_t __attribute__((address_space(4))) * kp;
        const uint32_t lines = (dict_span + 127u) >> 7, per = (lines + gx - 1u) / gx, perw = (per + 3u) / 4u;
        const uint32_t wv = __builtin_amdgcn_readfirstlane(wave), bxu = __builtin_amdgcn_readfirstlane(bx);
        // PREFK: wave w takes lines [w * perw, (w + 1) * perw) of the workgroup's share; PREFK4: wave w takes the WHOLE share of
        // workgroup bx + w * gx / 4 (the four spread shares, one per wave)
        const uint32_t share = (F & PREFK4) ? (bxu + wv * (gx / 4u)) % gx : bxu;
        const uint32_t first = (F & PREFK4) ? 0u : wv * perw, count = (F & PREFK4) ? per : perw;
#pragma unroll
        for (uint32_t k = 0; k < 18; k++) {
            const uint32_t within = first + k, line = share * per + within;
            const bool ok = k < count && within < per && line < lines;
            const uint64_t a = (uint64_t)(dict + 64 + 128ull * (ok ? line : 0u));
            const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
            kv[k] = *(kp)au;
        }
    }
    if (F & (PREFW | PREFS)) {
        const uint32_t lines = (dict_span + 127u) >> 7, per = (lines + gx - 1u) / gx;  // lines per share
        constexpr uint32_t NL = (F & TILE256) ? 3 : 5;  // 64 workgroups: 4 shares = 276 lines; 16 workgroups: 4 shares = 1096 lines
        uint32_t v[NL];
#pragma unroll
        for (uint32_t k = 0; k < NL; k++) {
            const uint32_t idx = threadIdx.x + k * 256u, s = idx / per, within = idx - s * per;
            const uint32_t share = ((F & PREFS) ? bx + s * (gx / 4u) : bx + s) % gx;
            const uint32_t line = share * per + within;
            v[k] = *(const uint32_t*)(dict + 64 + 128ull * ((s < 4u && line < lines) ? line : 0u));  // line < lines: inside the dictionary
        }
#pragma unroll
        for (uint32_t k = 0; k < NL; k++) asm volatile("" ::"v"(v[k]));
    }
    uint32_t* s_off = (uint32_t*)smem[wave];
    uint32_t* s_dlt = s_off + kPassRows + 4;
    uint8_t* obase = out + (size_t)pack * pack_out_stride + (size_t)tile * (kTileRows * 50 + 1024);  // every tile has its own stretch
    uint32_t run = 0;
    for (uint32_t pass = 0; pass < kTileRows / kPassRows; pass++) {
        // ---- the pass's rows: lengths, offsets (scan), source positions; four neighbouring rows per lane
        const uint32_t row0 = (pack * kTilesPerPack + tile) * kTileRows + pass * kPassRows;
#pragma unroll
        for (uint32_t q = 0; q < kPassRows / 256; q++) {
            const uint32_t rl = q * 256 + 4 * lane;
            uint32_t len[4], src[4], ex[4], s4 = 0;
            uint2 four = make_uint2(0, 0);
            if (F & IDS) four = *(const uint2*)(ids + row0 + rl);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t h = hash32(row0 + rl + g);
                if (F & IDS) {
                    const uint32_t id = (((g < 2 ? four.x : four.y) >> (16 * (g & 1))) & 0xFFFFu) % kDictEntries;
                    uint2 ab; __builtin_memcpy(&ab, doff + id, 8);
                    src[g] = ab.x; len[g] = (F & VARLEN) ? ab.y - ab.x : 48u;
                } else {
                    len[g] = (F & VARLEN) ? 20u + h % 61u : 48u;
                    src[g] = (h >> 8) % (dict_span - 128u);
                }
                ex[g] = s4; s4 += len[g];
            }
            const uint32_t inc = wave_incl_scan(s4), first = run + inc - s4;
            run += __shfl(inc, 63);
            *(uint4*)(s_off + rl) = make_uint4(first + ex[0], first + ex[1], first + ex[2], first + ex[3]);
            *(uint4*)(s_dlt + rl) = make_uint4(src[0] - first - ex[0], src[1] - first - ex[1], src[2] - first - ex[2], src[3] - first - ex[3]);
        }
        if (lane == 0) s_off[kPassRows] = run;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        const uint32_t b0 = s_off[0], b1 = s_off[kPassRows];
        for (uint32_t c0 = (b0 & ((F & ALIGN128) ? ~127u : ~15u)) + 16u * lane; c0 - 16u * lane < b1; c0 += 4 * 1024u) {
            uint32_t cp[4], rr[4];
            uint4 w[4];
            // chunk slots this step needs: all four, except in a pass's last step (the same in every lane)
            const uint32_t c0w = __builtin_amdgcn_readfirstlane(c0 - 16u * lane);
            const uint32_t nslot = (F & TRIM) ? min(4u, (b1 - c0w + 1023u) / 1024u) : 4u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if ((F & TRIM) && (uint32_t)u >= nslot) { cp[u] = c0; rr[u] = 0; continue; }
                cp[u] = (c0 + u * 1024u < b1 && c0 + u * 1024u + 16u > b0) ? c0 + u * 1024u : max(c0, b0 & ~15u);
                const uint32_t first = max(cp[u], b0);
                uint32_t r = 0;
                if (F & SEARCH) {
#pragma unroll
                    for (uint32_t step = kPassRows / 2; step >= 1; step >>= 1) {
                        const uint32_t c = r + step, o = s_off[min(c, kPassRows)];
                        if (c < kPassRows && o <= first) r = c;
                    }
                } else {
                    r = min(kPassRows - 1, (first - b0) / 48u);  // rows of 48 bytes in every variant without the search
                }
                rr[u] = r;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if ((F & TRIM) && (uint32_t)u >= nslot) continue;
                // a chunk may start up to 15 bytes before its row does: the position is signed (the dictionaries start 64
                // bytes into their allocation)
                int32_t sp;
                if (F & LDS_DLT) sp = (int32_t)(s_dlt[rr[u]] + cp[u]);
                else sp = (int32_t)((hash32(row0 + rr[u]) >> 8) % (dict_span - 128u) + (cp[u] & 31u));
                sp = min(max(sp, -48), (int32_t)dict_span + 64);  // belt and braces: whatever the tables say, the gather stays inside the allocation
                w[u] = ld16u(dict + 64 + (ptrdiff_t)sp);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c0 + u * 1024u >= b1 || c0 + u * 1024u + 16u <= b0) continue;  // past the pass's end, or (aligned grid) before its start
                uint4 v = w[u];
                const uint32_t cpos = cp[u], cend = min(cpos + 16u, b1);
                if (F & STRADDLE) {
                    uint32_t r = rr[u], rend = s_off[r + 1], pos = min(cend, rend);
                    if (pos < cend) {
                        uint64_t wl = (uint64_t)v.x | ((uint64_t)v.y << 32), wh = (uint64_t)v.z | ((uint64_t)v.w << 32);
                        while (pos < cend) {
                            do { r++; rend = s_off[r + 1]; } while (rend <= pos && r + 1 < kPassRows);
                            const uint32_t segend = min(cend, max(rend, pos + 1));
                            const int32_t xp = min(max((int32_t)(s_dlt[r] + cpos), -48), (int32_t)dict_span + 64);
                            const uint4 x = ld16u(dict + 64 + (ptrdiff_t)xp);
                            const int a = (int)(pos - cpos), e = (int)(segend - cpos);
                            const uint64_t ml = low_bytes(e) & ~low_bytes(a), mh = low_bytes(e - 8) & ~low_bytes(a - 8);
                            wl = (wl & ~ml) | (((uint64_t)x.x | ((uint64_t)x.y << 32)) & ml);
                            wh = (wh & ~mh) | (((uint64_t)x.z | ((uint64_t)x.w << 32)) & mh);
                            pos = segend;
                        }
                        v = make_uint4((uint32_t)wl, (uint32_t)(wl >> 32), (uint32_t)wh, (uint32_t)(wh >> 32));
                    }
                }
                st_nt(obase + min(cpos, 96u * 1024u), v);  // whole chunks everywhere: the edges of a pass overlap by design here
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
    if (F & (PREFK | PREFK4)) {
#pragma unroll
        for (uint32_t k = 0; k < 18; k++) asm volatile("" ::"s"(kv[k]));
    }
    if ((F & PREFL) && (F & LATE)) {
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) asm volatile("" ::"v"(lv[k]));
    }
}


template <int F>
__global__ __launch_bounds__(256, (F & OCC6) ? 6 : 1) void k_tiles(const uint8_t* __restrict__ dicts, const uint32_t* __restrict__ doffs,
                                                                  const uint16_t* __restrict__ ids, uint8_t* __restrict__ out,
                                                                  uint32_t npacks, uint32_t pack_out_stride) {
    // grid (16 or 64, packs); with MANY_DICTS all blocks of a pack share one XCD (linear ids equal mod 8)
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (((F & MANY_DICTS) && !(F & NOREMAP)) || (F & REMAP1)) {
        const uint32_t gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, full = npacks & ~7u;
        if (L < full * gx) { const uint32_t xcd = L & 7u, sidx = L >> 3; by = (sidx / gx) * 8u + ((xcd + sidx / gx) & 7u); bx = sidx % gx; }
    }
    tiles_body<F>(bx, by, gridDim.x, dicts, doffs, ids, out, npacks, pack_out_stride);
}

// Sub-batches: the same tile kernel over packs [pack0, pack0 + gridDim.y), preceded by a kernel that writes (k_fill) or reads
// (k_touch) those packs' dictionaries: does a dictionary that a kernel has just written or read -- tens of megabytes ago, not
// gigabytes -- come out of the memory-side cache when the tile kernel asks for it?
template <int F>
__global__ __launch_bounds__(256, (F & OCC6) ? 6 : 1) void k_tiles_sub(const uint8_t* __restrict__ dicts, const uint32_t* __restrict__ doffs,
                                                                      const uint16_t* __restrict__ ids, uint8_t* __restrict__ out,
                                                                      uint32_t npacks, uint32_t pack_out_stride, uint32_t pack0) {
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    const uint32_t gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x, full = gridDim.y & ~7u;
    if (L < full * gx) { const uint32_t xcd = L & 7u, sidx = L >> 3; by = (sidx / gx) * 8u + ((xcd + sidx / gx) & 7u); bx = sidx % gx; }
    if (pack0 + by >= npacks) return;
    tiles_body<F>(bx, pack0 + by, gridDim.x, dicts, doffs, ids, out, npacks, pack_out_stride);
}
template <bool WRITE>
__global__ __launch_bounds__(256) void k_fill_or_touch(uint8_t* __restrict__ dicts, uint32_t npacks, uint32_t pack0) {
    const uint32_t pack = pack0 + blockIdx.y, j = blockIdx.x;  // grid (16, packs of the sub-batch)
    if (pack >= npacks) return;
    uint8_t* d = dicts + (size_t)pack * (kDictBytes + 256) + 64;
    if (WRITE) {
        const uint32_t share = (kDictBytes / 16u + 15u) & ~15u;
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            const uint32_t o = j * share + (threadIdx.x + i * 256u) * 16u;
            if ((threadIdx.x + i * 256u) * 16u < share && o + 16u <= kDictBytes) *(uint4*)(d + o) = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
        }
    } else {
        const uint32_t lines = (kDictBytes + 127u) >> 7, per = (lines + 15u) / 16u;
        uint32_t v[2];
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t idx = threadIdx.x + i * 256u, line = j * per + idx;
            v[i] = *(const uint32_t*)(d + 128ull * ((idx < per && line < lines) ? line : 0u));
        }
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) asm volatile("" ::"v"(v[i]));
    }
}

// Producer / consumer pipeline in ONE launch: per XCD the blocks come in steps of 16 + 64; the first 16 of step k WRITE the
// dictionary of the XCD's k-th pack (a sixteenth each, the bytes it already holds), publish with a device-scope counter, and
// exit; the other 64 are the 256-row-tile workgroups of the pack of step k - LAG, which wait for that pack's counter to reach
// 16 before they gather.  Workgroups are dispatched in the order of their linear ids, a consumer only waits for producers
// with smaller ids, and producers wait for nothing: the wait ends.  It is bounded all the same (err is set, the tile skipped).
// The same launch shape with WARMERS instead of producers: the first 16 workgroups of step k only READ the dictionary of the
// XCD's k-th pack (one dword per line, a sixteenth each) and exit; nobody waits for anybody (it is a prefetch, not a dependency).
// The cold reads are then paid by workgroups that have nothing else to do, LAG packs ahead of the ones that gather.
template <int F, int LAG, int NW>
__global__ __launch_bounds__(256, 6) void k_warmers(const uint8_t* __restrict__ dicts, const uint32_t* __restrict__ doffs,
                                                    const uint16_t* __restrict__ ids, uint8_t* __restrict__ out, uint32_t npacks,
                                                    uint32_t pack_out_stride) {
    constexpr uint32_t kStep = NW + 64;
    const uint32_t L = blockIdx.x, xcd = L & 7u, s = L >> 3, k = s / kStep, j = s % kStep, npx = npacks >> 3;  // npacks is a multiple of 8
    if (j < (uint32_t)NW) {
        if (k >= npx) return;
        const uint32_t pack = k * 8u + ((xcd + k) & 7u);
        const uint8_t* d = dicts + (size_t)pack * (kDictBytes + 256) + 64;
        const uint32_t lines = (kDictBytes + 127u) >> 7, per = (lines + NW - 1u) / NW;  // lines per warmer
        constexpr uint32_t NL = (4385 / NW + 255) / 256 + 1;
        uint32_t v[NL];
#pragma unroll
        for (uint32_t i = 0; i < NL; i++) {
            const uint32_t idx = threadIdx.x + i * 256u, line = j * per + idx;
            v[i] = *(const uint32_t*)(d + 128ull * ((idx < per && line < lines) ? line : 0u));
        }
#pragma unroll
        for (uint32_t i = 0; i < NL; i++) asm volatile("" ::"v"(v[i]));
        return;
    }
    if (k < (uint32_t)LAG || k - LAG >= npx) return;
    const uint32_t kk = k - LAG, pack = kk * 8u + ((xcd + kk) & 7u);
    tiles_body<F>(j - NW, pack, 64u, dicts, doffs, ids, out, npacks, pack_out_stride);
}

template <int F, int LAG>
__global__ __launch_bounds__(256, 6) void k_pipe(const uint8_t* __restrict__ dicts, const uint32_t* __restrict__ doffs,
                                                 const uint16_t* __restrict__ ids, uint8_t* __restrict__ out, uint32_t npacks,
                                                 uint32_t pack_out_stride, uint32_t* __restrict__ flags, uint32_t* __restrict__ err) {
    const uint32_t L = blockIdx.x, xcd = L & 7u, s = L >> 3, k = s / 80u, j = s % 80u, npx = npacks >> 3;  // npacks is a multiple of 8
    if (j < 16u) {
        if (k >= npx) return;
        const uint32_t pack = k * 8u + ((xcd + k) & 7u);
        uint8_t* d = const_cast<uint8_t*>(dicts) + (size_t)pack * (kDictBytes + 256) + 64;
        const uint32_t share = (kDictBytes / 16u + 15u) & ~15u;  // bytes per producer, a multiple of 16
#pragma unroll
        for (uint32_t i = 0; i < 9; i++) {
            const uint32_t o = j * share + (threadIdx.x + i * 256u) * 16u;
            if ((threadIdx.x + i * 256u) * 16u < share && o + 16u <= kDictBytes) *(uint4*)(d + o) = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
        }
        // Producers and consumers of a pack run on the same XCD, whose L2 is where the stores land (the vector L1 writes through)
        // and where the counter lives: waiting for the stores' acknowledgements orders them before the count.  A device-scope
        // release would write the XCD's whole L2 back (measured: the pipeline then takes 4.2 ms).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + pack, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (k < (uint32_t)LAG || k - LAG >= npx) return;
    const uint32_t kk = k - LAG, pack = kk * 8u + ((xcd + kk) & 7u);
    __shared__ uint32_t ready;
    if (threadIdx.x == 0) {
        uint32_t spins = 0, v;
        while ((v = __hip_atomic_load(flags + pack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 16u && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(8);
        ready = v >= 16u;
        if (v < 16u) atomicAdd(err, 1u);
    }
    __syncthreads();
    if (!ready) return;
    tiles_body<F>(j - 16u, pack, 64u, dicts, doffs, ids, out, npacks, pack_out_stride);
}

int main() {
    const uint32_t npacks = 1024;  // x 64 tiles x 1024 rows = 64 M rows, ~3.3 GB of rows
    const size_t dict_total = (size_t)npacks * (kDictBytes + 256) + (2u << 20);
    uint8_t* dicts; CK(hipMalloc(&dicts, dict_total)); CK(hipMemset(dicts, 7, dict_total));
    std::vector<uint32_t> hdoff((size_t)npacks * (kDictEntries + 8));
    for (uint32_t p = 0; p < npacks; p++) {
        uint32_t pos = 0;
        for (uint32_t i = 0; i <= kDictEntries + 7; i++) { hdoff[(size_t)p * (kDictEntries + 8) + i] = pos; if (i < kDictEntries) pos += 20 + (i * 2654435761u >> 7) % 61; }
    }
    uint32_t* doffs; CK(hipMalloc(&doffs, hdoff.size() * 4)); CK(hipMemcpy(doffs, hdoff.data(), hdoff.size() * 4, hipMemcpyHostToDevice));
    const size_t nrows = (size_t)npacks * 65536;
    std::vector<uint16_t> hids(nrows + 64);
    uint64_t x = 88172645463325252ull;
    for (auto& v : hids) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint16_t)(x % kDictEntries); }
    uint16_t* ids; CK(hipMalloc(&ids, hids.size() * 2)); CK(hipMemcpy(ids, hids.data(), hids.size() * 2, hipMemcpyHostToDevice));
    // the same number of ids drawn from a Zipf(1.1) law over the entries, the hot entries scattered over the dictionary
    {
        std::vector<double> cdf(kDictEntries);
        double acc = 0;
        for (uint32_t k = 0; k < kDictEntries; k++) { acc += std::pow((double)(k + 1), -1.1); cdf[k] = acc; }
        std::vector<uint16_t> perm(kDictEntries);
        for (uint32_t k = 0; k < kDictEntries; k++) perm[k] = (uint16_t)k;
        for (uint32_t k = kDictEntries - 1; k > 0; k--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(perm[k], perm[x % (k + 1)]); }
        for (auto& v : hids) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const double u = (double)(x >> 11) / 9007199254740992.0 * acc;
            v = perm[(size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) % kDictEntries];
        }
    }
    uint16_t* zids; CK(hipMalloc(&zids, hids.size() * 2)); CK(hipMemcpy(zids, hids.data(), hids.size() * 2, hipMemcpyHostToDevice));
    const uint32_t pack_out_stride = 256 * (256 * 50 + 1024) + 65536;  // room for either tiling: 64 x 52224 or 256 x 13824 bytes, multiples of 128
    const size_t out_bytes = (size_t)npacks * pack_out_stride + (1u << 20);
    uint8_t* out; CK(hipMalloc(&out, out_bytes)); CK(hipMemset(out, 0, out_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto fn) {
        fn(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
            (void)hipEventRecord(e0); fn(); (void)hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s failed\n", name); exit(1); }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-86s %8.3f ms  %8.1f GB/s of rows\n", name, best, bytes / 1e9 / (best / 1e3));
    };
    const double fixed_bytes = (double)nrows * 48, var_bytes = (double)nrows * 50;
    uint32_t* flags; CK(hipMalloc(&flags, (npacks + 1) * 4)); CK(hipMemset(flags, 0, (npacks + 1) * 4));
    timeit("workgroup owns 64 KB: 48-byte rows, position hashed, one 1 MB dictionary", 65536.0 * 50000,
           [&] { hipLaunchKernelGGL(k_block, dim3(50000), dim3(256), 0, 0, dicts, out, 1u << 20); });
#define RUN(F, bytes, name) timeit(name, bytes, [&] { hipLaunchKernelGGL((k_tiles<(F)>), dim3(((F) & TILE256) ? 64 : 16, npacks), dim3(256), 0, 0, dicts, doffs, ((F) & ZIPF) ? zids : ids, out, npacks, pack_out_stride); })
    RUN(WAVE_TILES, fixed_bytes, "wave owns a 1024-row tile (48-byte rows, row by division, position hashed)");
    RUN(WAVE_TILES | LDS_DLT, fixed_bytes, "+ source position from the LDS table");
    RUN(WAVE_TILES | LDS_DLT | SEARCH, fixed_bytes, "+ 9-step row search in LDS");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6, fixed_bytes, "+ 6 waves per SIMD at most (the kernel's launch bound)");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN, var_bytes, "+ rows of 20..80 bytes (straddling chunks take their first row's bytes only)");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | STRADDLE, var_bytes, "+ the straddle loop (gathers of the following rows, byte masks)");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | STRADDLE | MANY_DICTS, var_bytes, "+ one 548 KB dictionary per pack, a pack's blocks on one XCD");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | STRADDLE | MANY_DICTS | IDS, var_bytes, "+ positions from doff[id], ids streamed (2 bytes a row)");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | STRADDLE | IDS, var_bytes, "  the same with ONE dictionary for all packs");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | MANY_DICTS | IDS, var_bytes, "  the same without the straddle loop");
    RUN(WAVE_TILES | LDS_DLT | SEARCH | VARLEN | STRADDLE | MANY_DICTS | IDS, var_bytes, "  the same without the occupancy bound");
    constexpr int FULL = WAVE_TILES | LDS_DLT | SEARCH | OCC6 | VARLEN | STRADDLE | MANY_DICTS | IDS;
    printf("-- candidate remedies on the full loop (first line: the full loop again)\n");
    RUN(FULL, var_bytes, "full loop, uniform ids");
    RUN(FULL | ZIPF, var_bytes, "full loop, Zipf(1.1) ids");
    RUN((FULL & ~MANY_DICTS) | ZIPF, var_bytes, "  Zipf ids, ONE dictionary for all packs");
    RUN(FULL | ZIPF | TRIM, var_bytes, "  Zipf ids + the last step of a pass skips idle chunk slots");
    RUN(FULL | ZIPF | ALIGN128, var_bytes, "  Zipf ids + chunk grid aligned to 128-byte lines");
    RUN(FULL | ZIPF | TRIM | ALIGN128, var_bytes, "  Zipf ids + both");
    RUN(FULL | ZIPF | TILE256, var_bytes, "  Zipf ids + a wave owns 256 rows (a quarter of the packs in flight)");
    RUN(FULL | ZIPF | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned grid");
    RUN(FULL | ZIPF | PREFV, var_bytes, "  Zipf ids + every workgroup streams 1/16 of the pack's dictionary into L2 first");
    RUN(FULL | ZIPF | PREFL, var_bytes, "  Zipf ids + the same warm-up with one dword load per 128-byte line");
    RUN(FULL | ZIPF | PREFL64, var_bytes, "  Zipf ids + warm-up with one dword load per 64 bytes");
    RUN(FULL | ZIPF | PREFL | TILE256, var_bytes, "  Zipf ids + line warm-up + 256-row tiles");
    RUN(FULL | PREFL | TILE256, var_bytes, "  uniform ids + line warm-up + 256-row tiles");
    RUN(FULL | ZIPF | PREFA1 | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + dictionary of the NEXT pack of the XCD warmed");
    RUN(FULL | ZIPF | PREFA2 | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + dictionary two packs ahead warmed");
    RUN(FULL | ZIPF | PREFA3 | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + dictionary three packs ahead warmed");
    RUN(FULL | PREFA2 | TILE256 | TRIM | ALIGN128, var_bytes, "  uniform ids + 256-row tiles + trim + aligned + dictionary two packs ahead warmed");
    RUN(FULL | ZIPF | PREFA2 | PREFS | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + two ahead AND own (four spread shares)");
    RUN(FULL | ZIPF | PREFA1, var_bytes, "  Zipf ids, 1024-row tiles + dictionary of the next pack warmed");
    RUN(FULL | ZIPF | PREFK | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + SCALAR-load warm-up of the own share");
    RUN(FULL | PREFK | TILE256 | TRIM | ALIGN128, var_bytes, "  uniform ids + 256-row tiles + trim + aligned + scalar-load warm-up of the own share");
    RUN(FULL | ZIPF | PREFK4 | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + scalar-load warm-up, 18 lines of four spread shares");
    RUN(FULL | ZIPF | PREFST | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned + the share of the dictionary STORED first");
    RUN(FULL | PREFST | TILE256 | TRIM | ALIGN128, var_bytes, "  uniform ids + 256-row tiles + trim + aligned + the share of the dictionary stored first");
    RUN(FULL | ZIPF | FEWDICTS | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + trim + aligned, FOUR dictionaries in all (per-pack addressing)");
    RUN(FULL | ZIPF | FEWDICTS, var_bytes, "  Zipf ids, 1024-row tiles, four dictionaries in all");
#define RUNP(F, LAG, name) timeit(name, var_bytes, [&] { (void)hipMemsetAsync(flags, 0, npacks * 4, 0); hipLaunchKernelGGL((k_pipe<(F), LAG>), dim3(8u * (npacks / 8u + LAG) * 80u), dim3(256), 0, 0, dicts, doffs, zids, out, npacks, pack_out_stride, flags, flags + npacks); })
    RUNP(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 1, "  PIPELINE: 16 producers store a pack's dictionary, its 64 tile workgroups follow one pack later");
    RUNP(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 2, "  pipeline, consumers two packs behind");
    RUNP(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 3, "  pipeline, consumers three packs behind");
    RUNP(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 0, "  pipeline, consumers of the same pack right behind its producers");
    { uint32_t e = 0; CK(hipMemcpy(&e, flags + npacks, 4, hipMemcpyDeviceToHost)); printf("  pipeline: %u workgroups gave up waiting\n", e); }
    RUN(FULL | ZIPF | PREFS, var_bytes, "  Zipf ids, 1024-row tiles + warm-up of four spread shares (a quarter of the dictionary per workgroup)");
    RUN(FULL | ZIPF | PREFW, var_bytes, "  Zipf ids, 1024-row tiles + warm-up of four consecutive shares");
    RUN(FULL | ZIPF | PREFS | TRIM | ALIGN128, var_bytes, "  Zipf ids, 1024-row tiles + four spread shares + trim + aligned grid");
    RUN(FULL | ZIPF | PREFL | TRIM | ALIGN128, var_bytes, "  Zipf ids, 1024-row tiles + line warm-up + trim + aligned grid");
    RUN(FULL | ZIPF | FEWDICTS | PREFL, var_bytes, "  Zipf ids, 1024-row tiles, four dictionaries in all + line warm-up (of resident lines)");
    RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM, var_bytes, "  Zipf ids, 1024-row tiles, four dictionaries in all + line warm-up reading the COLD per-pack dictionaries");
    RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM | TILE256 | TRIM | ALIGN128, var_bytes, "  the same with 256-row tiles + trim + aligned grid");
    RUN(FULL | ZIPF | PREFL | LATE, var_bytes, "  Zipf ids, 1024-row tiles + line warm-up nobody waits for separately");
    RUN(FULL | PREFL | LATE, var_bytes, "  uniform ids, 1024-row tiles + line warm-up nobody waits for separately");
    RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM | LATE, var_bytes, "  Zipf ids, four dictionaries in all + cold warm-up loads nobody waits for separately");
    RUN(FULL | ZIPF | PREFL | LATE | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids, 256-row tiles + trim + aligned + line warm-up (own share) nobody waits for separately");
    for (uint32_t region : {2u << 20, 16u << 20, 64u << 20, 256u << 20, 512u << 20}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_region), &region, 4));
        char nm[160]; snprintf(nm, sizeof nm, "  four dictionaries in all + warm-up loads spread over a region of %u MB", region >> 20);
        RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM, var_bytes, nm);
    }
    for (uint32_t region : {96u << 20, 128u << 20, 192u << 20}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_region), &region, 4));
        char nm[160]; snprintf(nm, sizeof nm, "  four dictionaries in all + warm-up loads spread over a region of %u MB", region >> 20);
        RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM, var_bytes, nm);
    }
    for (uint32_t spread : {2u, 4u, 8u}) {
        const uint32_t region = 60u << 20;  // 60 MB of lines (the spare 2 MB at the end absorb a share that starts at the region's last line)
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_region), &region, 4));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_spread), &spread, 4));
        char nm[160]; snprintf(nm, sizeof nm, "  four dictionaries in all + warm-up loads: 60 MB of lines, %u x 128 bytes apart (span %u MB)", spread, 60 * spread);
        RUN(FULL | ZIPF | FEWDICTS | PREFL | COLDWARM, var_bytes, nm);
    }
    { uint32_t zero = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_region), &zero, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_cold_spread), &zero, 4)); }
#define RUNW(F, LAG, NW, name) timeit(name, var_bytes, [&] { hipLaunchKernelGGL((k_warmers<(F), LAG, NW>), dim3(8u * (npacks / 8u + LAG) * (NW + 64u)), dim3(256), 0, 0, dicts, doffs, zids, out, npacks, pack_out_stride); })
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 1, 16, "  WARMERS: 16 workgroups read a pack's dictionary and exit, its 64 tile workgroups follow one pack later");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 2, 16, "  warmers, tile workgroups two packs behind");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 3, 16, "  warmers, tile workgroups three packs behind");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 2, 4, "  4 warmers per pack, tile workgroups two packs behind");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 2, 2, "  2 warmers per pack, tile workgroups two packs behind");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 4, 2, "  2 warmers per pack, tile workgroups four packs behind");
    RUNW(FULL | ZIPF | TILE256 | TRIM | ALIGN128, 0, 16, "  warmers right before the pack's own tile workgroups");
    for (uint32_t sb : {64u, 128u, 256u}) {
        for (int mode = 0; mode < 3; mode++) {
            char nm[200]; snprintf(nm, sizeof nm, "  sub-batches of %u packs (1024-row tiles, line warm-up)%s", sb,
                                   mode == 0 ? "" : (mode == 1 ? ", each preceded by a kernel that WRITES its dictionaries" : ", each preceded by a kernel that READS its dictionaries"));
            timeit(nm, var_bytes, [&] {
                for (uint32_t p0 = 0; p0 < npacks; p0 += sb) {
                    if (mode == 1) hipLaunchKernelGGL((k_fill_or_touch<true>), dim3(16, sb), dim3(256), 0, 0, dicts, npacks, p0);
                    if (mode == 2) hipLaunchKernelGGL((k_fill_or_touch<false>), dim3(16, sb), dim3(256), 0, 0, dicts, npacks, p0);
                    hipLaunchKernelGGL((k_tiles_sub<FULL | ZIPF | PREFL>), dim3(16, sb), dim3(256), 0, 0, dicts, doffs, zids, out, npacks, pack_out_stride, p0);
                }
            });
        }
    }
    RUN(FULL | ZIPF | PREFW | TILE256, var_bytes, "  Zipf ids + 256-row tiles + warm-up of four consecutive shares per workgroup");
    RUN(FULL | ZIPF | PREFS | TILE256, var_bytes, "  Zipf ids + 256-row tiles + warm-up of four spread shares per workgroup");
    RUN(FULL | PREFS | TILE256, var_bytes, "  uniform ids + 256-row tiles + warm-up of four spread shares");
    RUN(FULL | ZIPF | PREFS | TILE256 | TRIM | ALIGN128, var_bytes, "  Zipf ids + 256-row tiles + four spread shares + trim + aligned grid");
    RUN(FULL | PREFL, var_bytes, "  uniform ids + line warm-up");
    RUN(FULL | ZIPF | PREFL | LDS40, var_bytes, "  Zipf ids + line warm-up, 16 waves per compute unit (8 packs in flight per XCD)");
    RUN(FULL | ZIPF | PREFL | LDS80, var_bytes, "  Zipf ids + line warm-up, 8 waves per compute unit (4 packs in flight per XCD)");
    RUN(FULL | PREFL | LDS40, var_bytes, "  uniform ids + line warm-up, 16 waves per compute unit");
    RUN(FULL | PREFV, var_bytes, "  uniform ids + dictionary streamed into L2 first");
    RUN(FULL | ZIPF | PREFV | TILE256, var_bytes, "  Zipf ids + 256-row tiles + dictionary streamed into L2 first (1/64 per workgroup)");
    RUN(FULL | ZIPF | NOREMAP, var_bytes, "  Zipf ids, per-pack dictionaries WITHOUT the XCD relabelling");
    RUN((FULL & ~MANY_DICTS) | ZIPF | REMAP1, var_bytes, "  Zipf ids, ONE dictionary WITH the XCD relabelling");
    RUN(FULL | ZIPF | ONE_DOFF, var_bytes, "  Zipf ids, per-pack dictionaries, ONE offset table for all packs");
    RUN((FULL & ~STRADDLE) | ZIPF | PREFV, var_bytes, "  Zipf ids + dictionary streamed first, without the straddle loop");
    RUN(FULL | TILE256, var_bytes, "  uniform ids + a wave owns 256 rows");
    RUN(FULL | TILE256 | TRIM | ALIGN128, var_bytes, "  uniform ids + 256-row tiles + trim + aligned grid");
    return 0;
}
