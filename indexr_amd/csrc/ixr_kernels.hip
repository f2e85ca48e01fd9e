// ixr_kernels.hip -- CDNA4 (gfx950) kernels decoding IndexR VLT packs.  Wave size is 64 throughout.
//
// Wire format decoded here (all little-endian), with the reference routine each kernel replaces:
//   bit packing     value i of a run sits at bit i*bw of the byte string         IntPacker.java:54-70, LongPacker.java:54-71
//   DELTA numbers   u32 len | zz-varint first | per 128: zz-varint minDelta,    DeltaCodec.java:100-141, 204-245
//                   per 32: u8 bw + ceil(c/8)*bw bytes ; v = prev + minDelta + packed (wrapping)
//   RLE hybrid ids  u32 len | { uvarint h ; h&1 ? (h>>1) packed values : run of (h>>1) x fixed-width value }*
//                                                                                RLEPackingHybridCodec.java:97-128
//   DICT_COMPRESS   u32 dictStructSize | u32 entryCount | u8 compressed | entries | ids   DictStructUtil.java:83-142
//   SIMPLE_DICT     u32 entryCount | plain entries | u32 dataSize | ids                   SimpleDictCodec.java:131-159 ...
//   DELTA strings   u32 len | DELTA(prefixLen) | DELTA(suffixLen) | suffix bytes          DeltaCodec.java:355-388
//   decoded strings int32 offset[n+1] | bodies                                            StringsStruct.java:12-17
// (paths relative to indexr-vlt-segment/src/main/java/io/indexr/vlt/codec/ and indexr-common/.../data/ of the reference)
//
// Pipeline per group of packs (one launch each, thousands of packs per launch):
//   k_parse      one wave per serial stream: walks the data-dependent framing (mini-block widths, RLE headers) through an
//                LDS window, validates it and leaves an index: per mini-block (offset,width), per block minDelta, per
//                1024-row tile the RLE segment covering its first row.
//   k_delta      one workgroup per DELTA stream: bit-unpack + wave/workgroup prefix sum with a running carry.
//   k_dict_num   one wave per 1024-row tile: expand ids from the tile's segment record, gather dictionary entries, store rows.
//   string packs k_str_dict_scan -> k_front_decode -> k_str_bases -> k_str_fused  (see each kernel; k_str_index -> k_str_copy
//                is the two-kernel form of the last two)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <type_traits>

#include "ixr_device.h"
#include "ixr_launch.h"

namespace ixr {

// error codes (mirror include/ixr_gpu.h)
constexpr int32_t E_OVERFLOW = -1, E_CORRUPT = -2, E_ILLEGAL_ENC = -101, E_UNSUPPORTED_TYPE = -102;

// ---------------------------------------------------------------------------------------------------------------
// unaligned little-endian access built from aligned dword loads (the arena is readable 16 bytes past every pack)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    uint32_t sh = (uint32_t)(a & 3) << 3;
    uint64_t w = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
    return (uint32_t)(w >> sh);
}

// bits [bit, bit+bw) of the little-endian bit string starting at `base`; bw in 0..32
__device__ __forceinline__ uint32_t extract32(const uint8_t* base, uint32_t bit, uint32_t bw) {
    uintptr_t a = (uintptr_t)(base + (bit >> 3));
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    uint32_t sh = ((uint32_t)(a & 3) << 3) + (bit & 7);  // 0..31
    uint64_t w = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
    uint32_t v = (uint32_t)(w >> sh);
    return bw >= 32 ? v : (v & ((1u << bw) - 1u));
}

// bw in 0..64
__device__ __forceinline__ uint64_t extract64(const uint8_t* base, uint32_t bit, uint32_t bw) {
    uintptr_t a = (uintptr_t)(base + (bit >> 3));
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    uint32_t sh = ((uint32_t)(a & 3) << 3) + (bit & 7);  // 0..31
    uint64_t lo = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
    uint64_t hi = q[2];
    uint64_t v = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    return bw >= 64 ? v : (v & ((1ull << bw) - 1ull));
}

template <typename U> __device__ __forceinline__ U extract(const uint8_t* base, uint32_t bit, uint32_t bw);
template <> __device__ __forceinline__ uint32_t extract<uint32_t>(const uint8_t* base, uint32_t bit, uint32_t bw) { return extract32(base, bit, bw); }
template <> __device__ __forceinline__ uint64_t extract<uint64_t>(const uint8_t* base, uint32_t bit, uint32_t bw) { return extract64(base, bit, bw); }

__device__ __forceinline__ uint32_t bit_width32(uint32_t v) { return v ? 32u - (uint32_t)__clz((int)v) : 0u; }

// ---------------------------------------------------------------------------------------------------------------
// Stream window: a wave keeps kWin bytes of a pack in LDS and reads the data-dependent framing (varints, width bytes, RLE
// headers) from there.  One coalesced global load fills the window; every step of a serial walk then costs an LDS round
// trip instead of a trip to L2/HBM.  All lanes of the wave call these functions with the same arguments.
// ---------------------------------------------------------------------------------------------------------------
#ifndef IXR_FRONT_BATCH
#define IXR_FRONT_BATCH 8
#endif
#ifndef IXR_FUSED_CHUNKS
#define IXR_FUSED_CHUNKS 4      // 16-byte chunks per lane and step of k_str_fused
#endif
constexpr int kParseWaveThreshold = 2048;      // packs per launch from which every stream gets a workgroup of one wave (k_parse_waves)
constexpr uint32_t kWin = 4096;
constexpr uint32_t kWinAlloc = kWin + 16;  // bytes of LDS per window: data and 16 spare bytes so that 12-byte peeks never run off

// A wave-uniform 64-bit value, known to the compiler as such (kept in scalar registers).
__device__ __forceinline__ uint64_t uniform64(uint64_t a) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
}
// Loads *p, where p is the same in every lane, through the scalar cache (constant address space + uniform address).
// Only for memory that no kernel writes while this one runs.
template <class T>
__device__ __forceinline__ T uniform_load(const T* p) {
    typedef const T __attribute__((address_space(4))) * kp;
    return *(kp)uniform64((uint64_t)p);
}
template <>
__device__ __forceinline__ uint4 uniform_load<uint4>(const uint4* p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(4))) * kp;
    const u32x4 v = *(kp)uniform64((uint64_t)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------------------------
// XCD-aware block relabelling for (blocks-per-pack, packs) grids.  Workgroups are dealt round-robin over the 8 XCDs by
// linear id, which spreads the blocks of one pack -- and with them its dictionary -- over all 8 L2s.  Relabel so that all
// blocks of a pack have linear ids that are equal mod 8: one pack, one L2.  Placement only affects speed.
// ---------------------------------------------------------------------------------------------------------------
// Which of its 8 packs an XCD takes changes from one group of 8 packs to the next: plans list their packs column by
// column within a pack id, so with 4 or 8 columns a fixed slot would hand all packs of one column -- the cheap or the
// expensive ones -- to the same one or two XCDs.
__device__ __forceinline__ uint32_t xcd_slot(uint32_t group, uint32_t xcd) {
    return (xcd + group + (group >> 3) + (group >> 6)) & 7u;
}
__device__ __forceinline__ void pack_block_remap(uint32_t& bx, uint32_t& by) {
    const uint32_t gx = gridDim.x, count = gridDim.y;
    const uint32_t L = blockIdx.y * gx + blockIdx.x;
    const uint32_t full = count & ~7u;
    if (L < full * gx) {
        const uint32_t xcd = L & 7u, sidx = L >> 3;
        by = (sidx / gx) * 8u + xcd_slot(sidx / gx, xcd);
        bx = sidx % gx;
    } else {
        by = blockIdx.y;
        bx = blockIdx.x;
    }
}
// The same for grids of one workgroup per pack: the list position workgroup `b` of `count` takes.
__device__ __forceinline__ uint32_t pack_slot_remap(uint32_t b, uint32_t count) {
    return b < (count & ~7u) ? (b & ~7u) + xcd_slot(b >> 3, b & 7u) : b;
}


// ---------------------------------------------------------------------------------------------------------------
// Warming the XCD's L2 with a pack's dictionary.  The row kernels gather from the decoded dictionary in scratch, which an
// earlier kernel wrote and which has left the caches since (a group's dictionaries are a gigabyte).  The vector L1 of a
// compute unit returns its loads in order: a gather that has to go to memory holds back every cache hit queued behind it,
// so it pays to have the lines in L2 before the gathers ask for them.  All workgroups of a pack run on one XCD
// (pack_block_remap); each touches ONE dword of every 128-byte line of its share of the dictionary when it starts, with all
// loads issued before anything waits, and throws the data away.  Measured on the string kernel's loop (tools/ubench_bridge.hip,
// Zipf-distributed ids, 548 KB dictionaries): 1.90 -> 1.55 ms; one 16-byte load per lane and piece instead: 1.65.
// ---------------------------------------------------------------------------------------------------------------
#ifdef IXR_NO_WARM  // build switch for A/B timing
constexpr bool kWarm = false;
#else
constexpr bool kWarm = true;
#endif
constexpr uint32_t kWarmLoads = 3;            // per lane: 256 lanes x 3 lines x 128 bytes = 96 KB per workgroup, 1.5 MB per pack of 16 workgroups
constexpr uint32_t kWarmMinBytes = 32768;     // smaller dictionaries come along with the first gathers
struct Warm { uint32_t v[kWarmLoads]; };
// [d0, d0 + bytes) must be readable up to the next multiple of 4 (d0 is 4-byte aligned); `part` of `parts` is this workgroup's
// share.  A dictionary whose entries are used about once each (as many entries as rows: unique values, sorted columns) is
// not warmed: every line is fetched once either way, and the extra loads only cost (measured: unique strings 1.39 -> 1.46 ms).
__device__ __forceinline__ Warm warm_issue(const uint8_t* d0, uint32_t bytes, uint32_t part, uint32_t parts, uint32_t entries, uint32_t rows) {
    Warm w;
#pragma unroll
    for (uint32_t k = 0; k < kWarmLoads; k++) w.v[k] = 0;
    if (!kWarm || bytes < kWarmMinBytes || 2u * rows < 3u * entries) return w;
    const uint32_t lines = (bytes + 127u) >> 7, per = (lines + parts - 1u) / parts;
#pragma unroll
    for (uint32_t k = 0; k < kWarmLoads; k++) {
        const uint32_t idx = threadIdx.x + k * 256u, line = part * per + idx;
        if (idx < per && line < lines) w.v[k] = *(const uint32_t*)(d0 + 128ull * line);  // 128 * line <= bytes - 1: at most 3 bytes past the end
    }
    return w;
}
__device__ __forceinline__ void warm_done(const Warm& w) {  // the loads must be kept although nothing reads their data
#pragma unroll
    for (uint32_t k = 0; k < kWarmLoads; k++) asm volatile("" ::"v"(w.v[k]));
}

template <uint32_t kW>
struct WinT {
    static constexpr uint32_t kBytes = kW;                    // bytes of the pack the window can hold
    static constexpr uint32_t kLoad = kW < 4096u ? kW : 4096u; // bytes a plain load() brings in
    static constexpr uint32_t kAlloc = kW + 16;               // bytes of LDS: data, 16 spare bytes
    const uint8_t* base;  // pack start in global memory, 16-byte aligned
    uint32_t limit16;     // bytes readable from `base` in whole 16-byte chunks (pack + arena slack, rounded down)
    uint8_t* lds;         // kAlloc bytes, 16-byte aligned
    uint32_t w0;          // pack offset held by lds[0]; the window is [w0, w0 + wlen)
    uint32_t wlen;        // bytes of the window that are loaded (kLoad after load(), up to kW after load_n())

    __device__ __forceinline__ void init(const uint8_t* b, uint32_t pack_size, uint8_t* l) {
        base = b; limit16 = __builtin_amdgcn_readfirstlane((pack_size + kInSlack) & ~15u); lds = l; w0 = 0xFFFFFFFFu; wlen = kLoad;
    }
    // LDS instructions of one wave execute in program order, so ordering window writes against window reads needs a
    // fence for the compiler only, and only over LDS (a fence that covers global memory as well can make the wave wait
    // for the stores it has in flight).
    static __device__ __forceinline__ void wave_lds_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void load(uint32_t pos) {
        const int lane = threadIdx.x & 63;
        wave_lds_sync();
        w0 = __builtin_amdgcn_readfirstlane(pos & ~15u);
        wlen = kLoad;
        // all chunk loads are issued before the first is consumed: out-of-range chunks re-read the last valid one
        // (loading only the kilobytes a tile is expected to need was measured: 3 to 15 % slower, the chunk loads then sit
        // behind branches)
        uint4 v[kLoad / 1024];
#pragma unroll
        for (uint32_t i = 0; i < kLoad / 1024; i++) {
            const uint32_t off = w0 + i * 1024 + lane * 16;  // multiple of 16, so an in-range chunk ends <= limit16
            v[i] = *(const uint4*)(base + min(off, limit16 - 16u));
            if (off >= limit16) v[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t i = 0; i < kLoad / 1024; i++) *(uint4*)(lds + i * 1024 + lane * 16) = v[i];
        if (lane == 0) *(uint4*)(lds + kLoad) = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
    }
    // a longer stretch: [pos, pos + bytes) and at least kLoad bytes, at most kW, in kilobytes; eight loads per lane in flight
    __device__ __forceinline__ void load_n(uint32_t pos, uint32_t bytes) {
        const int lane = threadIdx.x & 63;
        wave_lds_sync();
        w0 = __builtin_amdgcn_readfirstlane(pos & ~15u);
        const uint32_t nk = __builtin_amdgcn_readfirstlane(min(kW / 1024u, max(kLoad / 1024u, (bytes + 15u + 1023u) / 1024u)));
        wlen = nk * 1024u;
        for (uint32_t k0 = 0; k0 < nk; k0 += 8) {
            uint4 v[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                const uint32_t off = w0 + (k0 + i) * 1024 + lane * 16;
                v[i] = *(const uint4*)(base + min(off, limit16 - 16u));
                if (off >= limit16) v[i] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (uint32_t i = 0; i < 8; i++)
                if (k0 + i < nk) *(uint4*)(lds + (k0 + i) * 1024 + lane * 16) = v[i];
        }
        if (lane == 0) *(uint4*)(lds + wlen) = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
    }
    __device__ __forceinline__ uint32_t len() const {
        if constexpr (kW == kLoad) return kW; else return wlen;
    }
    __device__ __forceinline__ bool has(uint32_t pos, uint32_t bytes) const { return pos >= w0 && pos + bytes <= w0 + len(); }
    // make [pos, pos + bytes) part of the window (bytes <= kLoad - 16)
    __device__ __forceinline__ void need(uint32_t pos, uint32_t bytes) {
        if (pos < w0 || pos + bytes > w0 + len()) load(pos);
    }
    // The walkers below read the same address in every lane; readfirstlane tells the compiler so, which keeps the whole
    // chain (positions, widths, branches) in scalar registers and scalar branches.
    __device__ __forceinline__ uint32_t u8(uint32_t pos) const {  // pos must be in the window
        return __builtin_amdgcn_readfirstlane((uint32_t)lds[pos - w0]);
    }
    // little-endian 64 bits at pack offset `pos`; [pos, pos + 8) must be in the window
    __device__ __forceinline__ uint64_t peek64(uint32_t pos) const {
        const uint32_t idx = pos - w0, a = idx & ~3u, sh = (idx & 3u) * 8u;
        const uint32_t* q = (const uint32_t*)(lds + a);
        const uint64_t lo = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
        return sh ? ((lo >> sh) | ((uint64_t)q[2] << (64 - sh))) : lo;
    }
    // the same for a position that is equal in all lanes: the three dwords go to scalar registers first, the shifts
    // that assemble the result are scalar
    __device__ __forceinline__ uint64_t peek64u(uint32_t pos) const {
        const uint32_t idx = pos - w0, a = idx & ~3u, sh = (idx & 3u) * 8u;
        const uint32_t* q = (const uint32_t*)(lds + a);
        const uint32_t q0 = __builtin_amdgcn_readfirstlane(q[0]), q1 = __builtin_amdgcn_readfirstlane(q[1]);
        const uint32_t q2 = __builtin_amdgcn_readfirstlane(q[2]);
        const uint64_t lo = (uint64_t)q0 | ((uint64_t)q1 << 32);
        return sh ? ((lo >> sh) | ((uint64_t)q2 << (64 - sh))) : lo;
    }
    // 4 bytes at a position that is equal in all lanes, in a scalar register; [pos, pos + 4) must be in the window
    __device__ __forceinline__ uint32_t peek32u(uint32_t pos) const {
        const uint32_t idx = pos - w0, a = idx & ~3u, sh = (idx & 3u) * 8u;
        const uint32_t* q = (const uint32_t*)(lds + a);
        const uint32_t q0 = __builtin_amdgcn_readfirstlane(q[0]), q1 = __builtin_amdgcn_readfirstlane(q[1]);
        return (uint32_t)(((uint64_t)q0 | ((uint64_t)q1 << 32)) >> sh);
    }
    // bits [bit, bit + bw) of the bit string starting at pack offset `pos`, bw <= 25; the bytes must be in the window
    __device__ __forceinline__ uint32_t bits(uint32_t pos, uint32_t bit, uint32_t bw) const {
        const uint32_t idx = pos - w0 + (bit >> 3), a = idx & ~3u, sh = (idx & 3u) * 8u + (bit & 7u);
        const uint32_t* q = (const uint32_t*)(lds + a);
        const uint64_t w = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
        return (uint32_t)(w >> sh) & ((1u << bw) - 1u);
    }
};
using Win = WinT<kWin>;  // the window of every kernel but the fused string kernel


// VarInteger.readUnsignedVarInt / readUnsignedVarLong (VarInteger.java:68-96) out of the window: at most `max_bytes`
// (5 or 10) bytes, none at or beyond `end`.  Returns the value, advances pos; sets err when the varint runs off the stream
// or does not terminate.
template <class WinX>
__device__ __forceinline__ uint64_t win_uvarint(WinX& w, uint32_t& pos, uint32_t end, int max_bytes, int32_t& err) {
    if (pos >= end) { err = E_CORRUPT; return 0; }
    w.need(pos, 16);
    const uint64_t lo = w.peek64u(pos);
    if (!(lo & 0x80u)) { pos++; return lo & 0x7Fu; }  // one-byte varint: the common case
    const uint64_t hi = w.peek64u(pos + 8);  // bytes 8 and 9 of a long varint
    const uint32_t avail = min((uint32_t)max_bytes, end - pos);
    uint64_t v = 0;
    uint32_t used = 0;
    bool open = true, terminated = false;
#pragma unroll
    for (uint32_t i = 0; i < 10; i++) {
        const uint32_t b = (uint32_t)((i < 8 ? lo : hi) >> (8 * (i & 7))) & 0xFFu;
        const bool take = open && i < avail;
        if (take) { v |= (uint64_t)(b & 0x7F) << (7 * i); used = i + 1; terminated = !(b & 0x80u); }
        open = take && (b & 0x80u);
    }
    pos += used;
    if (!terminated) err = E_CORRUPT;  // ran into the end of the stream, or max_bytes continuation bytes
    return v;
}

// VarInteger.readZigZagVarInt / readZigZagVarLong (VarInteger.java:106-127)
__device__ __forceinline__ int64_t unzigzag32(uint64_t raw) {
    uint32_t r = (uint32_t)raw;
    return (int64_t)(int32_t)((r >> 1) ^ (0u - (r & 1u)));
}
__device__ __forceinline__ int64_t unzigzag64(uint64_t r) { return (int64_t)((r >> 1) ^ (0ull - (r & 1ull))); }

// ---------------------------------------------------------------------------------------------------------------
// k_parse: serial framing walk.  One workgroup per pack, one wave per serial stream (wave 0 and 1: the DELTA streams,
// wave 2: the RLE id stream); each wave follows its chain through an LDS window.
// ---------------------------------------------------------------------------------------------------------------
// Walk one DELTA stream (DeltaCodec.decodeInts/decodeLongs framing, DeltaCodec.java:100-141, 204-245) and write its index.
// Where a block starts depends on every width byte before it, and a single wave issues one instruction every few cycles,
// so the serial part does the least it can: it follows the chain (length of the block's varint, four width bytes) and notes
// each block's position and widths in LDS (`s_blk`, one uint2 per block, up to 64 blocks a round).  Decoding the block
// minimum, building the index words and checking the widths then runs one block per lane.
//
// Streams often keep one block size for long stretches (a column of one bit width whose block minima have varints of one
// length).  When the last blocks of a round had equal sizes S, the next round guesses: lane l assumes that a block starts at
// pos + l * S and parses it there, out of a longer window (up to 16 KB), all lanes at once.  Lane 0's block is real; lane
// l's is real when every lane before it found a block of exactly S bytes.  So the round accepts the lanes up to and
// including the first whose block has another size (that block still starts where it was assumed to) and goes on behind
// it: up to 64 blocks for the price of one.  A guess that fails costs one parse; rounds that accept little send the walk
// back to the serial chain until sizes repeat again.
// The window is 16 KB where few packs are walked at a time (every stream's latency counts) and 4 KB in large launches
// (more walkers fit a compute unit).
// 16 bytes from global memory at any alignment, and byte i (0..15) of them
__device__ __forceinline__ void ld16g(const uint8_t* p, uint64_t& lo, uint64_t& hi) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
    hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
}
__device__ __forceinline__ uint32_t byte_of(uint64_t lo, uint64_t hi, uint32_t i) {
    return (uint32_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFFu);
}
using DeltaWin = WinT<16384>;
template <bool kFar, class DeltaWin>
__device__ int32_t walk_delta(DeltaWin& w, uint32_t pack_size, uint32_t off, uint32_t n, int width, int64_t* min_delta, uint32_t* mb,
                              int64_t* first_out, uint2* s_blk) {
    const uint32_t lane = threadIdx.x & 63;
    // the arguments are equal in all lanes; say so, so that the chain below stays in scalar registers
    pack_size = __builtin_amdgcn_readfirstlane(pack_size);
    off = __builtin_amdgcn_readfirstlane(off);
    n = __builtin_amdgcn_readfirstlane(n);
    width = __builtin_amdgcn_readfirstlane(width);
    if (off + 4 > pack_size) return E_CORRUPT;
    const int32_t len = (int32_t)__builtin_amdgcn_readfirstlane(ld32u(w.base + off));
    if (len < 0 || (uint64_t)off + 4 + (uint64_t)len > pack_size) return E_CORRUPT;
    uint32_t pos = off + 4;
    const uint32_t end = off + 4 + (uint32_t)len;
    int32_t err = 0;
    const uint32_t vb = width == 32 ? 5u : 10u, W = (uint32_t)width;
    uint64_t raw = win_uvarint(w, pos, end, (int)vb, err);
    *first_out = width == 32 ? unzigzag32(raw) : unzigzag64(raw);
    if (err) return err;
    const uint32_t nb = (n + kBlock - 1) / kBlock, nfull = n / kBlock;
    // a block is at most 10 + 4 * (1 + 4 * width) bytes: one window check per block, then plain LDS byte reads
    const uint32_t block_max = 10u + 4u * (1u + 4u * W) + 16u;
    bool bad = false;
    uint32_t guess = 0;       // block size to try next, 0: walk the chain
    uint32_t serial_len = 16; // blocks of the next serial round: short at first, to get to a guess early
    for (uint32_t b0 = 0; b0 < nb && !err;) {
        pos = __builtin_amdgcn_readfirstlane(pos);
        // ---- a round of guessed block starts
        if (guess && b0 < nfull) {
            const uint32_t want = min(64u, nfull - b0);
            // Where the window cannot hold a whole round (blocks of wide values: a kilobyte each at 64 bits), launches of few
            // packs (kFar) do not bring the blocks into it: a round needs the few framing bytes of every block, not the values
            // between them, and every lane fetches those from global memory itself (two round trips when the block's
            // mini-blocks share one width, one more for every width byte that sits elsewhere): 64 blocks a round instead of
            // the handful a window holds.  Launches that fill the chip with walkers keep to the window: they wait for
            // memory throughput, not for a chain, and the scattered 16-byte reads cost more of it than whole windows
            // (7 numeric columns x 1526 packs: 0.56 ms with windows, 0.62 with far rounds; one workgroup per pack: 1.05 / 0.89).
            const bool far = kFar && (DeltaWin::kBytes - 32u) / guess < want;
            uint32_t wend = 0xFFFFFFFFu, cover = want;
            if (!far) {
                const uint32_t span = min(want * guess + block_max, DeltaWin::kBytes - 16u);
                if (!w.has(pos, span)) w.load_n(pos, span);
                // lanes whose block lies in the window if it has the guessed size (lane 0's does whatever its size: span >= block_max)
                wend = w.w0 + w.len();
                cover = min(want, (wend - pos - 16u) / guess);
            }
            const uint32_t g = pos + lane * guess;  // pos < 2^24, a block is about a kilobyte at most
            const bool act = lane < cover;
            bool ok = act && g < end;
            const uint32_t gs = ok ? g : pos;  // lanes without a block read where lane 0 reads
            uint64_t lo, hi;
            if (far) {
                ld16g(w.base + gs, lo, hi);  // gs < end <= pack size: the 16 bytes end inside the arena's slack
            } else {
                lo = w.peek64(gs);
                hi = w.peek64(gs + 8);
            }
            uint64_t v = 0;
            uint32_t used = 0;
            bool open = true, term = false;
            const uint32_t avail = min(vb, end - gs);
#pragma unroll
            for (uint32_t i = 0; i < 10; i++) {
                const uint32_t byte = (uint32_t)((i < 8 ? lo : hi) >> (8 * (i & 7))) & 0xFFu;
                const bool take = open && i < avail;
                if (take) { v |= (uint64_t)(byte & 0x7Fu) << (7 * i); used = i + 1; term = !(byte & 0x80u); }
                open = take && (byte & 0x80u);
            }
            ok = ok && term;
            uint32_t p = gs + used, idx[4];
            bool wide = false;
            if (far) {
                // the first width byte came with the varint (used <= 10); the other three are fetched where they are if all
                // mini-blocks have the first one's width, and fetched again from their true place if not.  Nothing at or
                // beyond `end` is read: such a block is not a sound one (its position ends beyond `end`).
                const uint32_t bw0 = byte_of(lo, hi, used);
                const uint32_t stride = 1u + 4u * min(bw0, W);
                uint32_t spec[3];
#pragma unroll
                for (uint32_t m = 0; m < 3; m++) {
                    const uint32_t q = p + (m + 1u) * stride;
                    spec[m] = q < end ? (uint32_t)w.base[q] : 0u;
                }
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    uint32_t bw;
                    if (m == 0) bw = p < end ? bw0 : 0u;
                    else if (p == gs + used + m * stride) bw = spec[m - 1];
                    else bw = p < end ? (uint32_t)w.base[p] : 0u;
                    wide |= bw > W;
                    idx[m] = (p + 1) | (bw << 24);
                    p += 1 + 4 * min(bw, W);
                }
            } else {
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    const uint32_t bw = w.lds[min(p - w.w0, w.len() + 15u)];  // a block that runs off the window is not a covered one
                    wide |= bw > W;
                    idx[m] = (p + 1) | (bw << 24);
                    p += 1 + 4 * min(bw, W);
                }
            }
            const uint32_t size = p - gs;
            ok = ok && p <= end && p < (1u << 24);
            // the first covered lane whose block is not a sound block of the guessed size ends the round (and is part of it)
            const uint64_t odd = __builtin_amdgcn_ballot_w64(act && !(ok && size == guess));
            uint32_t last = odd ? (uint32_t)__builtin_ctzll(odd) : cover - 1u;
            // a block of another size may run off the window (lane 0's cannot): then its widths were not read from the pack;
            // the round ends before it and the chain takes it
            if (!far && last > 0 && (uint32_t)__builtin_amdgcn_readlane((int)p, (int)last) + 16u > wend) {
                last--;
                guess = 0;
            }
            if (!__builtin_amdgcn_readlane((int)ok, (int)last)) { err = E_CORRUPT; break; }  // a real block start, a broken block
            if (lane <= last) {
                const uint32_t b = b0 + lane;
                min_delta[b] = width == 32 ? unzigzag32(v) : unzigzag64(v);
                bad |= wide;
                *(uint2*)(mb + b * 4) = make_uint2(idx[0], idx[1]);  // the index words start 8-byte aligned
                *(uint2*)(mb + b * 4 + 2) = make_uint2(idx[2], idx[3]);
            }
            pos = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)last);
            b0 += last + 1;
            // a lone odd block in a long run keeps the guess; little accepted: back to the chain, for a few blocks at first (an
            // odd block among equal ones is the common reason, and eight blocks behind it show the old size again)
            if (last < 8u && last + 1u < cover) { guess = 0; serial_len = 8; }
            continue;
        }
        // ---- a round of the serial chain
        const uint32_t b1 = min(nb, b0 + serial_len);
        for (uint32_t b = b0; b < b1; b++) {
            // the position is the same in every lane: keeping it (and everything derived from it) in scalar registers makes
            // the loop's branches scalar branches instead of execution-mask bookkeeping
            pos = __builtin_amdgcn_readfirstlane(pos);
            if (pos >= end) { err = E_CORRUPT; break; }
            w.need(pos, block_max);
            const uint32_t start = pos;
            // the varint's length: bytes up to the first one without the continuation bit; it must end inside the stream
            // and within vb bytes (VarInteger.java:68-96)
            uint32_t vl = 1;
            if (w.u8(pos) >= 0x80u) {
                const uint64_t lo = w.peek64u(pos);
                const uint64_t stop = ~lo & 0x8080808080808080ull;
                if (stop) {
                    vl = ((uint32_t)__builtin_ctzll(stop) >> 3) + 1u;
                } else {
                    const uint32_t hi = (uint32_t)w.peek64u(pos + 8);
                    vl = !(hi & 0x80u) ? 9u : (!(hi & 0x8000u) ? 10u : 11u);
                }
                vl = __builtin_amdgcn_readfirstlane(vl);
                if (vl > min(vb, end - pos)) { err = E_CORRUPT; break; }
            }
            pos += vl;
            const uint32_t cnt = min((uint32_t)kBlock, n - b * kBlock);
            uint32_t bws = 0;
            if (cnt == (uint32_t)kBlock) {
                // full block: four mini-blocks of 32 values, 4 * bw payload bytes each.  Positions only grow, so running off the
                // stream (or past the 24 bits an index word holds) shows in the final position: one check per block.  Bytes
                // read beyond the stream's end in the meantime come from the window (zero filled past the arena's slack).
                // (Reading the candidates for all possible widths at once, to pay one trip to LDS per block instead of four, was
                // tried: a lone wave is bound by the number of instructions it issues, and that version issues more.)
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    const uint32_t bw = w.u8(pos);
                    bws |= bw << (8 * m);
                    pos += 1 + 4 * min(bw, W);
                }
            } else {
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) {
                    if (m * kMini < cnt && !err) {
                        const uint32_t c = min((uint32_t)kMini, cnt - m * kMini);
                        if (pos >= end) { err = E_CORRUPT; break; }
                        const uint32_t bw = w.u8(pos);
                        bws |= bw << (8 * m);
                        pos += 1 + ((c + 7) >> 3) * min(bw, W);
                    }
                }
                if (err) break;
            }
            if (pos > end || pos >= (1u << 24)) { err = E_CORRUPT; break; }
            if (lane == 0) s_blk[b - b0] = make_uint2(start, bws);
        }
        if (err) break;
        DeltaWin::wave_lds_sync();
        // one block per lane: the block minimum (its varint ends inside the stream, as checked above; the 16-byte load may run
        // into the arena's slack), the four index words, the width check
        const uint32_t b = b0 + lane;
        uint32_t my_size = 0;
        if (b < b1) {
            const uint2 rec = s_blk[lane];
            my_size = (b + 1 < b1 ? s_blk[lane + 1].x : pos) - rec.x;
            uint4 head;
            __builtin_memcpy(&head, w.base + rec.x, 16);
            const uint32_t hw[4] = {head.x, head.y, head.z, head.w};
            uint64_t v = 0;
            uint32_t used = 0;
            bool open = true;
#pragma unroll
            for (uint32_t i = 0; i < 10; i++) {
                const uint32_t byte = (hw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                if (open) { v |= (uint64_t)(byte & 0x7Fu) << (7 * i); used = i + 1; }
                open = open && (byte & 0x80u);
            }
            const int64_t md = width == 32 ? unzigzag32(v) : unzigzag64(v);
            const uint32_t cnt = min((uint32_t)kBlock, n - b * kBlock);
            uint32_t p = rec.x + used;
            min_delta[b] = md;
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                if (m * kMini < cnt) {
                    const uint32_t c = min((uint32_t)kMini, cnt - m * kMini);
                    const uint32_t bw = (rec.y >> (8 * m)) & 0xFFu;
                    bad |= bw > W;
                    mb[b * 4 + m] = (p + 1) | (bw << 24);
                    p += 1 + ((c + 7) >> 3) * min(bw, W);
                }
            }
        }
        DeltaWin::wave_lds_sync();  // the table is read before the next round overwrites it
        // guess when the round's last four blocks (or more) had one size
        const uint32_t cnt_r = b1 - b0;
        const uint32_t last_size = (uint32_t)__builtin_amdgcn_readlane((int)my_size, (int)(cnt_r - 1u));
        const uint64_t same = __builtin_amdgcn_ballot_w64(lane < cnt_r && my_size == last_size);
        const uint64_t differ = ~(same << (64u - cnt_r));
        const uint32_t streak = differ ? (uint32_t)__builtin_clzll(differ) : 64u;  // equal sizes from the round's last block down
        guess = (streak >= 4u && b1 < nfull) ? last_size : 0u;
        serial_len = min(64u, 2u * serial_len);  // no steady size in sight: longer rounds, fewer looks
        b0 = b1;
    }
    if (!err && pos != end) err = E_CORRUPT;
    if (err) return err;
    if (__ballot(bad)) err = E_CORRUPT;
    return err;
}

// ---------------------------------------------------------------------------------------------------------------
// RLE/bit-packing hybrid id stream (RLEPackingHybridCodec.decodeInts, RLEPackingHybridCodec.java:97-128): a chain of
// segments, each an unsigned varint header h (count = h >>> 1; h & 1: bit-packed, else a run) followed by
// ceil(count / 8) * bitWidth bytes of packed values or by the run's value in byteWidth bytes.  Where a header starts
// depends on every header before it, but following the chain one header at a time keeps a whole wave busy with scalar
// work.  Instead the wave looks at 64 bytes of the stream at once: every lane parses the bytes at its position as if a
// header started there (count, kind, value, position of the header after it).  Then a short scalar loop hops from the
// chunk's first byte, which is a true header, along the lanes' "next" values and marks the lanes it visits: a few scalar
// instructions per header.  The marked lanes hold the true segments; a wave scan of their counts gives every segment's
// first row.
// ---------------------------------------------------------------------------------------------------------------
template <typename U> __device__ __forceinline__ U wave_incl_scan(U v);

struct RleCand {
    uint32_t cnt;    // rows of the segment (clamped to 65536; `bad` when it was larger)
    uint32_t nb;     // bytes of the header
    uint32_t val;    // a run's value
    uint32_t nrel;   // position of the following header, relative to the chunk's first byte (always > lane)
    bool packed, bad;
};
// The header whose first bytes are `word` (little-endian, 8 bytes from its position), `avail` of which belong to the stream
// (0..5 matter).  nrel is left to the caller.
__device__ __forceinline__ RleCand rle_header(uint64_t word, uint32_t avail, uint32_t bw, uint32_t bytew, uint32_t vmask,
                                              uint32_t& payload) {
    // VarInteger.readUnsignedVarInt (VarInteger.java:68-81): seven bits per byte, the first byte without the top bit ends it
    const uint64_t term = ~word & 0x8080808080ull;
    const uint32_t nbr = term ? ((uint32_t)__builtin_ctzll(term) >> 3) + 1u : 6u;
    RleCand c;
    c.nb = max(min(nbr, avail), 1u);
    const uint32_t lo = (uint32_t)word, b4 = (uint32_t)(word >> 32) & 0x7Fu;
    uint32_t h = (lo & 0x7Fu) | ((lo >> 1) & 0x3F80u) | ((lo >> 2) & 0x1FC000u) | ((lo >> 3) & 0xFE00000u) | (b4 << 28);
    h &= c.nb >= 5u ? 0xFFFFFFFFu : ((1u << (7u * c.nb)) - 1u);
    c.packed = h & 1u;
    const uint32_t cnt = h >> 1;
    c.bad = nbr > avail || cnt > 65536u;  // the varint runs off the stream or has no end; more rows than a pack can hold
    c.cnt = min(cnt, 65536u);
    c.val = (uint32_t)(word >> (8u * c.nb)) & vmask;
    payload = c.packed ? ((c.cnt + 7u) >> 3) * bw : bytew;
    c.nrel = 0;
    return c;
}
// The header a lane sees at stream position c0 + lane; `end` is one past the stream's last byte.  [c0, c0 + 76) must be
// in the window.
template <class WinX>
__device__ __forceinline__ RleCand rle_candidate(const WinX& w, uint32_t c0, uint32_t end, uint32_t bw, uint32_t bytew,
                                                 uint32_t vmask) {
    const uint32_t lane = threadIdx.x & 63, pos = c0 + lane;
    const uint32_t avail = pos < end ? min(5u, end - pos) : 0u;
    uint32_t payload;
    RleCand c = rle_header(w.peek64(pos), avail, bw, bytew, vmask, payload);
    c.nrel = avail ? lane + c.nb + payload : 64u + lane;
    return c;
}
// Marks the lanes of the chunk that hold true headers (lane 0 does); `limit` (1..64) is the number of chunk bytes that
// belong to the stream, `exit_rel` is where the chain leaves them.
__device__ __forceinline__ uint64_t rle_chain(uint32_t nrel, uint32_t limit, uint32_t& exit_rel) {
    uint64_t mask = 0;
    uint32_t p = 0;
    do {
        mask |= 1ull << p;
        p = (uint32_t)__builtin_amdgcn_readlane((int)nrel, (int)p);
    } while (p < limit);
    exit_rel = p;
    return mask;
}

// Walk the id stream and record, for every tile of kRleTile rows, the segment that covers the tile's first row:
//   x = pack offset of the segment's payload (bit-packed values), or of the byte after a run's value
//   y = first row of the segment, z = its row count, w = 1 for bit-packed segments, (run value << 1) for runs
__device__ int32_t walk_rle(Win& w, uint32_t off, uint32_t len, uint32_t n, uint32_t bw, uint4* restart) {
    const uint32_t lane = threadIdx.x & 63;
    off = __builtin_amdgcn_readfirstlane(off);
    len = __builtin_amdgcn_readfirstlane(len);
    n = __builtin_amdgcn_readfirstlane(n);
    bw = __builtin_amdgcn_readfirstlane(bw);
    uint32_t pos = off + 4;
    const uint32_t end = off + 4 + len;
    const uint32_t bytew = (bw + 7) >> 3;
    const uint32_t vmask = (1u << (8 * bytew)) - 1u;
    uint32_t o = 0;
    uint32_t guess = 0;  // bytes of a segment (header and payload) to try next, 0: look at the next 64 bytes
    while (pos < end) {
        if (guess) {
            // ---- a round of guessed headers.  Ids without runs come as bit-packed segments of one length (the writer closes a
            // segment at 63 groups of 8 values: a kilobyte at 16 bits), far more than 64 bytes apart: a chunk then holds one
            // header and a round advances by one segment.  When that happened, lane l assumes that a header sits at
            // pos + l * guess and parses the 8 bytes there, fetched from global memory; lane 0's header is real, lane l's when
            // every lane before it found a sound segment of exactly `guess` bytes.  The round takes the lanes up to and
            // including the first with another size: up to 64 segments for the price of one trip to memory.
            const uint32_t g = pos + lane * guess;  // guess <= 5 + 8192 * 32, pack offsets < 2^31: no overflow
            const bool in = g < end;
            uint64_t word = 0;
            if (in) __builtin_memcpy(&word, w.base + g, 8);  // g < end <= pack size: the 8 bytes end inside the arena's slack
            uint32_t payload;
            const RleCand c = rle_header(word, in ? min(5u, end - g) : 0u, bw, bytew, vmask, payload);
            const uint32_t size = c.nb + payload;
            const bool sound = in && !c.bad && g + size <= end;
            const uint64_t odd = __builtin_amdgcn_ballot_w64(!(sound && size == guess));
            uint32_t last = odd ? (uint32_t)__builtin_ctzll(odd) : 63u;
            // the lane behind a run of sound segments starts at or before `end`; at `end` the stream is over (lane 0 is inside)
            if (!__builtin_amdgcn_readlane((int)in, (int)last)) last--;
            const bool mine = lane <= last;
            const uint32_t rows = mine ? c.cnt : 0u;
            const uint32_t inc = wave_incl_scan(rows);
            const uint32_t first = o + inc - rows;
            const bool bad = mine && (!sound || first + rows > n);
            if (__builtin_amdgcn_ballot_w64(bad)) return E_CORRUPT;
            if (mine && rows) {
                const uint32_t dp = g + c.nb;
                const uint4 rec = make_uint4(c.packed ? dp : dp + bytew, first, rows, c.packed ? 1u : (c.val << 1));
                for (uint32_t t = (first + kRleTile - 1) / kRleTile; t * (uint32_t)kRleTile < first + rows; t++) restart[t] = rec;
            }
            o += __builtin_amdgcn_readlane((int)inc, 63);
            pos = (uint32_t)__builtin_amdgcn_readlane((int)(g + size), (int)last);
            if (last < 8u) guess = 0;  // little taken: look at the bytes again
            continue;
        }
        if (!w.has(pos, 80u)) w.load(pos);
        const RleCand c = rle_candidate(w, pos, end, bw, bytew, vmask);
        uint32_t exit_rel;
        const uint64_t mask = rle_chain(c.nrel, min(64u, end - pos), exit_rel);
        const bool mine = (mask >> lane) & 1ull;
        const uint32_t rows = mine ? c.cnt : 0u;
        const uint32_t inc = wave_incl_scan(rows);
        const uint32_t first = o + inc - rows;  // < 2^23: at most 64 segments of at most 65536 rows
        // count > valCount - valIndex would write past the output; a payload must end inside the stream
        const bool bad = mine && (c.bad || first + rows > n || pos + c.nrel > end);
        if (__builtin_amdgcn_ballot_w64(bad)) return E_CORRUPT;
        if (mine && rows) {
            const uint32_t dp = pos + lane + c.nb;
            const uint4 rec = make_uint4(c.packed ? dp : dp + bytew, first, rows, c.packed ? 1u : (c.val << 1));
            // tiles whose first row lies inside this segment
            for (uint32_t t = (first + kRleTile - 1) / kRleTile; t * (uint32_t)kRleTile < first + rows; t++) restart[t] = rec;
        }
        o += __builtin_amdgcn_readlane((int)inc, 63);
        pos += exit_rel;
        // one header in the chunk and the next one far away: try whole segments of that size next
        if (mask == 1ull && exit_rel >= 256u) guess = exit_rel;
    }
    return o != n ? E_CORRUPT : 0;  // Preconditions.checkState(valIndex == valCount) :126
}

// What a pack is, from its descriptor and fixed-position header fields: kind, dictionary geometry, where its serial
// streams start (PackCompressor_VLT.decompress dispatch :130-170, DictStructUtil.decompress :83-142, DeltaCodec.decodeStrings
// :355-366).  `consumed` is the number of bytes the pack's framing claims.
__device__ __forceinline__ void pack_header(const DevPack& P, const uint8_t* base, PackMeta& m, int32_t& err, int64_t& consumed,
                                            bool& has_ids) {
    const uint32_t psz = (uint32_t)P.pack_size;
    const uint32_t n = (uint32_t)P.obj_count;
    const int W = P.width;  // 32, 64 or 0 (strings)
    m.status = 0; m.kind = K_NONE; m.entry_count = 0; m.id_bw = 0; m.ids_off = 0; m.ids_len = 0; m.entries_off = 0;
    m.entries_plain = 0; m.sfx_off = 0; m.sfx_len = 0; m.dict_struct_size = 0; m.max_len = 0; m.front_blocks = 0; m.front_stride = 0;
    m.sliced = 0;
    m.s[0].n = m.s[1].n = 0; m.s[0].in_off = m.s[1].in_off = 0; m.s[0].first = m.s[1].first = 0;
    err = 0;
    consumed = 0;
    has_ids = false;

    auto rd32 = [&](uint64_t off) -> uint32_t {  // bounds-checked header read
        if (off + 4 > psz) { if (!err) err = E_CORRUPT; return 0; }
        return ld32u(base + off);
    };

    if (n == 0 || n > 65536u || P.pack_size < 0 || P.data_size < 0) err = E_CORRUPT;
    if (!err) {
        switch (P.encode_type) {
            case 0:  // stored uncompressed (DataPack.java:180-182)
                m.kind = K_RAW;
                if (P.pack_size != P.data_size) err = E_OVERFLOW;
                consumed = psz;
                break;
            case 2:  // DELTA
                if (W) {
                    if ((uint64_t)n * (W / 8) != (uint64_t)P.data_size) err = E_OVERFLOW;
                    m.kind = K_DELTA_NUM;
                    m.s[0].n = n; m.s[0].in_off = 0;
                    consumed = 4 + (int64_t)(int32_t)rd32(0);
                } else {
                    m.kind = K_DELTA_STR;
                    uint32_t eds = rd32(0);
                    uint32_t l0 = rd32(4);
                    uint64_t o1 = 8ull + l0;
                    uint32_t l1 = rd32(o1);
                    uint64_t so = o1 + 4 + l1;
                    if ((int32_t)eds < 0 || (int32_t)l0 < 0 || (int32_t)l1 < 0 || so > psz || 8ull + l0 + l1 > eds) { if (!err) err = E_CORRUPT; }
                    m.s[0].n = n; m.s[0].in_off = 4;
                    m.s[1].n = n; m.s[1].in_off = (uint32_t)o1;
                    m.sfx_off = (uint32_t)so;
                    m.sfx_len = eds - (uint32_t)(8 + l0 + l1);
                    m.entry_count = n;
                    consumed = 4 + (int64_t)eds;
                }
                break;
            case 6: {  // DICT_COMPRESS
                m.kind = W ? K_DICT_NUM : K_DICT_STR;
                m.dict_struct_size = rd32(0);
                uint32_t ec = rd32(4);
                uint32_t comp = psz > 8 ? base[8] : 0;
                if (psz < 9 || ec == 0 || ec > n || (int32_t)m.dict_struct_size < 8) { if (!err) err = E_CORRUPT; ec = 1; }
                m.entry_count = ec;
                m.id_bw = bit_width32(ec - 1);
                uint64_t ids_off;
                if (W) {
                    if ((uint64_t)n * (W / 8) != (uint64_t)P.data_size) err = E_OVERFLOW;
                    if (m.dict_struct_size != 4 + ec * (W / 8) + 4 + 4 * n && !err) err = E_CORRUPT;
                    if (comp) {
                        m.s[0].n = ec; m.s[0].in_off = 9;
                        uint32_t l0 = rd32(9);
                        if ((int32_t)l0 < 0 && !err) err = E_CORRUPT;
                        ids_off = 13ull + l0;
                    } else {
                        m.entries_plain = 1; m.entries_off = 9;
                        ids_off = 9ull + (uint64_t)ec * (W / 8);
                    }
                } else {
                    if (comp) {
                        uint32_t eds = rd32(9);
                        uint32_t l0 = rd32(13);
                        uint64_t o1 = 17ull + l0;
                        uint32_t l1 = rd32(o1);
                        uint64_t so = o1 + 4 + l1;
                        if ((int32_t)eds < 0 || (int32_t)l0 < 0 || (int32_t)l1 < 0 || so > psz || 8ull + l0 + l1 > eds) { if (!err) err = E_CORRUPT; }
                        m.s[0].n = ec; m.s[0].in_off = 13;
                        m.s[1].n = ec; m.s[1].in_off = (uint32_t)o1;
                        m.sfx_off = (uint32_t)so;
                        m.sfx_len = eds - (uint32_t)(8 + l0 + l1);
                        ids_off = 13ull + eds;
                    } else {
                        m.entries_plain = 1; m.entries_off = 9;
                        uint32_t tot = rd32(9ull + 4ull * ec);
                        if ((int32_t)tot < 0 && !err) err = E_CORRUPT;
                        ids_off = 9ull + 4ull * (ec + 1) + tot;
                    }
                }
                uint32_t il = rd32(ids_off);
                if (((int32_t)il < 0 || ids_off + 4 + il > psz) && !err) err = E_CORRUPT;
                m.ids_off = (uint32_t)ids_off; m.ids_len = il;
                has_ids = true;
                consumed = (int64_t)ids_off + 4 + il;
                break;
            }
            case 4: {  // SIMPLE_DICT (legacy, decode only)
                m.kind = W ? K_DICT_NUM : K_DICT_STR;
                uint32_t ec = rd32(0);
                if (ec == 0 || ec > n) { if (!err) err = E_CORRUPT; ec = 1; }
                m.entry_count = ec;
                m.id_bw = bit_width32(ec - 1);
                m.entries_plain = 1; m.entries_off = 4;
                uint64_t dict_size;
                if (W) {
                    if ((uint64_t)n * (W / 8) != (uint64_t)P.data_size) err = E_OVERFLOW;
                    dict_size = (uint64_t)ec * (W / 8);
                } else {
                    uint32_t tot = rd32(4ull + 4ull * ec);
                    if ((int32_t)tot < 0 && !err) err = E_CORRUPT;
                    dict_size = 4ull * (ec + 1) + tot;
                }
                uint64_t ids_off = 4 + dict_size + 4;
                uint32_t il = rd32(ids_off);
                if (((int32_t)il < 0 || ids_off + 4 + il > psz) && !err) err = E_CORRUPT;
                m.ids_off = (uint32_t)ids_off; m.ids_len = il;
                has_ids = true;
                consumed = (int64_t)ids_off + 4 + il;
                break;
            }
            default:
                err = E_ILLEGAL_ENC;
        }
    }
    if (!err && (consumed < 0 || consumed > (int64_t)psz)) err = E_CORRUPT;

}

// ---------------------------------------------------------------------------------------------------------------
// The framing walk of launches of thousands of packs, in two kernels.  k_parse_head: one thread per pack derives the
// record (kind, geometry, where the streams start; status = bytes the framing claims, or the header's error).
// k_parse_waves: one WAVE (a workgroup of its own) per stream with the window walkers, grid (packs, 3): stream 0 and 1 the
// DELTA streams, stream 2 the id stream.  It completes the records (first value, status on error: the walkers know one
// error code, so the waves of a pack cannot disagree about it).  A walker holds one wave slot and 4.6 KB of LDS, where the
// workgroup-per-pack form holds four wave slots and three windows whatever the pack has: a compute unit carries four
// times the chains (7 numeric columns x 1526 packs: 1.03 -> 0.56 ms; 24 416 string packs: 1.76 ms).
// (One LANE per stream, reading the framing from global memory, was the form for tens of thousands of packs before: every
// step of a chain then pays a trip to memory, and the launch waits for the longest chain: 4.3 ms on those 24 416 string
// packs, 1.58 against 1.40 ms on the 76 296 packs of the mixed table.)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_parse_head(const DevPack* __restrict__ packs, const int* __restrict__ list, int count,
                                                    PackMeta* __restrict__ metas, const uint8_t* __restrict__ in) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= count) return;
    const int pi = list[li];
    const DevPack P = packs[pi];
    PackMeta m;
    int32_t err;
    int64_t consumed;
    bool has_ids;
    pack_header(P, in + P.in_off, m, err, consumed, has_ids);
    m.status = err ? err : (int32_t)consumed;
    metas[pi] = m;
}

template <class DeltaWin>
__global__ __launch_bounds__(64) void k_parse_waves(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                    PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                    uint8_t* __restrict__ scratch) {
    static_assert(DeltaWin::kAlloc >= kWinAlloc, "one LDS window serves either walker");
    __shared__ __align__(16) uint8_t s_win[DeltaWin::kAlloc];
    __shared__ uint2 s_blk[65];
    const int k = blockIdx.y;
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    PackMeta& M = metas[pi];
    if (M.status < 0) return;
    int32_t werr = 0;
    if (k < 2) {
        const StreamDesc sd = M.s[k];
        if (!sd.n) return;
        const DevPack P = packs[pi];
        DeltaWin w;
        w.init(in + P.in_off, (uint32_t)P.pack_size, s_win);
        int64_t* md = (int64_t*)(scratch + P.s_idx[k]);
        uint32_t* mb = (uint32_t*)(md + (P.obj_count + kBlock - 1) / kBlock);
        int64_t first = 0;
        werr = walk_delta<false>(w, (uint32_t)P.pack_size, sd.in_off, sd.n, P.width ? P.width : 32, md, mb, &first, s_blk);
        if (threadIdx.x == 0) M.s[k].first = first;
    } else {
        if (M.kind != K_DICT_NUM && M.kind != K_DICT_STR) return;
        const DevPack P = packs[pi];
        Win w;
        w.init(in + P.in_off, (uint32_t)P.pack_size, s_win);
        werr = walk_rle(w, M.ids_off, M.ids_len, (uint32_t)P.obj_count, M.id_bw, (uint4*)(scratch + P.s_restart));
    }
    if (threadIdx.x == 0 && werr) M.status = werr;
}

template <class DeltaWin>
__global__ __launch_bounds__(256) void k_parse(const DevPack* __restrict__ packs, const int* __restrict__ list, int count,
                                               PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                               uint8_t* __restrict__ scratch) {
    __shared__ __align__(16) uint8_t s_dwin[2][DeltaWin::kAlloc];  // the windows of the two DELTA walkers
    __shared__ __align__(16) uint8_t s_rwin[kWinAlloc];            // the window of the id stream's walker
    __shared__ uint2 s_blk[2][65];  // per DELTA stream: position and widths of the blocks of one round
    __shared__ int32_t s_werr[4];
    __shared__ int64_t s_first[2];
    const int k = threadIdx.x >> 6;  // wave = stream
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    const DevPack P = packs[pi];
    const uint8_t* base = in + P.in_off;
    const uint32_t psz = (uint32_t)P.pack_size;
    const uint32_t n = (uint32_t)P.obj_count;
    const int W = P.width;  // 32, 64 or 0 (strings)
    (void)count;

    // every wave derives the pack's header fields (a few uniform loads); wave 0 publishes them
    PackMeta m;
    int32_t err;
    int64_t consumed;
    bool has_ids;
    pack_header(P, base, m, err, consumed, has_ids);

    // ---- the serial walks, one per wave
    int32_t werr = 0;
    int64_t first = 0;
    if (!err && k < 3) {
        const StreamDesc sd = k == 0 ? m.s[0] : m.s[1];
        if (k < 2 && sd.n) {
            DeltaWin w;
            w.init(base, psz, s_dwin[k]);
            int64_t* md = (int64_t*)(scratch + (k == 0 ? P.s_idx[0] : P.s_idx[1]));
            uint32_t* mb = (uint32_t*)(md + (P.obj_count + kBlock - 1) / kBlock);
            werr = walk_delta<true>(w, psz, sd.in_off, sd.n, W ? W : 32, md, mb, &first, s_blk[k]);
        } else if (k == 2 && has_ids) {
            Win w;
            w.init(base, psz, s_rwin);
            werr = walk_rle(w, m.ids_off, m.ids_len, n, m.id_bw, (uint4*)(scratch + P.s_restart));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_werr[k] = werr;
        if (k < 2) s_first[k] = first;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        m.s[0].first = s_first[0];
        m.s[1].first = s_first[1];
        if (!err) err = s_werr[0] ? s_werr[0] : (s_werr[1] ? s_werr[1] : s_werr[2]);
        m.status = err ? err : (int32_t)consumed;
        metas[pi] = m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wave / workgroup scan helpers
// ---------------------------------------------------------------------------------------------------------------
// What lane (l - shift) of the same row of 16 lanes holds, or what the row's neighbour broadcasts, as a DPP operand: no trip
// through the LDS crossbar.  Lanes without a source get 0.  All 64 lanes must be active.
template <int kCtrl, int kRowMask, typename U>
__device__ __forceinline__ U dpp_or_zero(U v) {
    static_assert(sizeof(U) == 4 || sizeof(U) == 8, "32- or 64-bit values");
    if constexpr (sizeof(U) == 4) {
        return (U)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xF, false);
    } else {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, kCtrl, kRowMask, 0xF, false);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)v >> 32), kCtrl, kRowMask, 0xF, false);
        return (U)(((uint64_t)hi << 32) | lo);
    }
}
template <typename U> __device__ __forceinline__ U wave_incl_scan(U v) {
    v += dpp_or_zero<0x111, 0xF>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v);  // row_shr:8: every row of 16 lanes is scanned
    v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// k_delta: one workgroup per DELTA stream, 256 threads in large launches and 1024 in small ones (a stream is a chain of
// dependent iterations: with few streams in flight the chain is what the launch waits for, and four times the threads make it
// a quarter as long).  Each thread unpacks 4 or 8 consecutive values (same mini-block) per iteration; a thread-local sum, a wave
// scan and an LDS combine of the waves' totals give the running values; the carry travels in a register.  Replaces the two passes of DeltaCodec.decodeIntsBlock / decodeLongsBlock (DeltaCodec.java:112-127, 216-231).
// Also copies plain (uncompressed) dictionary entries into the aligned scratch dictionary (PlainCodec.java:48-70).
// ---------------------------------------------------------------------------------------------------------------
// How the numeric dictionary packs are shared out between the row kernels, by the size and use of their dictionaries:
// 2 to 64 KB: looked up in LDS (k_dict_lds); larger ones whose rows ask for the entries in order: the slice path of k_dict_num; the rest
// (and dictionaries below 2 KB, which stay in the vector L1): gathers from the scratch dictionary (k_dict_num).
constexpr uint32_t kDictLdsBytes = 65536, kDictLdsMinBytes = 2048;
constexpr uint32_t kSliceBlocks = 9;      // 128-entry blocks the slice path of k_dict_num decodes per tile: a stretch of 1025 entries at any alignment
constexpr uint32_t kSliceSpan = 1024;     // largest rise of the ids over a tile that one round covers
constexpr uint32_t kSliceMaxBw = 28;      // widest mini-block of a sliced dictionary: two neighbouring entries then sit in one 8-byte load
constexpr uint32_t kSliceFirstOff = 32768;  // sliced packs: the id of every tile's first row, behind the block records (at most 512 x 32 bytes)

template <typename U, int kThreads>
__global__ __launch_bounds__(kThreads) void k_delta(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                               PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                               uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, int dict_only) {
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    const int sk = blockIdx.y;
    const DevPack P = packs[pi];
    const PackMeta& M = metas[pi];
    if (M.status < 0) return;
    const int kind = M.kind;
    if ((dict_only & 1) && kind == K_DELTA_NUM) return;  // its rows would go to the output arena, which does not exist in this mode
    const uint8_t* base = in + P.in_off;
    const int tid = threadIdx.x;

    if (sk == 0 && M.entries_plain && kind == K_DICT_NUM) {
        // plain numeric entries: byte-granular copy into the aligned scratch dictionary
        const uint32_t bytes = M.entry_count * (uint32_t)sizeof(U);
        const uint8_t* src = base + M.entries_off;
        uint8_t* dst = scratch + P.s_dict;
        if (((uintptr_t)src & 3) == 0) {
            for (uint32_t i = tid; i < bytes / 4; i += kThreads) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
        } else {
            for (uint32_t i = tid; i < bytes; i += kThreads) dst[i] = src[i];
        }
        return;
    }
    const StreamDesc s = sk == 0 ? M.s[0] : M.s[1];
    if (s.n == 0) return;

    // A sorted dictionary whose rows ask for its entries in order (a column that grows with the row number: timestamps,
    // counters; the ids then step by 0 or 1) is not written to scratch at all: a tile of rows needs a stretch of at most
    // kSliceSpan entries, which k_dict_num decodes into LDS by itself from the value in front of the stretch's first
    // 128-entry block.  Those values, one per block, are all this kernel leaves for such a pack.  The choice is made from the
    // id of every tile's first row (the segment records of the framing walk give it without walking the stream): they must
    // rise from tile to tile by at most kSliceSpan.  Ids that misbehave inside a tile cost k_dict_num extra rounds, no more.
    bool sliced = false;
    if (kind == K_DICT_NUM && dict_only == 2 && (uint64_t)M.entry_count * sizeof(U) > kDictLdsBytes) {
        __shared__ int s_sliced;
        __shared__ uint32_t s_maxbw;
        if (tid == 0) s_maxbw = 0;
        __syncthreads();
        {   // the widest mini-block of the entry stream (the framing walk left one word per mini-block)
            const uint32_t* mbw = (const uint32_t*)((const int64_t*)(scratch + P.s_idx[0]) + (P.obj_count + kBlock - 1) / kBlock);
            uint32_t mw = 0;
            for (uint32_t i = tid; i < (s.n + kMini - 1) / kMini; i += kThreads) mw = max(mw, mbw[i] >> 24);
            if (mw) atomicMax(&s_maxbw, mw);
        }
        __syncthreads();
        const uint32_t ntiles = ((uint32_t)P.obj_count + kRleTile - 1) / kRleTile;
        uint32_t id = M.entry_count - 1u;
        if (tid < 64) {
            const uint32_t ec = M.entry_count, bw = M.id_bw;
            if ((uint32_t)tid < ntiles) {
                const uint4 rec = ((const uint4*)(scratch + P.s_restart))[tid];
                id = (rec.w & 1u) ? extract32(base + rec.x, ((uint32_t)tid * kRleTile - rec.y) * bw, bw) : (rec.w >> 1);
            }
            uint32_t next = (uint32_t)__shfl_down((int)id, 1);
            if ((uint32_t)tid + 1 >= ntiles) next = ec - 1u;
            const bool ok = (uint32_t)tid >= ntiles || (id < ec && next >= id && next - id <= kSliceSpan);
            const uint64_t no = __builtin_amdgcn_ballot_w64(!ok);
            if (tid == 0) {
                s_sliced = no == 0 && s_maxbw <= kSliceMaxBw && s.n == ec;
                if (s_sliced) metas[pi].sliced = 1;
            }
        }
        __syncthreads();
        sliced = s_sliced != 0;
        // the first id of every tile: where k_dict_num starts decoding for the tile
        if (sliced && (uint32_t)tid < ntiles && tid < 64) ((uint32_t*)(scratch + P.s_dict + kSliceFirstOff))[tid] = id;
    }

    U* dst;
    if (kind == K_DELTA_NUM) dst = (U*)(out + P.out_off);
    else if (kind == K_DICT_NUM) dst = (U*)(scratch + P.s_dict);
    else dst = (U*)(scratch + P.s_lens + (size_t)sk * 4 * (size_t)lens_stride(P.obj_count));  // string length streams (always 32 bit)

    const int64_t* min_delta = (const int64_t*)(scratch + (sk == 0 ? P.s_idx[0] : P.s_idx[1]));
    const uint32_t* mb = (const uint32_t*)(min_delta + (P.obj_count + kBlock - 1) / kBlock);
    const uint32_t n = s.n;
    const int lane = tid & 63, wave = tid >> 6;
    // consecutive values per thread (one mini-block header serves all of them): 8 pays for 32-bit values, 64-bit values
    // are faster with 4 (three dword loads per extraction already)
    constexpr int V = sizeof(U) == 4 ? 8 : 4;
    constexpr uint32_t kChunk = kThreads * V;  // values per workgroup iteration
    constexpr int kWaves = kThreads / 64;
    __shared__ U wsum[2][kWaves];
    __shared__ uint32_t s_win[12][kThreads];  // per thread: the up to 48 bytes its V values are packed in, dword k at [k][tid]
    U carry = (U)s.first;
    int buf = 0;
    // A chunk's values take two dependent round trips (the mini-block index, then the packed bits) and the workgroup
    // walks the chunks one after the other, so the trips are taken ahead of time: the index entries three chunks ahead
    // and the packed bits two chunks ahead are requested before a chunk is scanned and stored, and are in registers by
    // the time their chunk is unpacked.
    struct Head { uint32_t hdr; U md; };     // stage 1: mini-block record and block minimum of this thread's values
    struct Bits { uint4 w[3]; uint32_t sh, bw; U md; };  // stage 2: the up to 48 bytes the V values are packed in
    auto head_of = [&](uint32_t c0) {
        const uint32_t i0 = c0 + V * tid;
        Head h{0u, (U)0};
        if (i0 < n) { h.hdr = mb[i0 >> 5]; h.md = (U)min_delta[i0 >> 7]; }
        return h;
    };
    // The V values sit in V * bw consecutive bits: they are fetched with one to three 16-byte loads (instead of two or
    // three dword loads per value)
    auto bits_of = [&](uint32_t c0, const Head& h) {
        const uint32_t i0 = c0 + V * tid;
        Bits b;
        b.w[0] = b.w[1] = b.w[2] = make_uint4(0, 0, 0, 0);
        b.bw = h.hdr >> 24;
        b.md = h.md;
        b.sh = 0;
        if (i0 < n && b.bw) {
            const uint32_t bit = (i0 & 31) * b.bw;
            const uintptr_t a = (uintptr_t)(base + (h.hdr & 0xFFFFFFu) + (bit >> 3));
            const uint8_t* al = (const uint8_t*)(a & ~(uintptr_t)3);
            b.sh = ((uint32_t)(a & 3) << 3) + (bit & 7);  // 0..31
            // bits of this thread's values that exist: a load reaches less than 16 bytes past them (the arena's slack)
            const uint32_t nbits = b.sh + min((uint32_t)V, n - i0) * b.bw;  // <= 31 + 256
            __builtin_memcpy(&b.w[0], al, 16);
            if (nbits > 128) __builtin_memcpy(&b.w[1], al + 16, 16);
            if (nbits > 256) __builtin_memcpy(&b.w[2], al + 32, 16);
        }
        return b;
    };
    // stage 3: park the bytes in the thread's LDS column and pick the values out of it at their bit offsets (minDelta
    // added); zeros past the end
    auto unpack = [&](uint32_t c0, const Bits& b, U (&d)[V]) {
        const uint32_t i0 = c0 + V * tid;
#pragma unroll
        for (int q = 0; q < V; q++) d[q] = 0;
        if (i0 >= n) return;
        const uint32_t bw = b.bw;
        const U md = b.md;
        if (bw == 0) {
#pragma unroll
            for (int q = 0; q < V; q++)
                if (i0 + q < n) d[q] = md;
            return;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s_win[4 * k + 0][tid] = b.w[k].x; s_win[4 * k + 1][tid] = b.w[k].y;
            s_win[4 * k + 2][tid] = b.w[k].z; s_win[4 * k + 3][tid] = b.w[k].w;
        }
#pragma unroll
        for (int q = 0; q < V; q++) {
            if (i0 + q < n) {
                const uint32_t off = b.sh + q * bw, k = off >> 5, r = off & 31;
                const uint64_t lo = (uint64_t)s_win[k][tid] | ((uint64_t)s_win[k + 1][tid] << 32);
                if constexpr (sizeof(U) == 4) {
                    const uint32_t v = (uint32_t)(lo >> r);
                    d[q] = (bw >= 32 ? v : (v & ((1u << bw) - 1u))) + md;
                } else {
                    const uint64_t hi = s_win[k + 2][tid];
                    const uint64_t v = r ? ((lo >> r) | (hi << (64 - r))) : lo;
                    d[q] = (bw >= 64 ? v : (v & ((1ull << bw) - 1ull))) + md;
                }
            }
        }
    };
    Bits b0 = bits_of(0, head_of(0));
    Bits b1 = bits_of(kChunk, head_of(kChunk));
    Head h2 = head_of(2 * kChunk);
    for (uint32_t c0 = 0; c0 < n; c0 += kChunk, buf ^= 1) {
        const uint32_t i0 = c0 + V * tid;
        U d[V];
        unpack(c0, b0, d);
        b0 = b1;
        b1 = bits_of(c0 + 2 * kChunk, h2);
        h2 = head_of(c0 + 3 * kChunk);
#pragma unroll
        for (int q = 1; q < V; q++) d[q] += d[q - 1];
        const U incl = wave_incl_scan<U>(d[V - 1]);
        if (lane == 63) wsum[buf][wave] = incl;
        __syncthreads();
        U wprefix = 0, wtotal = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const U t = wsum[buf][w];
            if (w < wave) wprefix += t;
            wtotal += t;
        }
        const U pre = carry + wprefix + (incl - d[V - 1]);
        if (sliced) {
            // one 32-byte record per 128-entry block: its four mini-block words (position | width << 24), its minimum, and the
            // value in front of it (`pre` is the entry before this thread's first one)
            if ((i0 & (kBlock - 1)) == 0 && i0 < n) {
                const uint32_t b = i0 >> 7;
                uint4* rec = (uint4*)(scratch + P.s_dict) + 2 * b;
                const uint64_t md = (uint64_t)min_delta[b], pv = (uint64_t)pre;
                rec[0] = make_uint4(mb[4 * b], mb[4 * b + 1], mb[4 * b + 2], mb[4 * b + 3]);
                rec[1] = make_uint4((uint32_t)md, (uint32_t)(md >> 32), (uint32_t)pv, (uint32_t)(pv >> 32));
            }
        } else if (i0 + V - 1 < n) {
            if constexpr (sizeof(U) == 4) {
#pragma unroll
                for (int q = 0; q < V; q += 4) *(uint4*)(dst + i0 + q) = make_uint4(d[q] + pre, d[q + 1] + pre, d[q + 2] + pre, d[q + 3] + pre);
            } else {
                ulonglong2* o2 = (ulonglong2*)(dst + i0);
#pragma unroll
                for (int q = 0; q < V; q += 2) o2[q / 2] = make_ulonglong2(d[q] + pre, d[q + 1] + pre);
            }
        } else {
#pragma unroll
            for (int q = 0; q < V; q++)
                if (i0 + q < n) dst[i0 + q] = d[q] + pre;
        }
        carry += wtotal;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RLE tile expansion shared by the id-consuming kernels: a wave expands the ids (always < 65536) of its tile of kRleTile
// rows into LDS.  It starts from the segment record the parse kernel left for the tile, pulls the tile's stretch of the id
// stream into its LDS window with one coalesced load and walks the remaining segment headers there.  The consumers read
// the ids back in the order that suits their stores, which also lets them keep 16 independent gathers in flight.
// ---------------------------------------------------------------------------------------------------------------
template <class WinX>
__device__ __forceinline__ void rle_tile_expand(WinX& w, uint32_t bw, uint32_t ids_end, uint4 seg, uint32_t tile_start,
                                                uint32_t tile_end, uint16_t* ids) {
    const uint32_t lane = threadIdx.x & 63;
    bw = __builtin_amdgcn_readfirstlane(bw);
    ids_end = __builtin_amdgcn_readfirstlane(ids_end);
    tile_start = __builtin_amdgcn_readfirstlane(tile_start);
    tile_end = __builtin_amdgcn_readfirstlane(tile_end);
    const uint32_t bytew = (bw + 7) >> 3, vmask = (1u << (8 * bytew)) - 1u;  // bytew is 0..2
    // One segment expanded by the whole wave; all arguments are the same in every lane.
    auto expand_wide = [&](uint32_t dpos, uint32_t o, uint32_t cnt, bool packed, uint32_t val) {
        const uint32_t lo = max(o, tile_start), hi = min(o + cnt, tile_end);
        if (lo >= hi) return;
        if (packed) {
            // bytes holding rows [lo, hi) of this segment: at most kRleTile * 16 bits; a window smaller than that takes
            // the rows in stretches that fit (a stretch starts at most 15 bytes into the window, reads run 8 bytes over)
            const uint32_t rows_per = (((WinX::kBytes - 48u) * 8u) / max(bw, 1u)) & ~63u;
            for (uint32_t l2 = lo; l2 < hi; l2 += rows_per) {
                const uint32_t h2 = min(hi, l2 + rows_per);
                const uint32_t b0 = dpos + (((l2 - o) * bw) >> 3);
                const uint32_t b1 = dpos + (((h2 - o) * bw + 7) >> 3);
                if (!w.has(b0, b1 - b0 + 8)) w.load(b0);
#pragma unroll 4
                for (uint32_t i = l2 + lane; i < h2; i += 64) ids[i - tile_start] = (uint16_t)w.bits(dpos, (i - o) * bw, bw);
            }
        } else {
            for (uint32_t i = lo + lane; i < hi; i += 64) ids[i - tile_start] = (uint16_t)val;
        }
    };
    // the segment that covers the tile's first row, as the parse kernel recorded it
    uint32_t o, next;
    {
        const uint32_t dpos = __builtin_amdgcn_readfirstlane(seg.x), o0 = __builtin_amdgcn_readfirstlane(seg.y);
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(seg.z), sw = __builtin_amdgcn_readfirstlane(seg.w);
        expand_wide(dpos, o0, cnt, sw & 1u, sw >> 1);
        o = o0 + cnt;
        next = (sw & 1u) ? dpos + ((cnt + 7) >> 3) * bw : dpos;
    }
    // the headers after it, 64 bytes of the stream at a time (see rle_candidate): segments of up to kShort rows are
    // expanded by the lane that holds their header, longer ones by the whole wave one after the other
    constexpr uint32_t kShort = 32;
    while (o < tile_end && next < ids_end) {
        // the chunk, a header that starts at its last byte, and the payload of a short bit-packed segment behind it
        if (!w.has(next, 64u + 8u + (kShort / 8u) * 16u + 8u)) w.load(next);
        const RleCand c = rle_candidate(w, next, ids_end, bw, bytew, vmask);
        uint32_t exit_rel;
        const uint64_t mask = rle_chain(c.nrel, min(64u, ids_end - next), exit_rel);
        const bool mine = (mask >> lane) & 1ull;
        const uint32_t rows = mine ? c.cnt : 0u;
        const uint32_t inc = wave_incl_scan(rows);
        const uint32_t first = o + inc - rows;
        const uint32_t lo = max(first, tile_start), hi = min(first + rows, tile_end);
        const uint32_t dp = next + lane + c.nb;  // payload of a bit-packed segment
        const bool live = lo < hi;
        // The short segments' rows, one row per lane and pass: the rows are numbered through (a second scan), a row finds
        // the lane that holds its segment by a binary search over the lanes' running totals (six lane reads), fetches what
        // it needs of the segment from that lane and writes its id.  (One lane per segment writing its rows one after the
        // other was measured first: the wave then takes as many steps as the chunk's longest segment has rows.)
        const uint32_t srows = (live && rows <= kShort) ? hi - lo : 0u;
        const uint32_t sinc = wave_incl_scan(srows);
        const uint32_t stotal = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
        if (stotal) {
            const uint32_t sfirst = sinc - srows;
            const uint32_t meta = (lo - tile_start) | ((lo - first) << 10) | ((uint32_t)c.packed << 16);  // row in the tile, rows skipped (< 32)
            const uint32_t what = c.packed ? dp : c.val;
            for (uint32_t j0 = 0; j0 < stotal; j0 += 64) {
                const uint32_t j = j0 + lane;
                uint32_t own = 0;  // number of lanes whose running total is <= j: the lane of the segment row j belongs to
#pragma unroll
                for (uint32_t step = 32; step >= 1; step >>= 1)
                    if ((uint32_t)__shfl((int)sinc, (int)(own + step - 1u)) <= j) own += step;
                const uint32_t k = j - (uint32_t)__shfl((int)sfirst, (int)own);
                const uint32_t m = (uint32_t)__shfl((int)meta, (int)own), x = (uint32_t)__shfl((int)what, (int)own);
                if (j < stotal) {
                    const uint32_t id = (m >> 16) ? w.bits(x, (((m >> 10) & 63u) + k) * bw, bw) : x;
                    ids[(m & 1023u) + k] = (uint16_t)id;
                }
            }
        }
        uint64_t wide = __builtin_amdgcn_ballot_w64(live && rows > kShort);
        while (wide) {
            const int l = __builtin_ctzll(wide);
            wide &= wide - 1;
            expand_wide((uint32_t)__builtin_amdgcn_readlane((int)dp, l), (uint32_t)__builtin_amdgcn_readlane((int)first, l),
                        (uint32_t)__builtin_amdgcn_readlane((int)rows, l), __builtin_amdgcn_readlane((int)c.packed, l),
                        (uint32_t)__builtin_amdgcn_readlane((int)c.val, l));
        }
        o += __builtin_amdgcn_readlane((int)inc, 63);
        next += exit_rel;
    }
    // the tile's ids were written by this wave only and are read by this wave only
    WinX::wave_lds_sync();
}

// Common prologue of the tile kernels: which pack / tile this wave owns. Returns false when there is nothing to do.
struct TileCtx {
    int pi;
    uint32_t n, tile, tile_start, tile_end, ec;
};
__device__ __forceinline__ bool tile_ctx(TileCtx& t, const DevPack* packs, const int* list, const PackMeta* metas, int kind,
                                         uint32_t first_tile_block) {
    uint32_t bx, by;
    pack_block_remap(bx, by);
    t.pi = list[by];
    const PackMeta& M = metas[t.pi];
    if (M.status < 0 || M.kind != kind || bx < first_tile_block) return false;
    t.n = (uint32_t)packs[t.pi].obj_count;
    t.tile = (bx - first_tile_block) * 4 + (threadIdx.x >> 6);
    t.tile_start = t.tile * kRleTile;
    if (t.tile_start >= t.n) return false;
    t.tile_end = min(t.n, t.tile_start + (uint32_t)kRleTile);
    t.ec = M.entry_count;
    return true;
}

// Dictionaries of 2 to 64 KB are looked up in LDS (k_dict_lds).  Smaller ones stay in the vector L1 anyway and their packs
// -- few distinct values, often run-heavy id streams whose expansion is the work -- are better off with the smaller
// workgroups of k_dict_num (measured: 16 values in runs lose a third in the LDS kernel); larger ones do not fit.
template <typename U> __device__ __forceinline__ bool dict_in_lds(uint32_t entry_count) {
    const uint32_t bytes = entry_count * (uint32_t)sizeof(U);
    return bytes >= kDictLdsMinBytes && bytes <= kDictLdsBytes;
}
// ---------------------------------------------------------------------------------------------------------------
// k_dict_num: numeric dictionary packs.  One wave per tile of kRleTile rows: out[i] = dict[id[i]]
// (DictStruct.decodeInts/Longs/Floats/Doubles, indexr-common/.../data/DictStruct.java:57-95).
//
// Packs marked `sliced` by k_delta (sorted dictionary, rows that ask for its entries in order, no mini-block of the entry
// stream wider than kSliceMaxBw bits) have no dictionary in scratch.  k_delta left, per 128-entry block, a 32-byte record
// (the block's four mini-block words, its minimum, the value in front of it) and, per tile, the id of the tile's first row.
// The wave reads the records of the kSliceBlocks blocks from that id on and then the packed bits of those blocks -- two
// neighbouring entries per lane and block in ONE 8-byte load -- while it expands the tile's ids; it decodes the blocks
// into LDS, kRound at a time (every block starts from its own value, so they are independent: bits, + minDelta, one wave
// scan per block; DeltaCodec.decodeLongsBlock, DeltaCodec.java:216-231) and looks the rows up there.  Rows whose ids lie
// outside that stretch (ids that jump inside a tile) are produced afterwards, one 128-entry block at a time: any id
// stream decodes correctly, only not quickly.
// ---------------------------------------------------------------------------------------------------------------
template <typename U>
__device__ __forceinline__ void slice_two(uint64_t w, uint32_t sh, uint32_t bw, U md, bool has0, bool has1, U& a0, U& a1) {
    // two neighbouring entries out of the 8 bytes that hold them: sh <= 7, bw <= kSliceMaxBw, so sh + 2 * bw <= 63
    const uint64_t mask = (1ull << bw) - 1ull;
    a0 = has0 ? (U)((w >> sh) & mask) + md : (U)0;
    a1 = has1 ? (U)((w >> (sh + bw)) & mask) + md : (U)0;
}
template <typename U>
__device__ __forceinline__ void store_pair(U* p, U a, U b) {  // p is 2 * sizeof(U) aligned; streaming store
    if constexpr (sizeof(U) == 4) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 t = {(uint32_t)a, (uint32_t)b};
        __builtin_nontemporal_store(t, (u32x2*)p);
    } else {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        u64x2 t = {(uint64_t)a, (uint64_t)b};
        __builtin_nontemporal_store(t, (u64x2*)p);
    }
}
template <typename U>
__global__ __launch_bounds__(256, 6) void k_dict_num(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                  PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                  uint8_t* __restrict__ out, const uint8_t* __restrict__ scratch) {
    // per wave: the stream window and the id tile; sliced packs decode their stretch of the dictionary over both once the ids
    // are in registers, kRound blocks at a time, and keep the stretch's block records behind them
    constexpr uint32_t kRound = sizeof(U) == 8 ? 6 : 9;
    constexpr uint32_t kMainLds = kWinAlloc + 2 * kRleTile;
    static_assert(kRound * kBlock * sizeof(U) <= kMainLds && kWinAlloc % 16 == 0, "a round's entries fit the window and the id tile");
    __shared__ __align__(16) uint8_t smem[4][kMainLds + 32 * kSliceBlocks];
    TileCtx t;
    if (!tile_ctx(t, packs, list, metas, K_DICT_NUM, 0)) return;
    if (dict_in_lds<U>(t.ec)) return;  // k_dict_lds takes this pack
    const DevPack& P = packs[t.pi];
    const PackMeta& M = metas[t.pi];
    const bool sliced = __builtin_amdgcn_readfirstlane((int)M.sliced) != 0;
    const uint8_t* base = in + P.in_off;
    const uint32_t psz = (uint32_t)P.pack_size, ec = t.ec;
    uint8_t* wl = smem[threadIdx.x >> 6];
    uint16_t* ids = (uint16_t*)(wl + kWinAlloc);
    U* __restrict__ dst = (U*)(out + P.out_off) + t.tile_start;
    const uint32_t cnt = t.tile_end - t.tile_start, lane = threadIdx.x & 63;
    const uint4 seg = ((const uint4*)(scratch + P.s_restart))[t.tile];

    if (!sliced) {
        const U* __restrict__ dict = (const U*)(scratch + P.s_dict);
        const Warm warm = warm_issue(scratch + P.s_dict, ec * (uint32_t)sizeof(U), t.tile >> 2, gridDim.x, ec, t.n);
        Win w;
        w.init(base, psz, wl);
        rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, seg, t.tile_start, t.tile_end, ids);
        // a lane takes two neighbouring rows of every 128: ids come out of LDS two at a time, results leave as 8 / 16 byte
        // stores, and the 16 gathers are all in flight before the first store
        uint32_t id[16];
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t r = q * 128 + 2 * lane;
            const uint32_t two = ((const uint32_t*)ids)[q * 64 + lane];
            id[2 * q] = r < cnt ? (two & 0xFFFFu) : 0u;
            id[2 * q + 1] = r + 1 < cnt ? (two >> 16) : 0u;
            if (id[2 * q] >= ec) { bad = true; id[2 * q] = 0; }
            if (id[2 * q + 1] >= ec) { bad = true; id[2 * q + 1] = 0; }
        }
        U v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = dict[id[q]];
        warm_done(warm);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t r = q * 128 + 2 * lane;
            // streaming (non-temporal) stores: the rows are not read again here and must not push the dictionary out of L2
            if (r + 1 < cnt) store_pair<U>(dst + r, v[2 * q], v[2 * q + 1]);
            else if (r < cnt) dst[r] = v[2 * q];
        }
        if (bad) metas[t.pi].status = E_CORRUPT;
        return;
    }

    // ---- sliced pack.  The stretch: kSliceBlocks blocks from the block of the tile's first id on; its records are requested first
    const uint32_t nblk = (ec + kBlock - 1) / kBlock;  // >= 1: sliced packs have more than 64 KB of entries
    const uint4* __restrict__ recs = (const uint4*)(scratch + P.s_dict);  // two uint4 per block, nblk blocks
    const uint32_t first = min(uniform_load((const uint32_t*)(scratch + P.s_dict + kSliceFirstOff) + t.tile), ec - 1u);
    const uint32_t bA = first >> 7, nb = min(kSliceBlocks, nblk - bA);  // bA < nblk, nb >= 1
    uint4 myrec = make_uint4(0, 0, 0, 0);
    if (lane < 2u * nb) myrec = recs[2u * bA + lane];  // 2 * bA + lane < 2 * nblk
    uint32_t id[16];
    bool bad = false;
    {
        Win w;
        w.init(base, psz, wl);
        rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, seg, t.tile_start, t.tile_end, ids);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t r = q * 128 + 2 * lane;
            const uint32_t two = ((const uint32_t*)ids)[q * 64 + lane];
            id[2 * q] = r < cnt ? (two & 0xFFFFu) : 0xFFFFFFFFu;      // rows past the tile's end take part in nothing
            id[2 * q + 1] = r + 1 < cnt ? (two >> 16) : 0xFFFFFFFFu;
            if (r < cnt && id[2 * q] >= ec) { bad = true; id[2 * q] = 0xFFFFFFFFu; }
            if (r + 1 < cnt && id[2 * q + 1] >= ec) { bad = true; id[2 * q + 1] = 0xFFFFFFFFu; }
        }
    }
    U* slice = (U*)wl;
    uint4* lrec = (uint4*)(wl + kMainLds);  // [2 * kSliceBlocks]
    if (lane < 2u * kSliceBlocks) lrec[lane] = myrec;
    Win::wave_lds_sync();  // window and id tile are dead (their memory becomes the dictionary stretch); the records are in place
    const uint32_t e0 = 2 * lane;  // this lane's two entries of every block
    // the packed bits of the stretch: one 8-byte load per lane and block, all issued before any is used
    uint64_t bits[kSliceBlocks];
#pragma unroll
    for (uint32_t k = 0; k < kSliceBlocks; k++) {
        bits[k] = 0;
        if (k < nb) {
            const uint32_t word = ((const uint32_t*)(lrec + 2 * k))[lane >> 4];
            const uint32_t bw = word >> 24, in_blk = min((uint32_t)kBlock, ec - (bA + k) * kBlock);
            if (e0 < in_blk && bw) {
                // inside the pack (the framing walk checked the mini-block) or, for a block's last entries, its 16 bytes of slack
                const uint32_t at = min((word & 0xFFFFFFu) + (((e0 & 31u) * bw) >> 3), psz);
                __builtin_memcpy(&bits[k], base + at, 8);
            }
        }
    }
    uint32_t left = 0;  // bit q: row q of this lane is still to do
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (id[q] != 0xFFFFFFFFu) left |= 1u << q;
    // rows [c0, c1) of the dictionary are in LDS from slice[0] on: serve the rows that ask for them
    auto serve = [&](uint32_t c0, uint32_t c1) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t r = q * 128 + 2 * lane;
            const uint32_t ia = id[2 * q], ib = id[2 * q + 1];
            const bool ina = ((left >> (2 * q)) & 1u) && ia >= c0 && ia < c1, inb = ((left >> (2 * q + 1)) & 1u) && ib >= c0 && ib < c1;
            const U va = ina ? slice[ia - c0] : (U)0, vb = inb ? slice[ib - c0] : (U)0;
            if (ina && inb) store_pair<U>(dst + r, va, vb);
            else {
                if (ina) dst[r] = va;
                if (inb) dst[r + 1] = vb;
            }
            if (ina) left &= ~(1u << (2 * q));
            if (inb) left &= ~(2u << (2 * q));
        }
    };
    // ---- decode kRound blocks at a time: + minimum, scan from the block's own value, into LDS; then the lookups
#pragma unroll
    for (uint32_t k0 = 0; k0 < kSliceBlocks; k0 += kRound) {
        if (k0 < nb) {
            if (k0) Win::wave_lds_sync();  // the round before has been read
#pragma unroll
            for (uint32_t k = k0; k < k0 + kRound && k < kSliceBlocks; k++) {
                if (k < nb) {
                    const uint4 r1 = lrec[2 * k + 1];
                    const uint32_t word = ((const uint32_t*)(lrec + 2 * k))[lane >> 4];
                    const uint32_t bw = min(word >> 24, kSliceMaxBw), in_blk = min((uint32_t)kBlock, ec - (bA + k) * kBlock);
                    const U md = (U)((uint64_t)r1.x | ((uint64_t)r1.y << 32)), pv = (U)((uint64_t)r1.z | ((uint64_t)r1.w << 32));
                    U a0, a1;
                    slice_two<U>(bits[k], ((e0 & 31u) * bw) & 7u, bw, md, e0 < in_blk, e0 + 1 < in_blk, a0, a1);
                    const U two = a0 + a1;
                    const U before = pv + wave_incl_scan<U>(two) - two;
                    U* d = slice + (k - k0) * kBlock + e0;
                    d[0] = before + a0;
                    d[1] = before + two;
                }
            }
            Win::wave_lds_sync();
            serve((bA + k0) * kBlock, (bA + min(k0 + kRound, nb)) * kBlock);
        }
    }
    // ---- rows outside the stretch: the block of the smallest id still to do, decoded into the first 128 entries of the LDS
    // stretch, serves every row that asks for it; at most one round per block of the dictionary
    while (__builtin_amdgcn_ballot_w64(left != 0)) {
        uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 16; q++)
            if ((left >> q) & 1u) lo = min(lo, id[q]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lo = min(lo, (uint32_t)__shfl_xor((int)lo, d));
        const uint32_t b = min((uint32_t)__builtin_amdgcn_readfirstlane(lo) >> 7, nblk - 1u);
        const uint4 r0 = recs[2u * b], r1 = recs[2u * b + 1u];
        const uint32_t wsel = lane >> 4;
        const uint32_t word = wsel == 0 ? r0.x : (wsel == 1 ? r0.y : (wsel == 2 ? r0.z : r0.w));
        const uint32_t bw = min(word >> 24, kSliceMaxBw), in_blk = min((uint32_t)kBlock, ec - b * kBlock);
        uint64_t w = 0;
        if (e0 < in_blk && bw) {
            const uint32_t at = min((word & 0xFFFFFFu) + (((e0 & 31u) * bw) >> 3), psz);
            __builtin_memcpy(&w, base + at, 8);
        }
        const U md = (U)((uint64_t)r1.x | ((uint64_t)r1.y << 32)), pv = (U)((uint64_t)r1.z | ((uint64_t)r1.w << 32));
        U a0, a1;
        slice_two<U>(w, ((e0 & 31u) * bw) & 7u, bw, md, e0 < in_blk, e0 + 1 < in_blk, a0, a1);
        const U two = a0 + a1;
        const U before = pv + wave_incl_scan<U>(two) - two;
        Win::wave_lds_sync();  // the round before has been read
        slice[e0] = before + a0;
        slice[e0 + 1] = before + two;
        Win::wave_lds_sync();
        serve(b * kBlock, b * kBlock + kBlock);  // the row with the smallest id is among them: every round resolves a row
    }
    if (bad) metas[t.pi].status = E_CORRUPT;
}

// ---------------------------------------------------------------------------------------------------------------
// k_dict_lds: numeric dictionary packs whose dictionary is 2 to 64 KB (kDictLdsMinBytes .. kDictLdsBytes; about 129 KB of static
// LDS per workgroup with the 16 id tiles and stream windows).  A kernel that gathers from memory and stores runs at the
// speed of a copy, however well the gathers hit the caches: loads and stores share the vector memory path.  Lookups out of
// LDS leave that path to the id windows and the stores.  One workgroup of 16 waves per pack: the dictionary is copied
// to LDS once (coalesced), then every wave takes every 16th tile: expands its ids and looks its rows up there.
// (DictStruct.decodeInts/Longs/Floats/Doubles, indexr-common/.../data/DictStruct.java:57-95.)
// ---------------------------------------------------------------------------------------------------------------
using SmallWin = WinT<2048>;
template <typename U>
__global__ __launch_bounds__(1024) void k_dict_lds(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                   PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                   uint8_t* __restrict__ out, const uint8_t* __restrict__ scratch) {
    __shared__ __align__(16) uint8_t s_dict[kDictLdsBytes];
    __shared__ uint16_t s_ids[16][kRleTile];
    __shared__ __align__(16) uint8_t s_win[16][SmallWin::kAlloc];
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    const PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_NUM) return;
    const uint32_t ec = M.entry_count;
    if (!dict_in_lds<U>(ec)) return;  // k_dict_num takes this pack
    const DevPack& P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // the dictionary: 16-byte pieces (the scratch dictionary starts 16-byte aligned and has 16 spare bytes behind it)
        const uint4* src = (const uint4*)(scratch + P.s_dict);
        const uint32_t pieces = (ec * (uint32_t)sizeof(U) + 15u) / 16u;
        for (uint32_t i = tid; i < pieces; i += 1024) ((uint4*)s_dict)[i] = src[i];
    }
    __syncthreads();
    const U* dict = (const U*)s_dict;
    uint16_t* ids = s_ids[wave];
    SmallWin w;
    w.init(in + P.in_off, (uint32_t)P.pack_size, s_win[wave]);
    bool bad = false;
    // the pack's tiles, 16 at a time; a wave's tiles are its own business from here on (no barrier)
    for (uint32_t tile = wave; tile * (uint32_t)kRleTile < n; tile += 16) {
        const uint32_t tile_start = tile * (uint32_t)kRleTile;
        const uint32_t tile_end = min(n, tile_start + (uint32_t)kRleTile), cnt = tile_end - tile_start;
        U* __restrict__ dst = (U*)(out + P.out_off) + tile_start;
        const uint4 seg = ((const uint4*)(scratch + P.s_restart))[tile];
        SmallWin::wave_lds_sync();  // the tile before has been read
        rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, seg, tile_start, tile_end, ids);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t r = q * 128 + 2 * lane;
            const uint32_t two = ((const uint32_t*)ids)[q * 64 + lane];
            uint32_t i0 = r < cnt ? (two & 0xFFFFu) : 0u, i1 = r + 1 < cnt ? (two >> 16) : 0u;
            if (i0 >= ec) { bad = true; i0 = 0; }
            if (i1 >= ec) { bad = true; i1 = 0; }
            const U v0 = dict[i0], v1 = dict[i1];
            // streaming (non-temporal) stores: the rows are not read again here
            if (r + 1 < cnt) {
                if constexpr (sizeof(U) == 4) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 t = {(uint32_t)v0, (uint32_t)v1};
                    __builtin_nontemporal_store(t, (u32x2*)(dst + r));
                } else {
                    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                    u64x2 t = {(uint64_t)v0, (uint64_t)v1};
                    __builtin_nontemporal_store(t, (u64x2*)(dst + r));
                }
            } else if (r < cnt) dst[r] = v0;
        }
    }
    if (bad) metas[pi].status = E_CORRUPT;
}

// ---------------------------------------------------------------------------------------------------------------
// k_copy_raw: packs stored uncompressed (DataPack.java:180-182): data = cmpData.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy_raw(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                  const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                  uint8_t* __restrict__ out) {
    const int pi = list[blockIdx.y];
    const PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_RAW) return;
    const DevPack P = packs[pi];
    const uint32_t bytes = (uint32_t)P.pack_size;
    const uint4* src = (const uint4*)(in + P.in_off);
    uint4* dst = (uint4*)(out + P.out_off);
    const uint32_t nvec = bytes >> 4;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) {
        const uint32_t o = (nvec << 4) + threadIdx.x;
        out[P.out_off + o] = in[P.in_off + o];
    }
}

// ===============================================================================================================
// strings
// ===============================================================================================================

// Generic workgroup exclusive scan over `n` uint32 produced by `get(i)`, handing (i, exclusive prefix, value) to `put`;
// returns the grand total to every thread.  256 threads, 4 items per thread and iteration.
template <typename G, typename Pf>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t n, G get, Pf put) {
    __shared__ uint32_t wsum[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry = 0;
    int buf = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 1024, buf ^= 1) {
        const uint32_t i0 = c0 + 4 * tid;
        uint32_t v0 = i0 < n ? get(i0) : 0, v1 = i0 + 1 < n ? get(i0 + 1) : 0, v2 = i0 + 2 < n ? get(i0 + 2) : 0,
                 v3 = i0 + 3 < n ? get(i0 + 3) : 0;
        const uint32_t t = v0 + v1 + v2 + v3;
        const uint32_t incl = wave_incl_scan<uint32_t>(t);
        if (lane == 63) wsum[buf][wave] = incl;
        __syncthreads();
        const uint32_t w0 = wsum[buf][0], w1 = wsum[buf][1], w2 = wsum[buf][2], w3 = wsum[buf][3];
        uint32_t pre = carry + (incl - t);
        if (wave > 0) pre += w0;
        if (wave > 1) pre += w1;
        if (wave > 2) pre += w2;
        if (i0 < n) put(i0, pre, v0);
        if (i0 + 1 < n) put(i0 + 1, pre + v0, v1);
        if (i0 + 2 < n) put(i0 + 2, pre + v0 + v1, v2);
        if (i0 + 3 < n) put(i0 + 3, pre + v0 + v1 + v2, v3);
        carry += w0 + w1 + w2 + w3;
    }
    return carry;
}

__device__ __forceinline__ uint32_t block_max(uint32_t v) {
    __shared__ uint32_t mx[4];
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    if ((threadIdx.x & 63) == 0) mx[threadIdx.x >> 6] = v;
    __syncthreads();
    v = max(max(mx[0], mx[1]), max(mx[2], mx[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ uint32_t block_or(uint32_t v) {
    __shared__ uint32_t ov[4];
    for (int d = 32; d > 0; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d);
    if ((threadIdx.x & 63) == 0) ov[threadIdx.x >> 6] = v;
    __syncthreads();
    v = ov[0] | ov[1] | ov[2] | ov[3];
    __syncthreads();
    return v;
}

// Where a string pack's dictionary (or, for a front-coded pack, the output itself) lives as a StringsStruct.
__device__ __forceinline__ uint8_t* str_dict_ptr(const DevPack& P, const PackMeta& M, uint8_t* out, uint8_t* scratch) {
    return M.kind == K_DELTA_STR ? out + P.out_off : scratch + P.s_dict;
}
__device__ __forceinline__ uint64_t str_dict_cap(const DevPack& P, const PackMeta& M) {
    return M.kind == K_DELTA_STR ? (uint64_t)P.data_size : P.dict_cap;
}

// ---------------------------------------------------------------------------------------------------------------
// k_str_dict_scan: one workgroup per string pack.
//   front-coded entries: len[i] = prefixLen[i] + suffixLen[i];  offsets = exclusive scan(len) -> StringsStruct index,
//                        sufOff = exclusive scan(suffixLen); validates prefixLen[i] <= len[i-1] and the promised sizes.
//   plain entries:       copies the StringsStruct out of the pack (PlainCodec.decodeStrings, PlainCodec.java:73-77) and
//                        validates its offsets.
// (DeltaCodec.decodeStrings sizes, DeltaCodec.java:355-388; StringsStruct.java:12-17)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_str_dict_scan(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                       PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, int dict_only) {
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    PackMeta& M = metas[pi];
    if (M.status < 0 || (M.kind != K_DICT_STR && M.kind != K_DELTA_STR)) return;
    if ((dict_only & 1) && M.kind == K_DELTA_STR) return;  // a front-coded pack's "dictionary" is its output
    const DevPack P = packs[pi];
    const uint8_t* base = in + P.in_off;
    const uint32_t ec = M.entry_count;
    const int tid = threadIdx.x;
    uint8_t* dict = str_dict_ptr(P, M, out, scratch);
    const uint64_t cap = str_dict_cap(P, M);
    uint32_t* doff = (uint32_t*)dict;
    if ((uint64_t)4 * (ec + 1) > cap) { if (tid == 0) M.status = E_OVERFLOW; return; }

    if (M.entries_plain) {
        const uint8_t* src = base + M.entries_off;
        const uint32_t total = ld32u(src + 4ull * ec);
        const uint64_t bytes = 4ull * (ec + 1) + total;
        if (bytes > cap) { if (tid == 0) M.status = E_OVERFLOW; return; }
        uint32_t bad = 0, mx = 0;
        for (uint32_t i = tid; i < ec; i += 256) {
            uint32_t a = ld32u(src + 4ull * i), b = ld32u(src + 4ull * (i + 1));
            if (a > b || b > total) bad = 1; else mx = max(mx, b - a);
        }
        if (ld32u(src) != 0) bad = 1;
        bad = block_or(bad);
        mx = block_max(mx);
        if (bad) { if (tid == 0) M.status = E_CORRUPT; return; }
        // 16 bytes at a time (the scratch dictionary is 16-byte aligned, the entries in the pack are wherever they are), the
        // last bytes one by one
        for (uint64_t i = tid; i < bytes / 16; i += 256) {
            uint4 v;
            __builtin_memcpy(&v, src + 16 * i, 16);
            *(uint4*)(dict + 16 * i) = v;
        }
        for (uint64_t i = (bytes & ~15ull) + tid; i < bytes; i += 256) dict[i] = src[i];
        if (tid == 0) M.max_len = mx;
        return;
    }

    const int32_t* pre = (const int32_t*)(scratch + P.s_lens);
    const int32_t* suf = pre + lens_stride(P.obj_count);
    uint32_t* sufoff = (uint32_t*)(suf + lens_stride(P.obj_count));
    // One pass over the two length lists, 1024 entries at a time, four neighbouring entries per thread: validation
    // (non-negative, the prefix fits into the previous string, lengths bounded), the longest string, the exclusive scans of
    // the lengths (-> dictionary offsets) and of the suffix lengths (-> suffix offsets) with running carries, and this
    // thread's share of the exact 64-bit total.  Offsets of a list that turns out invalid are written all the same (inside
    // the arrays' bounds) and never read: the pack's status says so.
    __shared__ uint2 s_chunk[2][4];
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t bad = 0, mx = 0, total_len = 0, total_suf = 0;
    uint64_t s64 = 0;
    int buf = 0;
    for (uint32_t c0 = 0; c0 < ec; c0 += 1024, buf ^= 1) {
        const uint32_t i0 = c0 + 4u * tid;
        int32_t p[4] = {0, 0, 0, 0}, sl[4] = {0, 0, 0, 0};
        if (i0 + 3 < ec) {
            const int4 pv = *(const int4*)(pre + i0), sv = *(const int4*)(suf + i0);
            p[0] = pv.x; p[1] = pv.y; p[2] = pv.z; p[3] = pv.w;
            sl[0] = sv.x; sl[1] = sv.y; sl[2] = sv.z; sl[3] = sv.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < ec) { p[k] = pre[i0 + k]; sl[k] = suf[i0 + k]; }
        }
        int64_t prevlen = (i0 && i0 < ec) ? (int64_t)pre[i0 - 1] + suf[i0 - 1] : 0;
        uint32_t vl[4], vs[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = i0 + k < ec;
            if (in && (p[k] < 0 || sl[k] < 0 || p[k] > prevlen || (int64_t)p[k] + sl[k] > 0x7fffffff)) bad = 1;
            else if (in) mx = max(mx, (uint32_t)(p[k] + sl[k]));
            prevlen = (int64_t)p[k] + sl[k];
            vl[k] = in ? (uint32_t)(p[k] + sl[k]) : 0u;
            vs[k] = in ? (uint32_t)sl[k] : 0u;
            s64 += vl[k];
        }
        const uint32_t tl = vl[0] + vl[1] + vl[2] + vl[3], ts = vs[0] + vs[1] + vs[2] + vs[3];
        const uint32_t il = wave_incl_scan<uint32_t>(tl), is = wave_incl_scan<uint32_t>(ts);
        if (lane == 63) s_chunk[buf][wave] = make_uint2(il, is);
        __syncthreads();  // one barrier per chunk: the next chunk writes the other half of s_chunk
        uint32_t ol = total_len + il - tl, os = total_suf + is - ts;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint2 t = s_chunk[buf][w];
            if (w < wave) { ol += t.x; os += t.y; }
            total_len += t.x;
            total_suf += t.y;
        }
        if (i0 + 3 < ec) {
            *(uint4*)(doff + i0) = make_uint4(ol, ol + vl[0], ol + vl[0] + vl[1], ol + vl[0] + vl[1] + vl[2]);
            *(uint4*)(sufoff + i0) = make_uint4(os, os + vs[0], os + vs[0] + vs[1], os + vs[0] + vs[1] + vs[2]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k < ec) { doff[i0 + k] = ol; sufoff[i0 + k] = os; }
                ol += vl[k];
                os += vs[k];
            }
        }
    }
    bad = block_or(bad);
    mx = block_max(mx);
    if (bad) { if (tid == 0) M.status = E_CORRUPT; return; }
    // 32-bit wrap of the totals is impossible to hit without also exceeding `cap`: the exact total is a 64-bit reduction
    __shared__ unsigned long long tot64;
    if (tid == 0) tot64 = 0;
    __syncthreads();
    atomicAdd(&tot64, (unsigned long long)s64);
    __syncthreads();
    if (tid == 0) {
        if (tot64 + 4ull * (ec + 1) > cap) M.status = E_OVERFLOW;
        else if (total_suf != M.sfx_len) M.status = E_CORRUPT;
        else {
            doff[ec] = total_len;
            sufoff[ec] = total_suf;
            M.max_len = mx;
            // Block-parallel decode needs a (blocks x columns) uint16 summary; it lives in the pack's own, not yet
            // written, output region.  Front-coded packs (dictionary == output) and oversized cases take the serial path.
            const uint32_t nblk = (ec + kFrontBlock - 1) / kFrontBlock, stride = (mx + 63u) & ~63u;
            if (M.kind == K_DICT_STR && nblk > 1 && 2ull * nblk * stride <= (uint64_t)P.data_size) {
                M.front_blocks = nblk;
                M.front_stride = stride;
            }
        }
    }
    if (M.kind == K_DELTA_STR && tot64 + 4ull * (ec + 1) <= cap) {
        // front-coded pack decoded straight into the output: zero the unused tail (the reference's buffer is zero-filled)
        uint8_t* tail = dict + 4ull * (ec + 1) + tot64;
        const uint64_t rest = cap - (4ull * (ec + 1) + tot64);
        for (uint64_t i = tid; i < rest; i += 256) tail[i] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Front-coded string lists (string i = first prefixLen[i] bytes of string i-1 + its suffix, DeltaCodec.java:368-383).
//
// Column-parallel core: a wave owns 64 byte columns; lane j keeps the current byte of column j in a register while the
// wave steps through strings in order -- the prefix of a string is exactly what the registers already hold, only columns
// covered by the suffix take a new byte.  Nothing flows through memory between iterations.
//
// To decode blocks of kFrontBlock strings independently, each block needs the bytes its first strings inherit:
//   k_front_summary  per block and column: the byte the block's last writer of that column leaves (0xFFFF = none)
//   k_front_combine  per column, in block order: turns the summaries into "state before the block" (in place)
//   k_front_decode   per block: the core loop, started from that state
// The summary array (blocks x columns x uint16) lives in the pack's own output region, which is written only later.
// Lists that do not qualify (front-coded packs, whose dictionary is the output; one-block lists) use k_front_serial.
// ---------------------------------------------------------------------------------------------------------------
struct FrontCtx {
    uint32_t ec, nblk, stride;
    const uint8_t* sfx;
    const uint32_t* doff;
    uint8_t* body;
    const int32_t* pre;
    const int32_t* suf;
    const uint32_t* sufoff;
    uint16_t* sum;
};

// `ro` is the scratch base again, as a read-only view: the length / offset arrays read through it are never written by
// the kernels that use it, which lets the compiler fetch wave-uniform entries with scalar loads.
__device__ __forceinline__ bool front_ctx(FrontCtx& f, const DevPack& P, const PackMeta& M, const uint8_t* in, uint8_t* out,
                                          uint8_t* scratch, const uint8_t* ro, int use_ds, bool parallel) {
    if (M.status < 0 || (M.kind != K_DICT_STR && M.kind != K_DELTA_STR) || M.entries_plain) return false;
    if ((M.front_blocks != 0) != parallel) return false;
    f.ec = M.entry_count;
    f.nblk = M.front_blocks;
    f.stride = M.front_stride;
    f.sfx = in + P.in_off + M.sfx_off;
    uint8_t* dict = str_dict_ptr(P, M, out, scratch);
    f.doff = M.kind == K_DELTA_STR ? (const uint32_t*)dict : (const uint32_t*)(ro + P.s_dict);
    f.body = dict + 4ull * (f.ec + 1);
    f.pre = (const int32_t*)(ro + P.s_lens);
    f.suf = f.pre + lens_stride(P.obj_count);
    f.sufoff = (const uint32_t*)(f.suf + lens_stride(P.obj_count));
    f.sum = (uint16_t*)(out + (use_ds ? P.ds_off : P.out_off));
    return true;
}

__global__ __launch_bounds__(64) void k_front_serial(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                     const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, int dict_only) {
    const int pi = list[blockIdx.y];
    const DevPack P = packs[pi];
    if ((dict_only & 1) && metas[pi].kind == K_DELTA_STR) return;
    FrontCtx f;
    if (!front_ctx(f, P, metas[pi], in, out, scratch, scratch, 0, false)) return;
    const uint32_t max_len = metas[pi].max_len;
    const int lane = threadIdx.x;
    for (uint32_t col0 = blockIdx.x * 64; col0 < max_len; col0 += gridDim.x * 64) {
        const uint32_t j = col0 + lane;
        uint32_t cur = 0;
        for (uint32_t i0 = 0; i0 < f.ec; i0 += 64) {
            const uint32_t ii = min(i0 + lane, f.ec - 1);
            const uint32_t mp = (uint32_t)f.pre[ii], ms = (uint32_t)f.suf[ii], mo = f.doff[ii], mso = f.sufoff[ii];
            const uint32_t cnt = min(64u, f.ec - i0);
            for (uint32_t k = 0; k < cnt; k++) {
                const uint32_t p = __builtin_amdgcn_readlane(mp, k), sl = __builtin_amdgcn_readlane(ms, k);
                const uint32_t len = p + sl;
                if (len <= col0) continue;  // uniform: nothing of this string lies in our columns
                const uint32_t o = __builtin_amdgcn_readlane(mo, k), so = __builtin_amdgcn_readlane(mso, k);
                if (j >= p && j < len) cur = f.sfx[so + (j - p)];
                if (j < len) f.body[o + j] = (uint8_t)cur;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_front_summary(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                       const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ scratch,
                                                       const uint8_t* __restrict__ ro, int use_ds) {
    uint32_t bx, by;
    pack_block_remap(bx, by);
    const int pi = list[by];
    const DevPack& P = packs[pi];
    FrontCtx f;
    if (!front_ctx(f, P, metas[pi], in, out, scratch, ro, use_ds, true)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = bx * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (uint32_t b = wave; b < f.nblk; b += nwaves) {
        const uint32_t i0 = b * kFrontBlock;
        const uint32_t cnt = min((uint32_t)kFrontBlock, f.ec - i0);
        const uint32_t ii = min(i0 + lane, f.ec - 1);
        const uint32_t mp = (uint32_t)f.pre[ii], mlen = mp + (uint32_t)f.suf[ii], mso = f.sufoff[ii];
        // The last writer of column j in this block is the last string k with prefixLen[k] <= j (for every column a later
        // block can still inherit, i.e. j < length of the block's last string).  With M[k] = min(prefixLen[k..]) -- a
        // non-decreasing sequence -- that k is (number of k with M[k] <= j) - 1: a binary search over the lanes.
        uint32_t M = (uint32_t)lane < cnt ? mp : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_down(M, d);
            if (lane + d < 64) M = min(M, t);
        }
        const uint32_t len_last = __builtin_amdgcn_readlane(mlen, cnt - 1);
        for (uint32_t col0 = 0; col0 < f.stride; col0 += 64) {
            const uint32_t j = col0 + lane;
            uint32_t lo = 0;  // number of k with M[k] <= j, found bit by bit
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t probe = lo + step - 1;  // is M[probe] <= j ?
                const uint32_t mv = __shfl(M, (int)min(probe, 63u));
                if (probe < cnt && mv <= j) lo += step;
            }
            // lo in 0..63 from the loop; the 64th string needs one more probe
            if (lo == 63 && cnt == 64 && (uint32_t)__shfl(M, 63) <= j) lo = 64;
            const int owner = (int)lo - 1;
            const int src = owner < 0 ? 0 : owner;
            const uint32_t wp = __shfl(mp, src), wso = __shfl(mso, src);
            uint32_t val = 0xFFFFu;
            if (owner >= 0 && j < len_last) val = f.sfx[wso + (j - wp)];
            f.sum[(size_t)b * f.stride + j] = (uint16_t)val;
        }
    }
}

__global__ __launch_bounds__(64) void k_front_combine(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                      const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                      uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, int use_ds) {
    uint32_t bx, by;
    pack_block_remap(bx, by);
    const int pi = list[by];
    const DevPack P = packs[pi];
    FrontCtx f;
    if (!front_ctx(f, P, metas[pi], in, out, scratch, scratch, use_ds, true)) return;
    const int lane = threadIdx.x;
    for (uint32_t col0 = bx * 64; col0 < f.stride; col0 += gridDim.x * 64) {
        uint16_t* col = f.sum + col0 + lane;
        uint32_t state = 0;
        uint32_t b = 0;
        for (; b + 8 <= f.nblk; b += 8) {
            uint32_t t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) t[q] = col[(size_t)(b + q) * f.stride];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                col[(size_t)(b + q) * f.stride] = (uint16_t)state;
                if (t[q] != 0xFFFFu) state = t[q];
            }
        }
        for (; b < f.nblk; b++) {
            const uint32_t t = col[(size_t)b * f.stride];
            col[(size_t)b * f.stride] = (uint16_t)state;
            if (t != 0xFFFFu) state = t;
        }
    }
}

// Four blocks per wave: 16 lanes per block, a lane owns FOUR neighbouring byte columns as one dword, so a step moves four
// strings of up to 64 bytes with one load and one store per lane.  The blocks' lengths and offsets are staged in LDS (one
// 16-byte record per string, read back as a broadcast within each group of 16 lanes).  Columns 64.. take further rounds.
__device__ __forceinline__ uint32_t byte_mask(uint32_t x) { return x >= 4 ? 0xFFFFFFFFu : ((1u << (8 * x)) - 1u); }
// H rounds of 64 byte columns of one block: lane c of the block's 16 lanes owns the dwords at columns col0 + 64 h + 4 c.
#ifdef IXR_FRONT_EXACT_TAILS  // build switch: every tail byte by byte
constexpr bool kFrontWholeTails = false;
#else
constexpr bool kFrontWholeTails = true;
#endif
// A string's last dword is stored whole where that is safe: the bytes past its end belong to the strings after it, which
// this wave writes later (stores of one wave to the same bytes land in program order) -- as long as they stay inside the
// block (`block_end` = body offset where the block's last string ends; the next block is another wave's) and no string
// has been written ahead of its predecessors' tails (only the first 128 columns, string by string: `whole_tails`).
// Everywhere else the tail goes out byte by byte.
template <int H>
__device__ __forceinline__ void front_rounds(const FrontCtx& f, const uint4* mine, uint32_t b, uint32_t col0, int c, uint32_t block_end,
                                             bool whole_tails) {
    constexpr int kB = IXR_FRONT_BATCH;  // strings per step and group
    uint32_t j0[H], cur[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
        j0[h] = col0 + 64 * h + 4 * c;
        cur[h] = 0;
        if (b < f.nblk) {  // the columns' bytes before the block (one uint16 each)
            const uint2 st = *(const uint2*)(f.sum + (size_t)b * f.stride + j0[h]);
            cur[h] = (st.x & 0xFFu) | ((st.x >> 8) & 0xFF00u) | ((st.y & 0xFFu) << 16) | ((st.y << 8) & 0xFF000000u);
        }
    }
    for (uint32_t k0 = 0; k0 < (uint32_t)kFrontBlock; k0 += kB) {
        uint4 m[kB];
        bool reach = false;
#pragma unroll
        for (int q = 0; q < kB; q++) {
            m[q] = mine[k0 + q];
            reach |= m[q].y > j0[0];
        }
        if (!__any(reach)) continue;  // strings that end before these columns neither read nor write here
        uint32_t t[H][kB];
#pragma unroll
        for (int h = 0; h < H; h++) {
#pragma unroll
            for (int q = 0; q < kB; q++) {
                // unconditional, clamped load: where no column takes a new byte read the string's first suffix byte
                // position (always readable: the suffix bytes are followed by the id stream and the input slack)
                const bool fresh = j0[h] + 4 > m[q].x && j0[h] < m[q].y;
                const int32_t at = (int32_t)(m[q].w + (fresh ? j0[h] : m[q].x));  // may lie up to 3 bytes before the suffix: pack header
                __builtin_memcpy(&t[h][q], f.sfx + at, 4);
            }
        }
        // keep the loads together: without this the compiler sinks some of them into conditional blocks behind a full
        // wait, which costs further memory round trips per batch
#pragma unroll
        for (int h = 0; h < H; h++)
#pragma unroll
            for (int q = 0; q < kB; q += 8)
                asm volatile("" : "+v"(t[h][q]), "+v"(t[h][q + 1]), "+v"(t[h][q + 2]), "+v"(t[h][q + 3]), "+v"(t[h][q + 4]), "+v"(t[h][q + 5]), "+v"(t[h][q + 6]), "+v"(t[h][q + 7]));
#pragma unroll
        for (int q = 0; q < kB; q++) {  // string by string, so that a string's tail is stored before anything of the next one
#pragma unroll
            for (int h = 0; h < H; h++) {
                const uint32_t p = m[q].x, len = m[q].y;
                const uint32_t lo = p > j0[h] ? min(p - j0[h], 4u) : 0u, hi = len > j0[h] ? min(len - j0[h], 4u) : 0u;
                const uint32_t take = byte_mask(hi) & ~byte_mask(lo);  // columns inside the suffix
                cur[h] = (cur[h] & ~take) | (t[h][q] & take);
                const uint32_t at = m[q].z + j0[h];
                uint8_t* dst = f.body + at;
                if (hi == 4 || (hi && kFrontWholeTails && whole_tails && at + 4 <= block_end)) __builtin_memcpy(dst, &cur[h], 4);
                else if (hi) {
                    dst[0] = (uint8_t)cur[h];
                    if (hi > 1) dst[1] = (uint8_t)(cur[h] >> 8);
                    if (hi > 2) dst[2] = (uint8_t)(cur[h] >> 16);
                }
            }
        }
    }
}

// The same for 128 byte columns at a time with EIGHT columns per lane (one 8-byte load and one 8-byte store per lane and
// string instead of two of four bytes each): lists whose strings are longer than 64 bytes take their columns in such rounds.
__device__ __forceinline__ uint64_t byte_mask64(uint32_t x) { return x >= 8 ? ~0ull : ((1ull << (8 * x)) - 1ull); }
__device__ __forceinline__ void front_rounds64(const FrontCtx& f, const uint4* mine, uint32_t b, uint32_t col0, int c, uint32_t block_end,
                                               bool whole_tails) {
    constexpr int kB = IXR_FRONT_BATCH;
    const uint32_t j0 = col0 + 8 * c;
    uint64_t cur = 0;
    if (b < f.nblk) {  // the columns' bytes before the block (one uint16 each)
        const uint4 st = *(const uint4*)(f.sum + (size_t)b * f.stride + j0);
        const uint32_t lo = (st.x & 0xFFu) | ((st.x >> 8) & 0xFF00u) | ((st.y & 0xFFu) << 16) | ((st.y << 8) & 0xFF000000u);
        const uint32_t hi = (st.z & 0xFFu) | ((st.z >> 8) & 0xFF00u) | ((st.w & 0xFFu) << 16) | ((st.w << 8) & 0xFF000000u);
        cur = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    for (uint32_t k0 = 0; k0 < (uint32_t)kFrontBlock; k0 += kB) {
        uint4 m[kB];
        bool reach = false;
#pragma unroll
        for (int q = 0; q < kB; q++) {
            m[q] = mine[k0 + q];
            reach |= m[q].y > j0;
        }
        if (!__any(reach)) continue;  // strings that end before these columns neither read nor write here
        uint64_t t[kB];
#pragma unroll
        for (int q = 0; q < kB; q++) {
            // unconditional, clamped load (see front_rounds); may start up to 7 bytes before the suffix bytes: pack header
            const bool fresh = j0 + 8 > m[q].x && j0 < m[q].y;
            const int32_t at = (int32_t)(m[q].w + (fresh ? j0 : m[q].x));
            __builtin_memcpy(&t[q], f.sfx + at, 8);
        }
        asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
        static_assert(kB == 8, "the loads of a batch are kept together eight at a time");
#pragma unroll
        for (int q = 0; q < kB; q++) {  // string by string, so that a string's tail is stored before anything of the next one
            const uint32_t p = m[q].x, len = m[q].y;
            const uint32_t lo = p > j0 ? min(p - j0, 8u) : 0u, hi = len > j0 ? min(len - j0, 8u) : 0u;
            const uint64_t take = byte_mask64(hi) & ~byte_mask64(lo);  // columns inside the suffix
            cur = (cur & ~take) | (t[q] & take);
            const uint32_t at = m[q].z + j0;
            uint8_t* dst = f.body + at;
            if (hi == 8 || (hi && kFrontWholeTails && whole_tails && at + 8 <= block_end)) __builtin_memcpy(dst, &cur, 8);
            else if (hi) {
                uint32_t w = (uint32_t)cur, r = hi;
                if (hi >= 4) { __builtin_memcpy(dst, &w, 4); dst += 4; w = (uint32_t)(cur >> 32); r = hi - 4; }
                if (r > 0) dst[0] = (uint8_t)w;
                if (r > 1) dst[1] = (uint8_t)(w >> 8);
                if (r > 2) dst[2] = (uint8_t)(w >> 16);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_front_decode(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                      const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                      uint8_t* __restrict__ out, uint8_t* __restrict__ scratch,
                                                      const uint8_t* __restrict__ ro, int use_ds) {
    __shared__ uint4 s_meta[4][4 * kFrontBlock];  // per wave: (prefixLen, length, body offset, suffix position - prefixLen)
    __shared__ uint32_t s_end[4][4];              // per wave and block: body offset where the block's last string ends
    uint32_t bx, by;
    pack_block_remap(bx, by);
    const int pi = list[by];
    const DevPack& P = packs[pi];
    FrontCtx f;
    if (!front_ctx(f, P, metas[pi], in, out, scratch, ro, use_ds, true)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const uint32_t wave = bx * 4 + wv, nwaves = gridDim.x * 4;
    const uint32_t ntask = (f.nblk + 3) / 4;
    uint4* meta = s_meta[wv];
    for (uint32_t task = wave; task < ntask; task += nwaves) {
        const uint32_t b0 = task * 4;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the round before has read its records
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = (b0 + q) * kFrontBlock + lane;
            const uint32_t ii = min(i, f.ec - 1);
            const uint32_t p = (uint32_t)f.pre[ii];
            const uint32_t len = i < f.ec ? p + (uint32_t)f.suf[ii] : 0u;  // length 0: no column takes part
            const uint32_t o = f.doff[ii];
            meta[q * kFrontBlock + lane] = make_uint4(p, len, o, f.sufoff[ii] - p);
            // strings past the list's end have length 0 and the last string's offset: the maximum is the block's end
            uint32_t e = o + len;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) e = max(e, (uint32_t)__shfl_xor(e, d));
            if (lane == 0) s_end[wv][q] = e;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t b = b0 + g;
        const uint4* mine = meta + g * kFrontBlock;
        const uint32_t block_end = s_end[wv][g];
        // two rounds of 64 columns at a time where there are that many (a lane owns dword c of both): their loads are in
        // flight together and the strings' records are read once for both
        for (uint32_t col0 = 0; col0 < f.stride; col0 += 128) {
#ifdef IXR_FRONT_DWORD_PAIRS
            if (col0 + 64 < f.stride) front_rounds<2>(f, mine, b, col0, c, block_end, col0 == 0 && f.stride <= 128);  // wave-uniform
#else
            if (col0 + 64 < f.stride) front_rounds64(f, mine, b, col0, c, block_end, col0 == 0 && f.stride <= 128);  // wave-uniform
#endif
            else front_rounds<1>(f, mine, b, col0, c, block_end, col0 == 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_row_scan: one workgroup per string dictionary pack: turns the row lengths in the output index into offsets
// (offset[0] = 0, offset[i+1] = offset[i] + len[i]; StringsStruct.Builder, StringsStruct.java:120-158), checks that the
// body fits the output.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_scan(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                  PackMeta* __restrict__ metas, uint8_t* __restrict__ out) {
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_STR) return;
    const DevPack P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count;
    uint32_t* off = (uint32_t*)(out + P.out_off);
    if (4ull * (n + 1) > (uint64_t)P.data_size) { if (threadIdx.x == 0) M.status = E_OVERFLOW; return; }
    const uint32_t body_cap = (uint32_t)P.data_size - 4u * (n + 1);
    // Any row length is < 2^31 and validated dictionary slices are <= data_size, so a total beyond body_cap is detected
    // from the 64-bit sum; the in-place scan itself may wrap harmlessly in that case because the copy is skipped.
    uint64_t s64 = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) s64 += off[i + 1];
    __shared__ unsigned long long tot64;
    if (threadIdx.x == 0) tot64 = 0;
    __syncthreads();
    atomicAdd(&tot64, (unsigned long long)s64);
    __syncthreads();
    if (tot64 > body_cap) {  // the body must fit dpn.dataSize = index + body (DataPackCreator_VLT.java:185-205)
        if (threadIdx.x == 0) M.status = E_OVERFLOW;
        return;
    }
    block_excl_scan(
        n, [&](uint32_t i) { return off[i + 1]; },
        [&](uint32_t i, uint32_t ex, uint32_t v) { off[i + 1] = ex + v; });
    if (threadIdx.x == 0) off[0] = 0;
    // A body shorter than dpn.dataSize promises leaves a tail: the reference decodes into a zero-filled direct buffer
    // (PackCompressor_VLT.java:133, ByteBufferUtil.java:31-36), so the tail reads as zeros.
    uint8_t* tail = out + P.out_off + 4ull * (n + 1) + tot64;
    for (uint64_t i = threadIdx.x; i < body_cap - tot64; i += 256) tail[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// k_str_index: string dictionary packs, one workgroup of 4 or 16 waves per pack (see the launcher).  Per round each wave expands one tile of
// kRleTile ids, looks up the rows' dictionary slices and scans the row lengths; the workgroup chains the tiles and the
// rounds with a running total.  Writes the output index (offset[0] = 0, offset[i+1] = offset[i] + len[i];
// StringsStruct.Builder, StringsStruct.java:120-158) and each row's slice start into scratch, checks that the body fits,
// zero-fills what the body leaves of dpn.dataSize.  (DictStruct.decodeStrings, DictStruct.java:97-110.)
// ---------------------------------------------------------------------------------------------------------------
template <int kWaves>
__global__ __launch_bounds__(kWaves * 64) void k_str_index(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                    PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                    uint8_t* __restrict__ out, uint8_t* __restrict__ scratch) {
    __shared__ uint16_t s_ids[kWaves][kRleTile];
    __shared__ __align__(16) uint8_t s_win[kWaves][kWinAlloc];
    __shared__ uint32_t s_wsum[2][kWaves];
    __shared__ unsigned long long s_tot;
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_STR) return;
    const DevPack& P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count, ec = M.entry_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (4ull * (n + 1) > (uint64_t)P.data_size) { if (tid == 0) M.status = E_OVERFLOW; return; }
    const uint32_t body_cap = (uint32_t)P.data_size - 4u * (n + 1);
    const uint8_t* base = in + P.in_off;
    const uint32_t* __restrict__ doff = (const uint32_t*)(scratch + P.s_dict);
    // The reference decodes the DictStruct into a buffer of the dictStructSize the pack announces and fails when it does not
    // fit (DictCompressCodec.decode, DictCompressCodec.java:71-85; a larger buffer is accepted).  entryCount | offsets | bytes | 4n | ids
    if (M.dict_struct_size && 4ull + 4ull * (ec + 1) + doff[ec] + 4ull + 4ull * n > (uint64_t)M.dict_struct_size) {
        if (tid == 0) M.status = E_OVERFLOW;
        return;
    }
    uint32_t* __restrict__ off = (uint32_t*)(out + P.out_off);
    uint32_t* __restrict__ srcoff = (uint32_t*)(scratch + P.s_rows);
    const uint4* restarts = (const uint4*)(scratch + P.s_restart);
    Win w;
    w.init(base, (uint32_t)P.pack_size, s_win[wave]);
    const uint32_t id_bw = M.id_bw, ids_end = M.ids_off + 4 + M.ids_len;
    uint16_t* ids = s_ids[wave];
    __shared__ int s_bad;
    if (tid == 0) { s_tot = 0; s_bad = 0; }
    const uint32_t ntiles = (n + kRleTile - 1) / kRleTile;
    uint32_t carry = 0;   // offset of the round's first row (mod 2^32; the exact total is kept in sum64)
    uint64_t sum64 = 0;   // this lane's share of the exact total
    bool bad = false;
    for (uint32_t t0 = 0, round = 0; t0 < ntiles; t0 += kWaves, round++) {
        const uint32_t tile = t0 + wave;
        const uint32_t tile_start = min(tile * (uint32_t)kRleTile, n), tile_end = min(n, tile_start + (uint32_t)kRleTile);
        const uint32_t cnt = tile_end - tile_start;  // 0 for waves past the last tile
        if (cnt) rle_tile_expand(w, id_bw, ids_end, restarts[tile], tile_start, tile_end, ids);
        // a lane takes four neighbouring rows of every 256: ids come out of LDS four at a time, the index and the source
        // offsets leave as 16-byte stores
        uint2 ab[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = q * 256 + 4 * lane;
            const uint2 four = ((const uint2*)ids)[q * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t id = ((g < 2 ? four.x : four.y) >> (16 * (g & 1))) & 0xFFFFu;
                if (r + g >= cnt) id = 0;
                if (id >= ec) { bad = true; id = 0; }
                __builtin_memcpy(&ab[q * 4 + g], doff + id, 8);  // doff[id], doff[id + 1]
            }
        }
        // scan each group of 256 rows across the wave (four rows per lane), chain the groups
        uint32_t ex[16], run = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = q * 256 + 4 * lane;
            uint32_t len[4], s4 = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                len[g] = r + g < cnt ? ab[q * 4 + g].y - ab[q * 4 + g].x : 0u;
                ex[q * 4 + g] = s4;
                s4 += len[g];
            }
            sum64 += s4;
            const uint32_t inc = wave_incl_scan(s4);
            const uint32_t first = run + inc - s4;
#pragma unroll
            for (int g = 0; g < 4; g++) ex[q * 4 + g] += first;
            run += __builtin_amdgcn_readlane(inc, 63);
        }
        if (lane == 0) s_wsum[round & 1][wave] = run;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint32_t v = s_wsum[round & 1][w];
            before += w < wave ? v : 0u;
            total += v;
        }
        const uint32_t o0 = carry + before;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = q * 256 + 4 * lane;
            if (r + 3 < cnt) {
                *(uint4*)(off + tile_start + r) = make_uint4(o0 + ex[q * 4], o0 + ex[q * 4 + 1], o0 + ex[q * 4 + 2], o0 + ex[q * 4 + 3]);
                *(uint4*)(srcoff + tile_start + r) = make_uint4(ab[q * 4].x, ab[q * 4 + 1].x, ab[q * 4 + 2].x, ab[q * 4 + 3].x);
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    if (r + g < cnt) { off[tile_start + r + g] = o0 + ex[q * 4 + g]; srcoff[tile_start + r + g] = ab[q * 4 + g].x; }
            }
        }
        carry += total;
    }
    // exact total: detects a body that does not fit even when the 32-bit offsets wrapped
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum64 += __shfl_xor(sum64, d);
    if (lane == 0) atomicAdd(&s_tot, (unsigned long long)sum64);
    if (bad) s_bad = 1;
    __syncthreads();
    const uint64_t tot64 = s_tot;
    if (s_bad) { if (tid == 0) M.status = E_CORRUPT; return; }
    if (tot64 > body_cap) {  // the body must fit dpn.dataSize = index + body (DataPackCreator_VLT.java:185-205)
        if (tid == 0) M.status = E_OVERFLOW;
        return;
    }
    if (tid == 0) off[n] = (uint32_t)tot64;
    // A body shorter than dpn.dataSize promises leaves a tail: the reference decodes into a zero-filled direct buffer
    // (PackCompressor_VLT.java:133, ByteBufferUtil.java:31-36), so the tail reads as zeros.
    uint8_t* tail = out + P.out_off + 4ull * (n + 1) + tot64;
    for (uint64_t i = tid; i < body_cap - tot64; i += kWaves * 64) tail[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// k_str_bases: what k_str_fused needs before it can start: the body offset of every 1024-row tile of a string dictionary
// pack (the sum of the row lengths of the tiles before it), the checks k_str_index makes, offset[n] and the zero tail.
// One workgroup of 16 waves per pack.  The dictionary's string lengths are first put into LDS, one byte each when no
// string is longer than 255 bytes, else two bytes, so that a row costs an LDS read instead of a gather from global memory
// (dictionaries whose lengths do not fit 64 KB that way keep the gathers).  A wave then takes every 16th tile on its own:
// ids -> lengths -> one sum per tile; no barrier until all tiles are summed.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kBasesWaves = 16, kBasesTab = 65536;
// Two builds: with a 64 KB length table the kernel's LDS (table, sixteen id tiles and stream windows) lets one workgroup live
// on a compute unit; dictionaries whose lengths fit 12 KB (most: 12288 entries of up to 255 bytes) take the build with that
// table, of which two fit.  Every pack is taken by exactly one of the two launches.
constexpr uint32_t kBasesTabSmall = 12288;
using BasesWin = WinT<2048>;
template <uint32_t kTab>
__global__ __launch_bounds__(kBasesWaves * 64, kTab == kBasesTabSmall ? 8 : 4) void k_str_bases(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                                PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                                uint8_t* __restrict__ out, uint8_t* __restrict__ scratch) {
    __shared__ __align__(16) uint8_t s_tab[kTab];
    __shared__ __align__(16) uint8_t s_win[kBasesWaves][BasesWin::kAlloc];
    __shared__ uint16_t s_ids[kBasesWaves][kRleTile];
    __shared__ unsigned long long s_tile[IXR_PACK_TILES];
    __shared__ int s_bad;
    const int pi = list[pack_slot_remap(blockIdx.x, gridDim.x)];
    PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_STR) return;
    const DevPack& P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count, ec = M.entry_count;
    const uint32_t elem = M.max_len <= 255u ? 1u : 2u;
    if (((uint64_t)ec * elem <= kBasesTabSmall) != (kTab == kBasesTabSmall)) return;  // the other build's pack
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (4ull * (n + 1) > (uint64_t)P.data_size) { if (tid == 0) M.status = E_OVERFLOW; return; }
    const uint32_t body_cap = (uint32_t)P.data_size - 4u * (n + 1);
    const uint32_t* __restrict__ doff = (const uint32_t*)(scratch + P.s_dict);
    // The reference decodes the DictStruct into a buffer of the dictStructSize the pack announces and fails when it does not
    // fit (DictCompressCodec.decode, DictCompressCodec.java:71-85; a larger buffer is accepted).  entryCount | offsets | bytes | 4n | ids
    if (M.dict_struct_size && 4ull + 4ull * (ec + 1) + doff[ec] + 4ull + 4ull * n > (uint64_t)M.dict_struct_size) {
        if (tid == 0) M.status = E_OVERFLOW;
        return;
    }
    const uint32_t ntiles = (n + kRleTile - 1) / kRleTile;
    if (tid == 0) s_bad = 0;
    // ---- the length table
    const bool tab = ec > 0 && (uint64_t)ec * elem <= kTab && M.max_len <= 65535u;
    if (tab) {
        for (uint32_t i = tid; i < ec; i += kBasesWaves * 64) {
            const uint32_t len = doff[i + 1] - doff[i];
            if (elem == 1) s_tab[i] = (uint8_t)len;
            else ((uint16_t*)s_tab)[i] = (uint16_t)len;
        }
    }
    __syncthreads();
    // ---- one sum per tile
    const uint4* restarts = (const uint4*)(scratch + P.s_restart);
    BasesWin w;
    w.init(in + P.in_off, (uint32_t)P.pack_size, s_win[wave]);
    const uint32_t id_bw = M.id_bw, ids_end = M.ids_off + 4 + M.ids_len;
    uint16_t* ids = s_ids[wave];
    bool bad = false;
    for (uint32_t tile = wave; tile < ntiles; tile += kBasesWaves) {
        const uint32_t tile_start = tile * (uint32_t)kRleTile, tile_end = min(n, tile_start + (uint32_t)kRleTile);
        const uint32_t cnt = tile_end - tile_start;
        BasesWin::wave_lds_sync();  // the tile before has been read
        rle_tile_expand(w, id_bw, ids_end, restarts[tile], tile_start, tile_end, ids);
        uint64_t sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = q * 256 + 4 * lane;
            const uint2 four = ((const uint2*)ids)[q * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t id = ((g < 2 ? four.x : four.y) >> (16 * (g & 1))) & 0xFFFFu;
                const bool in_tile = r + g < cnt;
                if (!in_tile) id = 0;
                if (id >= ec) { bad = true; id = 0; }
                uint32_t len;
                if (tab) len = elem == 1 ? (uint32_t)s_tab[id] : (uint32_t)((const uint16_t*)s_tab)[id];
                else { uint2 ab; __builtin_memcpy(&ab, doff + id, 8); len = ab.y - ab.x; }
                sum += in_tile ? len : 0u;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) s_tile[tile] = sum;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (s_bad) { if (tid == 0) M.status = E_CORRUPT; return; }
    // ---- tile sums -> tile bases (wave 0), the exact total
    __shared__ unsigned long long s_tot;
    if (wave == 0) {
        const uint64_t mine = lane < ntiles ? s_tile[lane] : 0ull;
        const uint64_t inc = wave_incl_scan<uint64_t>(mine);
        if (lane < ntiles) ((uint32_t*)(scratch + P.s_rows))[lane] = (uint32_t)(inc - mine);
        if (lane == 63) s_tot = inc;
    }
    __syncthreads();
    const uint64_t tot64 = s_tot;
    if (tot64 > body_cap) {  // the body must fit dpn.dataSize = index + body (DataPackCreator_VLT.java:185-205)
        if (tid == 0) M.status = E_OVERFLOW;
        return;
    }
    uint32_t* __restrict__ off = (uint32_t*)(out + P.out_off);
    if (tid == 0) off[n] = (uint32_t)tot64;
    // A body shorter than dpn.dataSize promises leaves a tail: the reference decodes into a zero-filled direct buffer
    // (PackCompressor_VLT.java:133, ByteBufferUtil.java:31-36), so the tail reads as zeros.
    uint8_t* tail = out + P.out_off + 4ull * (n + 1) + tot64;
    for (uint64_t i = tid; i < body_cap - tot64; i += kBasesWaves * 64) tail[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// k_str_copy: string dictionary packs: body[offset[i] .. offset[i+1]) = dictBody[srcOff[i] ..].  One workgroup per tile of
// kStrTileRows rows; the tile's offsets are staged in LDS.  Each lane produces 16-byte aligned chunks of the output body:
// a branch-free binary search in the staged offsets finds the row of the chunk's first byte, then every row that
// overlaps the chunk contributes one unaligned 16-byte load merged under a byte mask.  Stores are full-width and
// coalesced however short the strings are.  (StringsStruct.Builder.addString, StringsStruct.java:120-131.)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 ld16u(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// 16-byte aligned streaming store: decoded rows are written once and not read again by the decoder, a non-temporal store
// keeps them from displacing the dictionaries in L2
__device__ __forceinline__ void st16_stream(uint8_t* p, uint4 w) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t = {w.x, w.y, w.z, w.w};
    __builtin_nontemporal_store(t, (u32x4*)p);
}
// Bytes [from, from + n) of the 16-byte value w to p, n = 1..15: the value is shifted down by `from` bytes and goes out as the
// 8-, 4-, 2- and 1-byte pieces of n (unaligned stores), four store instructions at most instead of one per byte.
__device__ __forceinline__ void st_part(uint8_t* p, uint4 w, uint32_t from, uint32_t n) {
    uint64_t lo = (uint64_t)w.x | ((uint64_t)w.y << 32), hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
    if (from >= 8) { lo = hi; hi = 0; from -= 8; }
    if (from) { lo = (lo >> (8 * from)) | (hi << (64 - 8 * from)); hi >>= 8 * from; }
    if (n & 8) { __builtin_memcpy(p, &lo, 8); p += 8; lo = hi; }
    if (n & 4) { const uint32_t v = (uint32_t)lo; __builtin_memcpy(p, &v, 4); p += 4; lo >>= 32; }
    if (n & 2) { const uint16_t v = (uint16_t)lo; __builtin_memcpy(p, &v, 2); p += 2; lo >>= 16; }
    if (n & 1) *p = (uint8_t)lo;
}
// 64-bit mask with the low 8*x bits set, x clamped to 0..8
__device__ __forceinline__ uint64_t low_bytes(int x) {
    x = max(x, 0);
    return x >= 8 ? ~0ull : ((1ull << (8 * x)) - 1ull);
}

__global__ __launch_bounds__(256) void k_str_copy(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                  const PackMeta* __restrict__ metas, uint8_t* __restrict__ out,
                                                  const uint8_t* __restrict__ scratch) {
    uint32_t bx, by;
    pack_block_remap(bx, by);
    const int pi = list[by];
    const PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_STR) return;
    const DevPack& P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count;
    const uint32_t r0 = bx * kStrTileRows;
    if (r0 >= n) return;
    const uint32_t rows = min((uint32_t)kStrTileRows, n - r0);
    const uint32_t* off = (const uint32_t*)(out + P.out_off);
    uint8_t* body = out + P.out_off + 4ull * (n + 1);
    // shift all body positions by A so that position 0 is 16-byte aligned in memory
    const uint32_t A = (uint32_t)((uintptr_t)body & 15);
    uint8_t* base16 = body - A;
    const uint32_t ec = M.entry_count;
    const uint8_t* dbody = scratch + P.s_dict + 4ull * (ec + 1);
    const uint32_t* srcoff = (const uint32_t*)(scratch + P.s_rows);

    __shared__ uint32_t s_off[kStrTileRows + 1];  // shifted output offsets of the tile's rows
    __shared__ uint32_t s_dlt[kStrTileRows];      // srcOff - s_off + 16 (mod 2^32): source position = s_dlt + output position
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i <= rows; i += 256) s_off[i] = off[r0 + i] + A;
    for (uint32_t i = tid; i < rows; i += 256) s_dlt[i] = srcoff[r0 + i] - (off[r0 + i] + A) + 16u;
    __syncthreads();
    const uint32_t b0 = s_off[0], b1 = s_off[rows];
    if (b1 <= b0) return;
    const uint8_t* dbody16 = dbody - 16;  // chunk byte k of row r comes from dbody16[s_dlt[r] + cpos + k]; the sum is >= 1

    // One 16-byte chunk: finish it given the row `r` of its first byte; handles chunks that straddle rows.
    auto finish = [&](uint32_t cpos, uint32_t first, uint32_t cend, uint32_t r, uint4 w) {
        uint32_t rend = s_off[r + 1];
        uint32_t pos = min(cend, rend);
        if (pos < cend) {
            uint64_t wl = (uint64_t)w.x | ((uint64_t)w.y << 32), wh = (uint64_t)w.z | ((uint64_t)w.w << 32);
            while (pos < cend) {
                do { r++; rend = s_off[r + 1]; } while (rend <= pos && r + 1 < rows);  // next row that is not empty
                const uint32_t segend = min(cend, max(rend, pos + 1));
                const uint4 v = ld16u(dbody16 + (uint32_t)(s_dlt[r] + cpos));
                const int a = (int)(pos - cpos), e = (int)(segend - cpos);
                const uint64_t ml = low_bytes(e) & ~low_bytes(a), mh = low_bytes(e - 8) & ~low_bytes(a - 8);
                wl = (wl & ~ml) | (((uint64_t)v.x | ((uint64_t)v.y << 32)) & ml);
                wh = (wh & ~mh) | (((uint64_t)v.z | ((uint64_t)v.w << 32)) & mh);
                pos = segend;
            }
            w = make_uint4((uint32_t)wl, (uint32_t)(wl >> 32), (uint32_t)wh, (uint32_t)(wh >> 32));
        }
        if (first == cpos && cend == cpos + 16u) {
            st16_stream(base16 + cpos, w);
        } else {
            const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t q = cpos + k;
                if (q >= first && q < cend) base16[q] = (uint8_t)(wv[k >> 2] >> (8 * (k & 3)));
            }
        }
    };

    // Two chunks per lane and step (4 KiB apart, so that each store instruction of the wave stays contiguous): the two
    // row searches run in one branch-free loop and both first loads are in flight together.
    for (uint32_t cA = (b0 & ~15u) + 16u * tid; cA < b1; cA += 2u * 16u * 256u) {
        const uint32_t cB = cA + 16u * 256u;
        const bool hasB = cB < b1;
        const uint32_t fA = max(cA, b0), eA = min(cA + 16u, b1);
        const uint32_t fB = hasB ? cB : fA, eB = min(cB + 16u, b1);
        // largest r in [0, rows) with s_off[r] <= first: s_off[0] = b0 <= first < b1 = s_off[rows]
        uint32_t rA = 0, rB = 0;
#pragma unroll
        for (uint32_t step = kStrTileRows / 2; step >= 1; step >>= 1) {
            const uint32_t ca = rA + step, cb = rB + step;
            const uint32_t oa = s_off[min(ca, (uint32_t)kStrTileRows)], ob = s_off[min(cb, (uint32_t)kStrTileRows)];
            if (ca < rows && oa <= fA) rA = ca;
            if (cb < rows && ob <= fB) rB = cb;
        }
        const uint4 wA = ld16u(dbody16 + (uint32_t)(s_dlt[rA] + cA));
        const uint4 wB = ld16u(dbody16 + (uint32_t)(s_dlt[rB] + (hasB ? cB : cA)));
        finish(cA, fA, eA, rA, wA);
        if (hasB) finish(cB, fB, eB, rB, wB);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_str_fused: row index and body copy of string dictionary packs in one kernel, one WAVE per tile of kRleTile rows; no
// workgroup barrier anywhere.  The wave expands its tile's ids, looks the rows' dictionary slices up, scans the row lengths
// (the tile's first offset comes from k_str_bases, which leaves one word per tile instead of eight bytes per
// row; having every tile publish its byte count and look back at the tiles before it instead was measured: the round trips
// of the device-scope atomics under a saturated store stream cost three times what the small kernel does),
// writes the tile's part of the output index and keeps the rows' offsets in LDS; then it produces the tile's stretch
// of the body in 16-byte aligned chunks exactly as k_str_copy does (row of the chunk's first byte by binary search in
// LDS, one unaligned 16-byte load per row that overlaps the chunk, merged under byte masks, full-width streaming stores).
// Nothing about the rows is read back from memory: what a copy kernel gets from HBM at the head of its dependency chain
// is what it waits for, and per-row offsets are 8 bytes for every ~55 the body holds.
// (DictStruct.decodeStrings, DictStruct.java:97-110; StringsStruct.Builder, StringsStruct.java:120-158.)
// ---------------------------------------------------------------------------------------------------------------
#ifndef IXR_FUSED_PASS_ROWS
#define IXR_FUSED_PASS_ROWS 512  // rows whose offsets are in LDS at a time (256, 512 or 1024)
#endif
// The kernel waits for gathers that miss L2, so what it needs is waves: the LDS of a wave is kept small (a 2 KB stream window
// instead of the usual 4 KB, the row offsets of half a tile at a time) so that eight workgroups fit a compute unit.
using FusedWin = WinT<2048>;
constexpr uint32_t kFusedPassRows = IXR_FUSED_PASS_ROWS;
constexpr uint32_t kFusedDltOff = (4 * (kFusedPassRows + 1) + 15) & ~15u;  // the deltas start 16-byte aligned behind the offsets
#ifndef IXR_FUSED_WAVE_LDS
#define IXR_FUSED_WAVE_LDS 0    // build switch: pad a wave's LDS to this many bytes (fewer workgroups, so fewer packs in flight, per compute unit)
#endif
constexpr uint32_t kFusedWaveNeed =                     // per wave: stream window + id tile while expanding, then offsets[] + deltas[]
    ((FusedWin::kAlloc + 2 * kRleTile > kFusedDltOff + 4 * kFusedPassRows ? FusedWin::kAlloc + 2 * kRleTile : kFusedDltOff + 4 * kFusedPassRows) + 15) & ~15u;
constexpr uint32_t kFusedWaveLds = kFusedWaveNeed > IXR_FUSED_WAVE_LDS ? kFusedWaveNeed : IXR_FUSED_WAVE_LDS;
static_assert(kRleTile % kFusedPassRows == 0 && kFusedPassRows % 256 == 0, "a pass is a whole number of 256-row groups");

#ifndef IXR_FUSED_WAVES_PER_SIMD
#define IXR_FUSED_WAVES_PER_SIMD 6
#endif
__global__ __launch_bounds__(256, IXR_FUSED_WAVES_PER_SIMD) void k_str_fused(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                   const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                   uint8_t* __restrict__ out, const uint8_t* __restrict__ scratch) {
    __shared__ __align__(16) uint8_t smem[4][kFusedWaveLds];
    uint32_t bx, by;
    pack_block_remap(bx, by);
    const int pi = list[by];
    const PackMeta& M = metas[pi];
    if (M.status < 0 || M.kind != K_DICT_STR) return;
    const DevPack& P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count, ec = M.entry_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = bx * 4 + wave;
    const uint32_t tile_start = tile * (uint32_t)kRleTile;
    if (tile_start >= n) return;
    const uint32_t tile_end = min(n, tile_start + (uint32_t)kRleTile), cnt = tile_end - tile_start;
    const uint32_t* __restrict__ doff = (const uint32_t*)(scratch + P.s_dict);
    const uint8_t* dbody = scratch + P.s_dict + 4ull * (ec + 1);
    uint32_t* __restrict__ off = (uint32_t*)(out + P.out_off);
    uint8_t* body = out + P.out_off + 4ull * (n + 1);
    // shift all body positions by A so that position 0 is 16-byte aligned in memory
    const uint32_t A = (uint32_t)((uintptr_t)body & 15);
    uint8_t* base16 = body - A;
    const uint32_t tbase = uniform_load((const uint32_t*)(scratch + P.s_rows) + tile);  // offset of the tile's first row
    uint8_t* wl = smem[wave];
    // this workgroup's share of the dictionary (offsets and bytes; its size was checked against the scratch region by the
    // dictionary scan) on its way into L2 while the ids are expanded
    const Warm warm = warm_issue(scratch + P.s_dict, 4u * (ec + 1u) + uniform_load(doff + ec), bx, gridDim.x, ec, n);

    // ---- the tile's ids and the rows' dictionary slices
    uint2 ab[16];
    {
        uint16_t* ids = (uint16_t*)(wl + FusedWin::kAlloc);
        FusedWin w;
        w.init(in + P.in_off, (uint32_t)P.pack_size, wl);
        rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, uniform_load((const uint4*)(scratch + P.s_restart) + tile), tile_start,
                        tile_end, ids);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = q * 256 + 4 * lane;
            const uint2 four = ((const uint2*)ids)[q * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t id = ((g < 2 ? four.x : four.y) >> (16 * (g & 1))) & 0xFFFFu;
                if (r + g >= cnt || id >= ec) id = 0;  // ids were checked by the index kernel
                __builtin_memcpy(&ab[q * 4 + g], doff + id, 8);  // doff[id], doff[id + 1]
            }
        }
        FusedWin::wave_lds_sync();  // the id tile and the window are dead: their memory becomes the offset tables
    }
    warm_done(warm);
    uint32_t* s_off = (uint32_t*)wl;                   // [rows + 1] shifted output offsets of the pass's rows
    uint32_t* s_dlt = (uint32_t*)(wl + kFusedDltOff);  // [rows] srcOff - s_off + 16 (mod 2^32)
    const uint8_t* dbody16 = dbody - 16;  // chunk byte k of row r comes from dbody16[s_dlt[r] + cpos + k]; the sum is >= 1
    uint32_t run = tbase + A;
    constexpr int kGroups = kFusedPassRows / 256;  // 256-row groups per pass
#pragma unroll
    for (int pass = 0; pass < (int)(kRleTile / kFusedPassRows); pass++) {
        const uint32_t pass_first = pass * kFusedPassRows;
        if (pass_first >= cnt) break;
        const uint32_t rows = min(cnt - pass_first, kFusedPassRows);
        // ---- scan of the row lengths: the output index leaves as 16-byte stores, the shifted offsets and the source deltas
        // of the pass's rows go to LDS
#pragma unroll
        for (int qq = 0; qq < kGroups; qq++) {
            const int q = pass * kGroups + qq;
            const uint32_t r = q * 256 + 4 * lane;   // row in the tile
            const uint32_t rl = r - pass_first;      // row in the pass
            uint32_t len[4], ex[4], s4 = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                len[g] = r + g < cnt ? ab[q * 4 + g].y - ab[q * 4 + g].x : 0u;
                ex[g] = s4;
                s4 += len[g];
            }
            const uint32_t inc = wave_incl_scan(s4);
            const uint32_t first = run + inc - s4;
#pragma unroll
            for (int g = 0; g < 4; g++) ex[g] += first;
            run += __builtin_amdgcn_readlane(inc, 63);
            if (r + 3 < cnt) {
                *(uint4*)(off + tile_start + r) = make_uint4(ex[0] - A, ex[1] - A, ex[2] - A, ex[3] - A);
            } else {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    if (r + g < cnt) off[tile_start + r + g] = ex[g] - A;
            }
            if (r < cnt) {  // rows past the tile's end have length 0: their entries equal the tile's end offset
                *(uint4*)(s_off + rl) = make_uint4(ex[0], ex[1], ex[2], ex[3]);
                *(uint4*)(s_dlt + rl) = make_uint4(ab[q * 4].x - ex[0] + 16u, ab[q * 4 + 1].x - ex[1] + 16u, ab[q * 4 + 2].x - ex[2] + 16u,
                                                   ab[q * 4 + 3].x - ex[3] + 16u);
            }
        }
        if (lane == 0) s_off[rows] = run;
        FusedWin::wave_lds_sync();
        const uint32_t b0 = s_off[0], b1 = s_off[rows];

        // One 16-byte chunk: finish it given the row `r` of its first byte; handles chunks that straddle rows.
        auto finish = [&](uint32_t cpos, uint32_t first, uint32_t cend, uint32_t r, uint4 w) {
            uint32_t rend = s_off[r + 1];
            uint32_t pos = min(cend, rend);
            if (pos < cend) {
                uint64_t wlo = (uint64_t)w.x | ((uint64_t)w.y << 32), whi = (uint64_t)w.z | ((uint64_t)w.w << 32);
                while (pos < cend) {
                    do { r++; rend = s_off[r + 1]; } while (rend <= pos && r + 1 < rows);  // next row that is not empty
                    const uint32_t segend = min(cend, max(rend, pos + 1));
                    const uint4 v = ld16u(dbody16 + (uint32_t)(s_dlt[r] + cpos));
                    const int a = (int)(pos - cpos), e = (int)(segend - cpos);
                    const uint64_t ml = low_bytes(e) & ~low_bytes(a), mh = low_bytes(e - 8) & ~low_bytes(a - 8);
                    wlo = (wlo & ~ml) | (((uint64_t)v.x | ((uint64_t)v.y << 32)) & ml);
                    whi = (whi & ~mh) | (((uint64_t)v.z | ((uint64_t)v.w << 32)) & mh);
                    pos = segend;
                }
                w = make_uint4((uint32_t)wlo, (uint32_t)(wlo >> 32), (uint32_t)whi, (uint32_t)(whi >> 32));
            }
            if (first == cpos && cend == cpos + 16u) {
                st16_stream(base16 + cpos, w);
            } else {  // the first and the last chunk of a pass are shared with the neighbouring pass or tile
                st_part(base16 + first, w, first - cpos, cend - first);
            }
        };

        // U chunks per lane and step (1 KiB apart, so that each store instruction of the wave stays contiguous): the
        // row searches run in one branch-free loop and all first loads are in flight together.
        constexpr int U = IXR_FUSED_CHUNKS;
        for (uint32_t c0 = (b0 & ~15u) + 16u * lane; c0 < b1; c0 += U * 16u * 64u) {
            uint32_t cp[U], fp[U], ep[U], rr[U];
            bool has[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                cp[u] = c0 + u * 16u * 64u;
                has[u] = cp[u] < b1;
                if (!has[u]) cp[u] = c0;  // lanes past the end repeat their first chunk's search and load, and store nothing
                fp[u] = max(cp[u], b0);
                ep[u] = min(cp[u] + 16u, b1);
                rr[u] = 0;
            }
            // largest r in [0, rows) with s_off[r] <= first: s_off[0] = b0 <= first < b1 = s_off[rows]
#pragma unroll
            for (uint32_t step = kFusedPassRows / 2; step >= 1; step >>= 1) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t c = rr[u] + step;
                    const uint32_t o = s_off[min(c, rows)];
                    if (c < rows && o <= fp[u]) rr[u] = c;
                }
            }
            uint4 w[U];
#pragma unroll
            for (int u = 0; u < U; u++) w[u] = ld16u(dbody16 + (uint32_t)(s_dlt[rr[u]] + cp[u]));
#pragma unroll
            for (int u = 0; u < U; u++)
                if (has[u]) finish(cp[u], fp[u], ep[u], rr[u], w[u]);
        }
        FusedWin::wave_lds_sync();  // the next pass overwrites the tables
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_dict_struct: materialises the uncompressed DictStruct of DICT_COMPRESS packs (PackCompressor_VLT.getDictStruct,
// PackCompressor_VLT.java:172-186; layout DictStruct.java:14-55): entryCount | entries | 4n | ids.
// grid.x = 1 + tiles/4: block 0 copies header + entries from the scratch dictionary, the others expand the ids.
// The id region can be unaligned (string dictionaries have arbitrary byte size), so ids are stored bytewise then.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dict_struct(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                     PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                     uint8_t* __restrict__ out, const uint8_t* __restrict__ scratch) {
    __shared__ uint16_t s_ids[4][kRleTile];
    __shared__ __align__(16) uint8_t s_win[4][kWinAlloc];
    const int pi = list[blockIdx.y];
    const PackMeta& M = metas[pi];
    if (M.status < 0 || (M.kind != K_DICT_NUM && M.kind != K_DICT_STR) || M.dict_struct_size == 0) return;
    const DevPack P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count;
    const uint32_t ec = M.entry_count;
    const uint8_t* dict = scratch + P.s_dict;
    uint8_t* ds = out + P.ds_off;
    uint32_t dict_size;
    if (M.kind == K_DICT_NUM) dict_size = ec * (uint32_t)(P.width / 8);
    else dict_size = 4 * (ec + 1) + ((const uint32_t*)dict)[ec];
    if (4ull + dict_size + 4ull + 4ull * n != (uint64_t)M.dict_struct_size) {
        if (blockIdx.x == 0 && threadIdx.x == 0) metas[pi].status = E_CORRUPT;
        return;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            *(uint32_t*)ds = ec;
            const uint32_t v = 4 * n;
            uint8_t* p = ds + 4 + dict_size;
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        }
        // ds + 4 and dict are both 4-byte aligned
        const uint32_t words = dict_size >> 2;
        for (uint32_t i = threadIdx.x; i < words; i += 256) ((uint32_t*)(ds + 4))[i] = ((const uint32_t*)dict)[i];
        for (uint32_t i = (words << 2) + threadIdx.x; i < dict_size; i += 256) ds[4 + i] = dict[i];
        return;
    }
    const uint32_t tile = (blockIdx.x - 1) * 4 + (threadIdx.x >> 6);
    const uint32_t tile_start = tile * kRleTile;
    if (tile_start >= n) return;
    const uint32_t tile_end = min(n, tile_start + (uint32_t)kRleTile);
    uint8_t* idp = ds + 4 + dict_size + 4 + 4ull * tile_start;
    const bool aligned = (((uintptr_t)idp) & 3) == 0;
    const uint4 seg = ((const uint4*)(scratch + P.s_restart))[tile];
    uint16_t* ids = s_ids[threadIdx.x >> 6];
    Win w;
    w.init(in + P.in_off, (uint32_t)P.pack_size, s_win[threadIdx.x >> 6]);
    rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, seg, tile_start, tile_end, ids);
    const uint32_t cnt = tile_end - tile_start;
    for (uint32_t r = threadIdx.x & 63; r < cnt; r += 64) {
        const uint32_t id = ids[r];
        if (aligned) ((uint32_t*)idp)[r] = id;
        else {
            uint8_t* p = idp + 4ull * r;
            p[0] = (uint8_t)id; p[1] = (uint8_t)(id >> 8); p[2] = 0; p[3] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Rows out of an already uncompressed DictStruct (PackCompressor_VLT.getDataFromDictStruct, PackCompressor_VLT.java:188-213;
// DictStruct.decodeXxx, DictStruct.java:57-110).  The "pack" handed to these kernels is the DictStruct itself.
//   k_ds_meta   fills the meta record so that the regular dictionary kernels can be reused (plain entries at offset 4)
//   k_ds_rows   reads the ids straight from the DictStruct (they may be unaligned) and gathers numbers, or produces the
//               row lengths / source offsets of strings for k_row_scan + k_str_copy
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_ds_meta(const DevPack* __restrict__ packs, const int* __restrict__ list, int count,
                          PackMeta* __restrict__ metas, const uint8_t* __restrict__ in) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= count) return;
    const int pi = list[li];
    const DevPack P = packs[pi];
    const uint8_t* base = in + P.in_off;
    const uint32_t psz = (uint32_t)P.pack_size, n = (uint32_t)P.obj_count;
    PackMeta m;
    m.status = P.pack_size; m.kind = P.width ? K_DICT_NUM : K_DICT_STR; m.entry_count = 0; m.id_bw = 0; m.ids_off = 0;
    m.ids_len = 0; m.entries_off = 4; m.entries_plain = 1; m.sfx_off = 0; m.sfx_len = 0; m.dict_struct_size = psz; m.max_len = 0;
    m.front_blocks = 0; m.front_stride = 0; m.sliced = 0;
    m.s[0].n = m.s[1].n = 0; m.s[0].in_off = m.s[1].in_off = 0; m.s[0].first = m.s[1].first = 0;
    int32_t err = 0;
    if (psz < 8 || n == 0 || n > 65536u) err = E_CORRUPT;
    if (!err) {
        const uint32_t ec = ld32u(base);
        if (ec == 0 || ec > 65536u) err = E_CORRUPT;
        else {
            m.entry_count = ec;
            uint64_t dict_size;
            if (P.width) dict_size = (uint64_t)ec * (P.width / 8);
            else if (4ull + 4ull * (ec + 1) > psz) { err = E_CORRUPT; dict_size = 0; }
            else dict_size = 4ull * (ec + 1) + ld32u(base + 4 + 4ull * ec);
            if (!err && 4 + dict_size + 4 + 4ull * n > psz) err = E_CORRUPT;
            if (!err && dict_size > P.dict_cap) err = E_OVERFLOW;
            if (!err && ld32u(base + 4 + dict_size) != 4 * n) err = E_CORRUPT;
            m.ids_off = (uint32_t)(4 + dict_size + 4);
        }
    }
    if (err) m.status = err;
    metas[pi] = m;
}

template <int MODE>  // 0: 32-bit numbers, 1: 64-bit numbers, 2: strings
__global__ __launch_bounds__(256) void k_ds_rows(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                 PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                 uint8_t* __restrict__ out, uint8_t* __restrict__ scratch) {
    const int pi = list[blockIdx.y];
    const PackMeta& M = metas[pi];
    if (M.status < 0) return;
    const DevPack P = packs[pi];
    const uint32_t n = (uint32_t)P.obj_count, ec = M.entry_count;
    const uint8_t* ids = in + P.in_off + M.ids_off;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t id = ld32u(ids + 4ull * i);
    bool bad = false;
    if (id >= ec) { bad = true; id = 0; }
    if constexpr (MODE == 0) {
        ((uint32_t*)(out + P.out_off))[i] = ((const uint32_t*)(scratch + P.s_dict))[id];
    } else if constexpr (MODE == 1) {
        ((uint64_t*)(out + P.out_off))[i] = ((const uint64_t*)(scratch + P.s_dict))[id];
    } else {
        if (4ull * (n + 1) > (uint64_t)P.data_size) return;  // no room for the index: k_row_scan reports the overflow
        const uint32_t* doff = (const uint32_t*)(scratch + P.s_dict);
        const uint32_t a = doff[id], b = doff[id + 1];
        ((uint32_t*)(out + P.out_off))[i + 1] = b - a;
        ((uint32_t*)(scratch + P.s_rows))[i] = a;
    }
    if (bad) metas[pi].status = E_CORRUPT;
}

// ---------------------------------------------------------------------------------------------------------------
// Predicate evaluation on dictionary ids (ExtIndex_DictBits.java:33-200).
//   k_filter_search  one wave per pack: looks the keys up in the decoded dictionary (BinarySearch.java: plain < and > on the
//                    entry type, unsigned bytes then length for strings), one key per lane, and reduces them to what the
//                    row pass needs: NONE / ALL, an id range [from, to), or a list of matched ids.  `like` (:182-200)
//                    is left to k_filter_like: the pattern against every dictionary entry, one workgroup per pack.
//   k_filter_rows    one wave per 1024-row tile: expands the ids, tests them (range compare, or a bit mask over the
//                    dictionary entries in LDS for id lists), ballots 64 rows into one bitmap word.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ int search_num(const T* e, int ec, T key) {
    int from = 0, to = ec - 1;
    while (from <= to) {
        const int mid = (int)(((unsigned)from + (unsigned)to) >> 1);
        const T v = e[mid];
        if (v < key) from = mid + 1;
        else if (v > key) to = mid - 1;
        else return mid;
    }
    return -(from + 1);
}
__device__ __forceinline__ int search_str(const uint32_t* doff, const uint8_t* body, int ec, const uint8_t* key, int klen) {
    int from = 0, to = ec - 1;
    while (from <= to) {
        const int mid = (int)(((unsigned)from + (unsigned)to) >> 1);
        const uint32_t a = doff[mid], b = doff[mid + 1];
        const int elen = (int)(b - a), len = min(elen, klen);
        int c = 0;
        for (int i = 0; i < len && c == 0; i++) c = (int)body[a + i] - (int)key[i];  // BytesUtil.compareBytes :7-19
        if (c == 0) c = elen - klen;
        if (c < 0) from = mid + 1;
        else if (c > 0) to = mid - 1;
        else return mid;
    }
    return -(from + 1);
}

// SQLLike.match (indexr-common/src/main/java/io/indexr/util/SQLLike.java:12-94): % any run of bytes, _ one byte, \\ takes the
// next pattern byte literally.  The reference recurses over every split point of a %; whether some split matches is what
// the loop below finds by going back to the last % and letting it take one more byte, without a stack.
__device__ __forceinline__ bool like_match(const uint8_t* s, int slen, const uint8_t* p, int plen) {
    int si = 0, pi = 0, star_p = -1, star_s = 0;
    for (;;) {
        if (si >= slen) {  // :28-36: only % may be left
            while (pi < plen && p[pi] == '%') pi++;
            return pi >= plen;
        }
        if (pi < plen) {
            const uint8_t c = p[pi];
            if (c == '\\') {  // :49-53 then :83-91; a trailing escape never matches (:39-41)
                if (pi + 1 < plen && p[pi + 1] == s[si]) { si++; pi += 2; continue; }
            } else if (c == '_') {  // :54-58
                si++; pi++; continue;
            } else if (c == '%') {  // :59-79
                star_p = pi; star_s = si; pi++; continue;
            } else if (c == s[si]) {
                si++; pi++; continue;
            }
        }
        if (star_p < 0) return false;
        si = ++star_s;
        pi = star_p + 1;
    }
}

__global__ __launch_bounds__(64) void k_filter_search(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                      const PackMeta* __restrict__ metas, const uint8_t* __restrict__ scratch,
                                                      DevFilter* __restrict__ filters, const uint8_t* __restrict__ values,
                                                      uint32_t* __restrict__ matches) {
    const int pi = list[blockIdx.x], lane = threadIdx.x;
    const DevPack& P = packs[pi];
    const PackMeta& M = metas[pi];
    DevFilter& F = filters[pi];
    if (M.status < 0) { if (lane == 0) F.result = M.status; return; }
    if (P.encode_type != 6 || (M.kind != K_DICT_NUM && M.kind != K_DICT_STR)) { if (lane == 0) F.result = -3; return; }
    const int ec = (int)M.entry_count, nv = F.n_values, op = F.op;
    const uint8_t* dict = scratch + P.s_dict;
    const uint8_t* vals = values + F.values_off;
    auto lookup = [&](int v) -> int {  // searchEntry :33-58
        if (P.width == 0) {
            const int32_t* koff = (const int32_t*)vals;
            const uint8_t* kbody = vals + 4ull * ((uint32_t)nv + 1);
            return search_str((const uint32_t*)dict, dict + 4ull * ((uint32_t)ec + 1), ec, kbody + koff[v], koff[v + 1] - koff[v]);
        }
        const int64_t num = ((const int64_t*)vals)[v];
        switch (P.data_type) {
            case 1: return search_num<int32_t>((const int32_t*)dict, ec, (int32_t)(uint32_t)(uint64_t)num);
            case 2: return search_num<int64_t>((const int64_t*)dict, ec, num);
            case 3: return search_num<float>((const float*)dict, ec, (float)__longlong_as_double(num));
            default: return search_num<double>((const double*)dict, ec, __longlong_as_double(num));
        }
    };
    if (op == 1 || op == 2) {  // equal :60-79, in :81-113
        uint32_t total = 0;
        for (int v0 = 0; v0 < nv; v0 += 64) {
            const int v = v0 + lane;
            const int id = v < nv ? lookup(v) : -1;
            const uint64_t hit = __ballot(id >= 0);
            if (id >= 0) matches[F.match_off + total + __popcll(hit & ((1ull << lane) - 1ull))] = (uint32_t)id;
            total += (uint32_t)__popcll(hit);
        }
        if (lane == 0) {
            F.n_match = total;
            F.result = total == 0 ? 0 : ((op == 1 && ec == 1) ? 1 : 2);
        }
    } else if (op == 3 || op == 4) {  // greater :115-145
        if (lane == 0) {
            const int id = lookup(0);
            const int start = id >= 0 ? (op == 4 ? id : id + 1) : -id - 1;
            F.from = (uint32_t)start; F.to = (uint32_t)ec; F.n_match = 0;
            F.result = start >= ec ? 0 : (start == 0 ? 1 : 2);
        }
    } else if (op == 5 && nv >= 2) {  // between :147-180
        const int id = lane < 2 ? lookup(lane) : 0;
        const int id1 = __shfl(id, 0), id2 = __shfl(id, 1);
        if (lane == 0) {
            const int from = id1 >= 0 ? id1 : -id1 - 1, to = id2 >= 0 ? id2 + 1 : -id2 - 1;
            F.from = (uint32_t)max(from, 0); F.to = (uint32_t)max(to, 0); F.n_match = 0;
            if (!(from >= 0 && from < ec) || !(to >= 0 && to <= ec)) F.result = -100;  // Preconditions.checkState :165-166
            else F.result = to - from <= 0 ? 0 : (to - from >= ec ? 1 : 2);
        }
    } else if (op == 6) {  // like: k_filter_like
    } else if (lane == 0) {
        F.result = -1000;
    }
}

// like :182-200: a row matches when its dictionary entry does.  One workgroup per pack, one entry per thread and round;
// a round's entries lie next to each other in the dictionary body and are staged in LDS when they fit (the matcher reads
// byte after byte, each read waiting for the one before).  The ids of the matching entries are appended to the pack's
// match list in no particular order.
constexpr int kLikeThreads = 512, kLikeLds = 40960, kLikePat = 256;
__global__ __launch_bounds__(kLikeThreads) void k_filter_like(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                              const PackMeta* __restrict__ metas, const uint8_t* __restrict__ scratch,
                                                              DevFilter* __restrict__ filters, const uint8_t* __restrict__ values,
                                                              uint32_t* __restrict__ matches) {
    __shared__ uint32_t s_total;
    __shared__ __align__(16) uint8_t s_body[kLikeLds + 16];
    __shared__ uint8_t s_pat[kLikePat];
    const int pi = list[blockIdx.x], lane = threadIdx.x & 63;
    const DevPack& P = packs[pi];
    const PackMeta& M = metas[pi];
    DevFilter& F = filters[pi];
    if (F.op != 6 || M.status < 0 || P.encode_type != 6 || (M.kind != K_DICT_NUM && M.kind != K_DICT_STR)) return;  // k_filter_search answered
    if (P.width != 0) { if (threadIdx.x == 0) F.result = E_UNSUPPORTED_TYPE; return; }  // "column type ... is illegal in LIKE"
    const int ec = (int)M.entry_count;
    const uint8_t* dict = scratch + P.s_dict;
    const uint8_t* vals = values + F.values_off;
    const int32_t* koff = (const int32_t*)vals;
    const uint8_t* pat = vals + 8 + koff[0];
    const int plen = koff[1] - koff[0];
    const uint32_t* doff = (const uint32_t*)dict;
    const uint8_t* body = dict + 4ull * ((uint32_t)ec + 1);  // 4-byte aligned: the dictionary slot is 16-byte aligned
    const bool pat_lds = plen <= kLikePat;
    if (threadIdx.x == 0) s_total = 0;
    if (pat_lds) for (int i = threadIdx.x; i < plen; i += kLikeThreads) s_pat[i] = pat[i];
    for (int e0 = 0; e0 < ec; e0 += kLikeThreads) {  // workgroup-uniform trip count
        const uint32_t lo = doff[e0], hi = doff[min(e0 + kLikeThreads, ec)];
        const uint32_t lo4 = lo & ~3u;  // body + lo4 is 4-byte aligned
        const bool staged = hi - lo4 <= (uint32_t)kLikeLds;
        __syncthreads();  // the round before is done with s_body; s_total / s_pat are written
        if (staged) {
            for (uint32_t i = threadIdx.x * 4; i < hi - lo4; i += kLikeThreads * 4)
                *(uint32_t*)(s_body + i) = *(const uint32_t*)(body + lo4 + i);  // may read up to 3 bytes past hi: inside the slot's slack
            __syncthreads();
        }
        const int e = e0 + (int)threadIdx.x;
        bool hit = false;
        if (e < ec) {
            const uint32_t a = doff[e], len = doff[e + 1] - a;
            if (staged && pat_lds) hit = like_match(s_body + (a - lo4), (int)len, s_pat, plen);
            else if (staged) hit = like_match(s_body + (a - lo4), (int)len, pat, plen);
            else hit = like_match(body + a, (int)len, pat, plen);
        }
        const uint64_t m = __ballot(hit);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_total, (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (hit) matches[F.match_off + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)e;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        F.n_match = s_total;
        F.result = s_total == 0 ? 0 : 2;  // fixBitmapInPack (PackExtIndex.java:32-39): empty -> NONE, never ALL
    }
}

__global__ __launch_bounds__(256) void k_filter_rows(const DevPack* __restrict__ packs, const int* __restrict__ list,
                                                     const PackMeta* __restrict__ metas, const uint8_t* __restrict__ in,
                                                     DevFilter* __restrict__ filters,
                                                     const uint32_t* __restrict__ matches, unsigned long long* __restrict__ bitmaps,
                                                     const uint8_t* __restrict__ scratch) {
    __shared__ uint16_t s_ids[4][kRleTile];
    __shared__ __align__(16) uint8_t s_win[4][kWinAlloc];
    __shared__ uint32_t s_emask[65536 / 32];  // one bit per dictionary entry: is it one of the matched ids?
    const int pi = list[blockIdx.y], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const DevPack& P = packs[pi];
    const PackMeta& M = metas[pi];
    const DevFilter F = filters[pi];
    const uint32_t n = (uint32_t)P.obj_count;
    const uint32_t tile = blockIdx.x * 4 + wave, tile_start = tile * kRleTile;
    unsigned long long* words = bitmaps + (size_t)pi * 1024 + tile * 16;
    const bool by_set = F.result == 2 && (F.op == 1 || F.op == 2 || F.op == 6);
    if (by_set) {  // workgroup-uniform
        const uint32_t ew = (M.entry_count + 31) / 32;
        for (uint32_t i = threadIdx.x; i < ew; i += 256) s_emask[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < F.n_match; i += 256) {
            const uint32_t id = matches[F.match_off + i];
            atomicOr(&s_emask[id >> 5], 1u << (id & 31));
        }
        __syncthreads();
    }
    if (F.result != 2 || tile_start >= n) {
        // NONE, errors and rows past the pack: zeros; ALL: ones for the rows that exist
        if (lane < 16) {
            unsigned long long w = 0;
            if (F.result == 1 && tile_start < n) {
                const uint32_t r = tile_start + lane * 64;
                w = r + 64 <= n ? ~0ull : (r < n ? ((1ull << (n - r)) - 1ull) : 0ull);
            }
            words[lane] = w;
        }
        return;
    }
    const uint32_t tile_end = min(n, tile_start + (uint32_t)kRleTile), cnt = tile_end - tile_start;
    uint16_t* ids = s_ids[wave];
    const uint4 seg = ((const uint4*)(scratch + P.s_restart))[tile];
    Win w;
    w.init(in + P.in_off, (uint32_t)P.pack_size, s_win[wave]);
    rle_tile_expand(w, M.id_bw, M.ids_off + 4 + M.ids_len, seg, tile_start, tile_end, ids);
    bool bad = false;
    unsigned long long mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t r = q * 64 + lane;
        const uint32_t id = r < cnt ? ids[r] : 0u;
        if (id >= M.entry_count) bad = true;
        bool hit;
        if (by_set) hit = id < M.entry_count && ((s_emask[id >> 5] >> (id & 31)) & 1u);
        else hit = id >= F.from && id < F.to;
        const unsigned long long word = __ballot(hit && r < cnt);
        if (lane == q) mine = word;
    }
    if (lane < 16) words[lane] = mine;
    if (bad) filters[pi].result = E_CORRUPT;
}

// ---------------------------------------------------------------------------------------------------------------
// k_scan_filter: predicates on the rows of a decoded pack (ExtIndex_SimpleBits.java:25-331, what packs without a
// dictionary get).  One wave per 1024-row tile, one row per lane and round, 64 rows ballot into one bitmap word; all 1024
// words of a pack are written.  results[d] must be 0 on entry: it becomes 2 when any row matches (fixBitmapInPack,
// PackExtIndex.java:32-39: an empty bitmap is NONE, the rest comes back as it is) or a negative error.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int compare_bytes(const uint8_t* a, int alen, const uint8_t* b, int blen) {  // BytesUtil.compareBytes :7-19
    const int len = min(alen, blen);
    for (int i = 0; i < len; i++) {
        const int c = (int)a[i] - (int)b[i];
        if (c) return c;
    }
    return alen - blen;
}
template <typename T> __device__ __forceinline__ T scan_key(int64_t num) {
    if constexpr (sizeof(T) == 4 && !std::is_floating_point<T>::value) return (T)(uint32_t)(uint64_t)num;  // (int) numValue
    else if constexpr (!std::is_floating_point<T>::value) return (T)num;
    else return (T)__longlong_as_double(num);  // (float) Double.longBitsToDouble(numValue) / Double.longBitsToDouble(numValue)
}
template <typename T> __device__ __forceinline__ bool scan_num(T v, int op, int nv, const int64_t* keys, T k0, T k1) {
    switch (op) {
        case 1: return v == k0;
        case 2: { bool hit = false; for (int i = 0; i < nv && !hit; i++) hit = v == scan_key<T>(keys[i]); return hit; }
        case 3: return v > k0;
        case 4: return v > k0 || v == k0;
        default: return v >= k0 && v <= k1;
    }
}
__global__ __launch_bounds__(256) void k_scan_filter(const ScanFilterDesc* __restrict__ descs, const uint8_t* __restrict__ decoded,
                                                     const uint8_t* __restrict__ values, unsigned long long* __restrict__ bitmaps,
                                                     int32_t* __restrict__ results) {
    const ScanFilterDesc D = descs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = (uint32_t)D.obj_count, tile = blockIdx.x * 4 + wave, tile_start = tile * 1024u;
    unsigned long long* words = bitmaps + (size_t)blockIdx.y * 1024 + tile * 16;
    const uint8_t* src = decoded + D.src_off;
    const uint8_t* vals = values + D.values_off;
    const int op = D.op, nv = D.n_values, dt = D.data_type;
    int err = 0;
    if (op == 6 && dt != 5) err = E_UNSUPPORTED_TYPE;  // "column type ... is illegal in LIKE"
    else if (dt == 5 ? 4ull * (n + 1) > (uint64_t)D.data_size : (uint64_t)n * ((dt == 1 || dt == 3) ? 4u : 8u) > (uint64_t)D.data_size) err = -1000;
    unsigned long long mine = 0;
    bool bad = false;
    if (!err && tile_start < n) {
        const int64_t* nk = (const int64_t*)vals;
        const int32_t* koff = (const int32_t*)vals;
        const uint8_t* kbody = vals + 4ull * ((uint32_t)nv + 1);
        const int32_t* off = (const int32_t*)src;
        const uint8_t* body = src + 4ull * (n + 1);
        const int64_t body_cap = (int64_t)D.data_size - 4ll * (n + 1);
#pragma unroll 1
        for (int q = 0; q < 16; q++) {
            const uint32_t r = tile_start + q * 64 + lane;
            bool hit = false;
            if (r < n) {
                switch (dt) {
                    case 1: hit = scan_num<int32_t>(((const int32_t*)src)[r], op, nv, nk, scan_key<int32_t>(nk[0]), scan_key<int32_t>(nk[nv > 1])); break;
                    case 2: hit = scan_num<int64_t>(((const int64_t*)src)[r], op, nv, nk, scan_key<int64_t>(nk[0]), scan_key<int64_t>(nk[nv > 1])); break;
                    case 3: hit = scan_num<float>(((const float*)src)[r], op, nv, nk, scan_key<float>(nk[0]), scan_key<float>(nk[nv > 1])); break;
                    case 4: hit = scan_num<double>(((const double*)src)[r], op, nv, nk, scan_key<double>(nk[0]), scan_key<double>(nk[nv > 1])); break;
                    default: {
                        const int32_t a = off[r], b = off[r + 1];
                        if (a < 0 || b < a || (int64_t)b > body_cap) { bad = true; break; }
                        const uint8_t* sp = body + a;
                        const int sl = b - a;
                        if (op == 1 || op == 2) {  // equal :72-84, in :146-161: same length and same bytes
                            for (int i = 0; i < nv && !hit; i++) {
                                const int kl = koff[i + 1] - koff[i];
                                hit = kl == sl && compare_bytes(sp, sl, kbody + koff[i], kl) == 0;
                            }
                        } else if (op == 3 || op == 4) {  // greater :215-229
                            const int c = compare_bytes(sp, sl, kbody + koff[0], koff[1] - koff[0]);
                            hit = c > 0 || (op == 4 && c == 0);
                        } else if (op == 5) {  // between :287-306
                            hit = compare_bytes(sp, sl, kbody + koff[0], koff[1] - koff[0]) >= 0 &&
                                  compare_bytes(sp, sl, kbody + koff[1], koff[2] - koff[1]) <= 0;
                        } else {  // like :314-331
                            hit = like_match(sp, sl, kbody + koff[0], koff[1] - koff[0]);
                        }
                    }
                }
            }
            const unsigned long long word = __ballot(hit);
            if (lane == q) mine = word;
        }
    }
    if (lane < 16) words[lane] = mine;
    const unsigned long long anyhit = __ballot(mine != 0), anybad = __ballot(bad);
    if (lane == 0) {
        if (err) { if (tile == 0) results[blockIdx.y] = err; }
        else if (anybad) atomicMin(&results[blockIdx.y], E_CORRUPT);
        else if (anyhit) atomicCAS(&results[blockIdx.y], 0, 2);  // leaves an error another tile reported
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_select: the rows of a decoded pack that a bitmap picks, in row order (DataPack.foreach(BitMap position, XxxSetter),
// DataPack.java:273-306, 324-357, 375-408, 426-459, 708-762: every set bit below objCount; BitMap.ALL is every row).
// One workgroup of 16 waves per descriptor; a wave owns 64 bitmap words (4096 rows), one word per lane.  Popcounts give
// every word's first output position (wave scan, one workgroup exchange of the 16 wave totals); then the wave goes
// through its words, lane l looking at bit l: the picked rows of a word land next to each other, so loads and stores
// stay contiguous.  Strings add the same two steps for byte positions (sum of the picked lengths).
//   numbers   int32 rowId[n] | value[n] (8-byte values start at a multiple of 8)
//   strings   int32 rowId[n] | int32 arrayOffset[n] | int32 arrayLength[n] | the picked rows' bytes one after the other
// counts[d] = number of picked rows, or a negative error.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_select(const SelectDesc* __restrict__ descs, const uint8_t* __restrict__ decoded,
                                                 const uint8_t* __restrict__ bitmaps, uint8_t* __restrict__ out, int32_t* __restrict__ counts) {
    // readlane hands back a signed int: widen through uint32_t
    auto lane64 = [](unsigned long long v, int j) {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), j) << 32) |
               (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((uint32_t)v, j);
    };
    __shared__ uint32_t s_cnt[16];
    __shared__ unsigned long long s_bytes[16];
    __shared__ int s_bad;
    const SelectDesc D = descs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = (uint32_t)D.obj_count;
    const uint8_t* src = decoded + D.src_off;
    uint8_t* dst = out + D.dst_off;
    int32_t* row_id = (int32_t*)dst;
    const int dt = D.data_type;
    const uint32_t w = (dt == 1 || dt == 3) ? 4u : 8u;
    if (dt == 5 ? 4ull * (n + 1) > (uint64_t)D.data_size : (uint64_t)n * w > (uint64_t)D.data_size) {
        if (tid == 0) counts[blockIdx.x] = -1000;
        return;
    }
    if (tid == 0) s_bad = 0;
    // this lane's word: rows [64 * wi, 64 * wi + 64), bits at and above objCount dropped
    const uint32_t wi = (uint32_t)wave * 64u + (uint32_t)lane;
    unsigned long long word = 0;
    if (wi * 64u < n) {
        word = D.all ? ~0ull : ((const unsigned long long*)(bitmaps + D.bitmap_off))[wi];
        if (n - wi * 64u < 64u) word &= (1ull << (n - wi * 64u)) - 1ull;
    }
    const uint32_t c = (uint32_t)__popcll(word);
    const uint32_t c_incl = wave_incl_scan<uint32_t>(c);
    if (lane == 63) s_cnt[wave] = c_incl;
    const unsigned long long below = (1ull << lane) - 1ull;

    const int32_t* off = (const int32_t*)src;
    const uint8_t* body = src + 4ull * (n + 1);
    const int64_t body_cap = (int64_t)D.data_size - 4ll * (n + 1);
    bool bad = false;
    if (dt == 5) {  // strings: the bytes this wave picks, for the waves after it
        unsigned long long mine = 0;
        for (int j = 0; j < 64; j++) {
            const unsigned long long wj = lane64(word, j);
            if (!((wj >> lane) & 1ull)) continue;
            const uint32_t r = ((uint32_t)wave * 64u + (uint32_t)j) * 64u + (uint32_t)lane;
            const int32_t a = off[r], b = off[r + 1];
            if (a < 0 || b < a || (int64_t)b > body_cap) bad = true;
            else mine += (uint32_t)(b - a);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if (lane == 0) s_bytes[wave] = mine;
    }
    __syncthreads();
    uint32_t count_base = 0, total = 0;
    unsigned long long byte_base = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t v = s_cnt[k];
        count_base += k < wave ? v : 0u;
        total += v;
        if (dt == 5) byte_base += k < wave ? s_bytes[k] : 0ull;
    }
    const uint32_t c_excl = count_base + c_incl - c;  // first output position of this lane's word

    int32_t* aoff = row_id + n;
    int32_t* alen = aoff + n;
    uint8_t* bytes = dst + 12ull * n;
    uint8_t* vals = dst + ((4ull * n + 7) & ~7ull);
    for (int j = 0; j < 64; j++) {
        const unsigned long long wj = lane64(word, j);
        if (wj == 0) continue;
        const bool pick = (wj >> lane) & 1ull;
        const uint32_t r = ((uint32_t)wave * 64u + (uint32_t)j) * 64u + (uint32_t)lane;
        const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane(c_excl, j) + (uint32_t)__popcll(wj & below);
        if (dt != 5) {
            if (pick) {
                row_id[pos] = (int32_t)r;
                if (w == 4) ((uint32_t*)vals)[pos] = ((const uint32_t*)src)[r];
                else ((uint64_t*)vals)[pos] = ((const uint64_t*)src)[r];
            }
            continue;
        }
        int32_t a = 0;
        uint32_t len = 0;
        bool ok = pick;
        if (pick) {
            a = off[r];
            const int32_t b = off[r + 1];
            if (a < 0 || b < a || (int64_t)b > body_cap) ok = false;  // reported by the first pass
            else len = (uint32_t)(b - a);
        }
        const unsigned long long l_incl = wave_incl_scan<unsigned long long>((unsigned long long)len);
        const unsigned long long bp = byte_base + l_incl - len;
        byte_base += lane64(l_incl, 63);  // the next word's bytes follow this word's
        // rows whose offsets step backwards in between can overlap: more bytes than the pack's body holds do not fit
        if (ok && bp + len > (unsigned long long)body_cap) { bad = true; ok = false; }
        if (ok) {
            row_id[pos] = (int32_t)r;
            aoff[pos] = (int32_t)bp;
            alen[pos] = (int32_t)len;
            // Neither side is aligned; the memory system takes unaligned accesses.  No byte loops: a piece that ends at the
            // row's last byte covers the tail and may overlap the piece before it (same bytes, same lane, program order).
            const uint8_t* sp = body + a;
            uint8_t* dp = bytes + bp;
            auto piece = [&](uint32_t at, auto tag) {
                decltype(tag) v;
                __builtin_memcpy(&v, sp + at, sizeof(v));
                __builtin_memcpy(dp + at, &v, sizeof(v));
            };
            if (len >= 16) {
                uint32_t k = 0;
                for (; k + 64 <= len; k += 64) {
                    uint4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) __builtin_memcpy(&v[q], sp + k + 16 * q, 16);
#pragma unroll
                    for (int q = 0; q < 4; q++) __builtin_memcpy(dp + k + 16 * q, &v[q], 16);
                }
                const uint32_t m = (len - k) >> 4;  // 0..3 whole pieces left, then the tail piece
                uint4 v[4];
#pragma unroll
                for (uint32_t q = 0; q < 3; q++)
                    if (q < m) __builtin_memcpy(&v[q], sp + k + 16 * q, 16);
                __builtin_memcpy(&v[3], sp + len - 16, 16);
#pragma unroll
                for (uint32_t q = 0; q < 3; q++)
                    if (q < m) __builtin_memcpy(dp + k + 16 * q, &v[q], 16);
                __builtin_memcpy(dp + len - 16, &v[3], 16);
            } else if (len >= 8) {
                piece(0, uint2{}); piece(len - 8, uint2{});
            } else if (len >= 4) {
                piece(0, uint32_t{}); piece(len - 4, uint32_t{});
            } else if (len >= 2) {
                piece(0, uint16_t{}); piece(len - 2, uint16_t{});
            } else if (len == 1) {
                dp[0] = sp[0];
            }
        }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid == 0) counts[blockIdx.x] = s_bad ? E_CORRUPT : (int32_t)total;
}

// ---------------------------------------------------------------------------------------------------------------
// k_handback: decoded pack -> Spark column vector contents (DefaultPackReader.readByCopy / readBySet,
// DefaultPackReader.java:164-241).  grid = (blocks per descriptor, descriptors); elementwise, memory bound.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_handback(const HandbackDesc* __restrict__ descs, const uint8_t* __restrict__ decoded,
                                                  uint8_t* __restrict__ out, int32_t* __restrict__ status) {
    const HandbackDesc D = descs[blockIdx.y];
    const uint8_t* src = decoded + D.src_off;
    uint8_t* dst = out + D.dst_off;
    const uint32_t n = (uint32_t)D.obj_count, r0 = (uint32_t)D.row_offset, cnt = (uint32_t)D.row_count;
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
    const bool reporter = tid == 0;
    switch (D.sql_type) {
        case 1: case 3:  // INT, FLOAT: the rows as they are
            if ((uint64_t)n * 4 > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            for (uint32_t i = tid; i < cnt; i += nthreads) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[r0 + i];
            if (reporter) status[blockIdx.y] = (int32_t)(cnt * 4);
            return;
        case 2: case 4:  // BIGINT, DOUBLE
            if ((uint64_t)n * 8 > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            for (uint32_t i = tid; i < cnt; i += nthreads) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[r0 + i];
            if (reporter) status[blockIdx.y] = (int32_t)(cnt * 8);
            return;
        case 6:  // DATE: (int) (ms / 86400000), truncating division, low 32 bits
            if ((uint64_t)n * 8 > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            for (uint32_t i = tid; i < cnt; i += nthreads) {
                const int64_t v = ((const int64_t*)src)[r0 + i];
                ((int32_t*)dst)[i] = (int32_t)(uint32_t)(uint64_t)(v / 86400000LL);
            }
            if (reporter) status[blockIdx.y] = (int32_t)(cnt * 4);
            return;
        case 8:  // DATETIME: ms * 1000, wrapping
            if ((uint64_t)n * 8 > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            for (uint32_t i = tid; i < cnt; i += nthreads) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[r0 + i] * 1000ull;
            if (reporter) status[blockIdx.y] = (int32_t)(cnt * 8);
            return;
        case 5: {  // VARCHAR: arrayOffset[cnt] | arrayLength[cnt] | bytes of the rows in order
            if (4ull * (n + 1) > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            const int32_t* off = (const int32_t*)src;
            const int32_t a = off[r0], b = off[r0 + cnt];
            if (a < 0 || b < a || 4ull * (n + 1) + (uint64_t)b > (uint64_t)D.data_size) { if (reporter) status[blockIdx.y] = -1000; return; }
            int32_t* aoff = (int32_t*)dst;
            int32_t* alen = aoff + cnt;
            for (uint32_t i = tid; i < cnt; i += nthreads) {
                const int32_t s = off[r0 + i], e = off[r0 + i + 1];
                aoff[i] = s - a;
                alen[i] = e - s;
            }
            // the rows' bytes are contiguous in the decoded body: one unaligned block copy, 16 bytes per lane
            const uint8_t* sb = src + 4ull * (n + 1) + (uint32_t)a;
            uint8_t* db = dst + 8ull * cnt;
            const uint32_t len = (uint32_t)(b - a), whole = len & ~15u;
            for (uint32_t i = tid * 16u; i < whole; i += nthreads * 16u) {
                uint4 v;
                __builtin_memcpy(&v, sb + i, 16);
                __builtin_memcpy(db + i, &v, 16);
            }
            for (uint32_t i = whole + tid; i < len; i += nthreads) db[i] = sb[i];
            if (reporter) status[blockIdx.y] = (int32_t)(8u * cnt + len);
            return;
        }
        default:
            if (reporter) status[blockIdx.y] = -102;  // "Unsupported date type" (DefaultPackReader.java:233-234)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_checksum: position-weighted checksum of every decoded pack (same formula as ixr_checksum on the host), so that
// full-size outputs can be verified where they lie.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_checksum(const DevPack* __restrict__ packs, int count, const uint8_t* __restrict__ out,
                                                  unsigned long long* __restrict__ sums) {
    const int pi = blockIdx.x;
    if (pi >= count) return;
    const DevPack P = packs[pi];
    const uint64_t bytes = (uint64_t)P.data_size;
    const uint8_t* p = out + P.out_off;
    const uint64_t words = bytes >> 3;
    unsigned long long sum = 0;
    for (uint64_t i = threadIdx.x; i < words; i += 256) {
        const unsigned long long w = ((const unsigned long long*)p)[i];
        sum += (w + 0x9E3779B97F4A7C15ull) * (2 * i + 1);
    }
    if (threadIdx.x == 0 && (bytes & 7)) {
        unsigned long long w = 0;
        for (uint32_t k = 0; k < (bytes & 7); k++) w |= (unsigned long long)p[8 * words + k] << (8 * k);
        sum += (w + 0x9E3779B97F4A7C15ull) * (2 * words + 1);
    }
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) sums[pi] = part[0] + part[1] + part[2] + part[3];
}

// ---------------------------------------------------------------------------------------------------------------
// The batch that allocates nothing: its descriptors arrive as a kernel argument, one thread turns them into the device
// records the decode kernels read (the same layout function the host used for its bound check).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_static_setup(StaticBatch batch, uint8_t* block) {
    DevPack* packs = (DevPack*)block;
    PackMeta* metas = (PackMeta*)(block + kStaticMetaOff);
    int* lists = (int*)(block + kStaticListOff);
    static_assert(sizeof(DevPack) * kStaticMaxPacks <= kStaticMetaOff, "records overlap");
    static_assert(kStaticMetaOff + sizeof(PackMeta) * kStaticMaxPacks <= kStaticListOff, "records overlap");
    static_assert(kStaticListOff + sizeof(int) * SL_COUNT * kStaticMaxPacks <= kStaticHeader, "records overlap");
    // the last call's results must not be mistaken for this call's: every record starts at zero
    for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(PackMeta) * kStaticMaxPacks / 4); i += blockDim.x) ((uint32_t*)metas)[i] = 0;
    if (threadIdx.x == 0) {
        int counts[SL_COUNT];
        static_layout(batch.d, batch.n, batch.dict_cap, packs, lists, counts);
    }
}
__global__ void k_export_status(const PackMeta* __restrict__ metas, int count, int32_t* __restrict__ d_status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) d_status[i] = metas[i].status;
}

// ===============================================================================================================
// launchers
// ===============================================================================================================
static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

void launch_parse(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    // Thousands of packs: one wave per stream.  Fewer: one workgroup per pack, a wave per stream, all of them resident at
    // once either way; with 16 KB windows when even those leave the chip mostly empty.
    const int waves_from = c.wave_walk_from > 0 ? c.wave_walk_from : kParseWaveThreshold;
    if (count >= waves_from) {
        hipLaunchKernelGGL(k_parse_head, dim3(cdiv((unsigned)count, 256)), dim3(256), 0, c.stream, c.packs, list, count, c.metas, c.in);
        hipLaunchKernelGGL(k_parse_waves<Win>, dim3((unsigned)count, 3), dim3(64), 0, c.stream, c.packs, list, c.metas, c.in, c.scratch);
        return;
    }
    if (count < 1024)
        hipLaunchKernelGGL(k_parse<DeltaWin>, dim3((unsigned)count), dim3(256), 0, c.stream, c.packs, list, count, c.metas, c.in, c.scratch);
    else
        hipLaunchKernelGGL(k_parse<Win>, dim3((unsigned)count), dim3(256), 0, c.stream, c.packs, list, count, c.metas, c.in, c.scratch);
}
void launch_delta(const LaunchCtx& c, const int* list, int count, int width, int streams) {
    if (count <= 0) return;
    // the kernel's mode word: 0 rows and dictionaries, 1 dictionaries only, 2 as 0 with sorted dictionaries that are taken in
    // order left to the slice path of k_dict_num
    const int mode = c.dict_only ? 1 : (c.slice ? 2 : 0);
    const bool few = (long)count * streams < 512;  // fewer streams than two per compute unit: long workgroups, short chains
    if (width == 64) {
        if (few) hipLaunchKernelGGL((k_delta<uint64_t, 1024>), dim3(count, streams), dim3(1024), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch, mode);
        else hipLaunchKernelGGL((k_delta<uint64_t, 256>), dim3(count, streams), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch, mode);
    } else {
        if (few) hipLaunchKernelGGL((k_delta<uint32_t, 1024>), dim3(count, streams), dim3(1024), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch, mode);
        else hipLaunchKernelGGL((k_delta<uint32_t, 256>), dim3(count, streams), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch, mode);
    }
}
void launch_dict_num(const LaunchCtx& c, const int* list, int count, int width) {
    if (count <= 0) return;
    const unsigned gx = IXR_PACK_TILES / 4;
    // every pack is taken by exactly one of the two kernels: by the size of its dictionary
    if (width == 64) {
        hipLaunchKernelGGL(k_dict_lds<uint64_t>, dim3(count), dim3(1024), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
        hipLaunchKernelGGL(k_dict_num<uint64_t>, dim3(gx, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out,
                           c.scratch);
    } else {
        hipLaunchKernelGGL(k_dict_lds<uint32_t>, dim3(count), dim3(1024), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
        hipLaunchKernelGGL(k_dict_num<uint32_t>, dim3(gx, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out,
                           c.scratch);
    }
}
void launch_copy_raw(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_copy_raw, dim3(8, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out);
}
void launch_str_dict_scan(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_str_dict_scan, dim3(count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch,
                       c.dict_only);
}
void launch_front_serial(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_front_serial, dim3(4, count), dim3(64), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch,
                       c.dict_only);
}
void launch_front_summary(const LaunchCtx& c, const int* list, int count, int use_ds) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_front_summary, dim3(32, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch,
                       (const uint8_t*)c.scratch, use_ds);
}
void launch_front_combine(const LaunchCtx& c, const int* list, int count, int use_ds) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_front_combine, dim3(4, count), dim3(64), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch, use_ds);
}
void launch_front_decode(const LaunchCtx& c, const int* list, int count, int use_ds) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_front_decode, dim3(16, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch,
                       (const uint8_t*)c.scratch, use_ds);
}
void launch_str_index(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    // few packs: 16 waves each shorten the one pack's chain of rounds; many packs: workgroups of 4 waves, which find room on
    // a compute unit beside the numeric kernels of the other stream
    if (count < 1024) hipLaunchKernelGGL((k_str_index<16>), dim3(count), dim3(1024), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
    else hipLaunchKernelGGL((k_str_index<4>), dim3(count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
}
void launch_str_bases(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_str_bases<kBasesTabSmall>, dim3(count), dim3(kBasesWaves * 64), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
    hipLaunchKernelGGL(k_str_bases<kBasesTab>, dim3(count), dim3(kBasesWaves * 64), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
}
void launch_str_fused(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_str_fused, dim3(IXR_PACK_TILES / 4, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
}
void launch_row_scan(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_row_scan, dim3(count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.out);
}
void launch_str_copy(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_str_copy, dim3(IXR_PACK_MAX_ROWS_ / kStrTileRows, count), dim3(256), 0, c.stream, c.packs, list, c.metas,
                       c.out, c.scratch);
}
void launch_ds_meta(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_ds_meta, dim3(cdiv((unsigned)count, 64)), dim3(64), 0, c.stream, c.packs, list, count, c.metas, c.in);
}
void launch_ds_rows(const LaunchCtx& c, const int* list, int count, int mode) {
    if (count <= 0) return;
    const dim3 g(IXR_PACK_MAX_ROWS_ / 256, count), b(256);
    if (mode == 0) hipLaunchKernelGGL(k_ds_rows<0>, g, b, 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
    else if (mode == 1) hipLaunchKernelGGL(k_ds_rows<1>, g, b, 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
    else hipLaunchKernelGGL(k_ds_rows<2>, g, b, 0, c.stream, c.packs, list, c.metas, c.in, c.out, c.scratch);
}
void launch_filter_search(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint8_t* values, uint32_t* matches) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_filter_search, dim3(count), dim3(64), 0, c.stream, c.packs, list, c.metas, c.scratch, filters, values, matches);
}
void launch_filter_like(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint8_t* values, uint32_t* matches) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_filter_like, dim3(count), dim3(kLikeThreads), 0, c.stream, c.packs, list, c.metas, c.scratch, filters, values, matches);
}
void launch_filter_rows(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint32_t* matches,
                        unsigned long long* bitmaps) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_filter_rows, dim3(IXR_PACK_TILES / 4, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in, filters,
                       matches, bitmaps, c.scratch);
}
void launch_scan_filter(hipStream_t stream, const ScanFilterDesc* descs, int count, const uint8_t* decoded, const uint8_t* values,
                        unsigned long long* bitmaps, int32_t* results) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_scan_filter, dim3(IXR_PACK_TILES / 4, count), dim3(256), 0, stream, descs, decoded, values, bitmaps, results);
}
void launch_select(hipStream_t stream, const SelectDesc* descs, int count, const uint8_t* decoded, const uint8_t* bitmaps, uint8_t* out,
                   int32_t* counts) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_select, dim3(count), dim3(1024), 0, stream, descs, decoded, bitmaps, out, counts);
}
void launch_handback(hipStream_t stream, const HandbackDesc* descs, int count, const uint8_t* decoded, uint8_t* out, int32_t* status) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_handback, dim3(16, count), dim3(256), 0, stream, descs, decoded, out, status);
}
void launch_checksum(const LaunchCtx& c, int count, unsigned long long* sums) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_checksum, dim3(count), dim3(256), 0, c.stream, c.packs, count, c.out, sums);
}
void launch_static_setup(hipStream_t stream, const StaticBatch& batch, uint8_t* block) {
    hipLaunchKernelGGL(k_static_setup, dim3(1), dim3(64), 0, stream, batch, block);
}
void launch_export_status(hipStream_t stream, const PackMeta* metas, int count, int32_t* d_status) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_export_status, dim3(cdiv((unsigned)count, 64)), dim3(64), 0, stream, metas, count, d_status);
}
void launch_dict_struct(const LaunchCtx& c, const int* list, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_dict_struct, dim3(1 + IXR_PACK_TILES / 4, count), dim3(256), 0, c.stream, c.packs, list, c.metas, c.in,
                       c.out, c.scratch);
}

}  // namespace ixr
