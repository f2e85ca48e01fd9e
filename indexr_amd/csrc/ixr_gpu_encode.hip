// ixr_gpu_encode.hip -- pack encoder on the device: the batch form of the write path
//   DataPackCreator_VLT.from              indexr-vlt-segment/.../segment/pack/DataPackCreator_VLT.java:21-213 (dpn statistics)
//   PackCompressor_VLT.compress           indexr-vlt-segment/.../segment/pack/PackCompressor_VLT.java:23-128
//   DictCompressCodec.encode*             .../codec/dict/DictCompressCodec.java:40-69
//   DictCodec.encode* (sorted dictionary) .../codec/dict/DictCodec.java:79-131, 154-206, 229-281, 304-356, 379-433
//   DictStructUtil.compress               .../codec/dict/DictStructUtil.java:25-81
//   DeltaCodec.encodeInts / encodeLongs   .../codec/delta/DeltaCodec.java:40-97, 144-201; encodeStrings :268-352
//   RLEPackingHybridCodec.encodeInts      .../codec/rle/RLEPackingHybridCodec.java:38-94
// The bytes it writes are those of the host encoder in ixr_encode.cpp (and of the reference), pack for pack.
//
// Per group of packs, numeric packs (INT, LONG, FLOAT, DOUBLE):
//   k_enc_stats  dpn.minValue / maxValue / sum (DataPackCreator_VLT.java:21-163)
//   k_enc_keys   rows -> unsigned keys whose order is the dictionary's order (signed ints; Float/Double.compare with the
//                NaNs collapsed); notes the first NaN row of a pack (its bits become the NaN entry)
//   sort         rocprim::segmented_radix_sort_keys, one segment per pack (the one library primitive of this library)
//   k_enc_dict   one workgroup per pack: distinct keys -> dictionary entries; if the dictionary is allowed
//                (<= ceil(n * ratio) entries) every row's id by the reference's plain < / > binary search
//   k_enc_pack   one workgroup per pack: DICT_COMPRESS (entries plain or DELTA, ids RLE / bit-packing hybrid) or DELTA
//                over the rows.  Sizes first (workgroup scans over blocks / runs / segments), then every block, header
//                and group of 8 values is written at its final position independently.
// String packs:
//   k_enc_str    one workgroup per pack, everything from the StringsStruct to the DICT_COMPRESS pack (see the kernel)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <type_traits>
#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <vector>

#include "../../include/ixr_gpu.h"
#include "ixr_launch.h"

namespace ixr {

constexpr int kEncThreads = 1024, kEncWaves = kEncThreads / 64;
constexpr int32_t kEncOverflow = -1, kEncBug = -100, kEncImproper = -3;

struct EncPack {
    uint64_t src_off, dst_off;
    int32_t data_type, is_indexed, obj_count, dst_cap;
    int32_t max_entries;  // ceil((float)n * ratio), DictCodec.java:83
    int32_t data_size;
    uint64_t key_off;     // element offset of the pack's keys in the group's key arrays (of its width)
    uint64_t s_entries;   // scratch: dictionary entries, raw bit patterns, obj_count * width bytes
    uint64_t s_ids;       // scratch: uint16 id per row
    uint64_t s_run_of;    // scratch: uint16 run index per row
    uint64_t s_heads;     // scratch: uint32 first row of every run, obj_count + 1 entries
    uint64_t s_seg_of;    // scratch: uint16 segment index per run
    uint64_t s_segs;      // scratch: uint32 first row | RLE flag << 31 per segment, obj_count + 1 entries
    uint64_t s_pos;       // scratch: uint32 byte position of every segment
    // string packs only
    uint64_t s_table;     // scratch: uint32 hash table of rows (first row seen with each distinct string), 2 * n2 slots
    uint64_t s_rep;       // scratch: uint32 per row: the row that represents its string in the table
    uint64_t s_rank;      // scratch: uint16 per row: dictionary id of the string a representative row holds
    uint64_t s_perm;      // scratch: uint32 representative rows, sorted by their strings: slot e holds entry e (n2 slots)
    uint64_t s_key;       // scratch: uint64 per sort slot: the first 8 bytes of the slot's string, big endian (n2 slots)
    uint64_t s_pre;       // scratch: uint32 prefixLen per entry (front coding)
    uint64_t s_suf;       // scratch: uint32 suffixLen per entry
    uint64_t s_ent_off;   // scratch: uint32 offset of entry e in the dictionary body, obj_count + 1 entries
    uint64_t s_suf_off;   // scratch: uint32 offset of entry e's suffix in the suffix bytes, obj_count + 1 entries
};
struct EncMeta {
    int32_t first_nan;    // first row holding a NaN (INT32_MAX: none)
    int32_t entry_count;
    int32_t use_dict;
    int32_t pack_size;    // out: compressed size or a negative error
    int32_t encode_type;  // out
    int32_t pad[3];
    int64_t min_value, max_value, sum;  // out: DataPackNode statistics
};

// ---------------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
// Exclusive prefix sum over the 1024 threads of the workgroup; *total receives the sum.  s_w: 17 words of LDS.
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const uint32_t x = lane < kEncWaves ? s_w[lane] : 0u;
        const uint32_t xi = wave_incl_scan(x, lane);
        if (lane < kEncWaves) s_w[lane] = xi - x;
        if (lane == kEncWaves - 1) s_w[kEncWaves] = xi;
    }
    __syncthreads();
    const uint32_t r = s_w[wave] + incl - v;
    *total = s_w[kEncWaves];
    __syncthreads();
    return r;
}
__device__ __forceinline__ int bit_width64(uint64_t v) { return v ? 64 - __clzll((long long)v) : 0; }
__device__ __forceinline__ int uvarint_len(uint64_t v) {
    int n = 1;
    while (v >= 0x80) { v >>= 7; n++; }
    return n;
}
__device__ __forceinline__ int put_uvarint(uint8_t* p, uint64_t v) {  // VarInteger.writeUnsignedVarLong :25-47
    int n = 0;
    while (v >= 0x80) { p[n++] = (uint8_t)(v | 0x80); v >>= 7; }
    p[n++] = (uint8_t)v;
    return n;
}
__device__ __forceinline__ void put_u32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
template <typename U> __device__ __forceinline__ uint64_t zigzag(U v) {  // VarInteger zig-zag of the signed reading of v
    using S = typename std::make_signed<U>::type;
    return (uint64_t)(U)((v << 1) ^ (U)((S)v >> (sizeof(U) * 8 - 1)));
}
// `cnt` (1..8) values of `bw` (1..64) bits, value k at bit k*bw, into exactly bw bytes (IntPacker / LongPacker: a group of 8
// values always takes bw bytes, missing values are zero bits).
template <typename V> __device__ __forceinline__ void pack_group(uint8_t* out, const V* v, int cnt, int bw) {
    uint64_t lo = 0, hi = 0;
    int have = 0, written = 0;
    for (int k = 0; k < cnt; k++) {
        const uint64_t x = (uint64_t)v[k];
        lo |= x << have;
        if (have && have + bw > 64) hi |= x >> (64 - have);
        have += bw;
        while (have >= 8) {
            out[written++] = (uint8_t)lo;
            lo = (lo >> 8) | (hi << 56);
            hi >>= 8;
            have -= 8;
        }
    }
    if (have > 0) out[written++] = (uint8_t)lo;
    for (; written < bw; written++) out[written] = 0;
}

// order-preserving unsigned keys
__device__ __forceinline__ uint32_t fkey(uint32_t b, bool* nan) {  // Float.compare: -0.0 < 0.0, every NaN alike and greatest
    *nan = (b & 0x7fffffffu) > 0x7f800000u;
    if (*nan) b = 0x7fc00000u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t dkey(uint64_t b, bool* nan) {
    *nan = (b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
    if (*nan) b = 0x7ff8000000000000ull;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
template <typename U> __device__ __forceinline__ U unkey_fp(U k) {
    constexpr U sign = (U)1 << (sizeof(U) * 8 - 1);
    return (k & sign) ? (k ^ sign) : ~k;
}

// ---------------------------------------------------------------------------------------------------------------
// k_enc_keys
// ---------------------------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(1024) void k_enc_keys(const EncPack* __restrict__ packs, const int* __restrict__ list, EncMeta* __restrict__ metas,
                                                   const uint8_t* __restrict__ data, U* __restrict__ keys) {
    const int pi = list[blockIdx.y];
    const EncPack& P = packs[pi];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i >= (uint32_t)P.obj_count) return;
    const U b = ((const U*)(data + P.src_off))[i];
    constexpr U sign = (U)1 << (sizeof(U) * 8 - 1);
    U k;
    if (P.data_type == IXR_INT || P.data_type == IXR_LONG) k = b ^ sign;
    else {
        bool nan;
        if constexpr (sizeof(U) == 4) k = fkey(b, &nan);
        else k = dkey(b, &nan);
        if (nan) atomicMin(&metas[pi].first_nan, (int32_t)i);
    }
    keys[P.key_off + i] = k;
}

// ---------------------------------------------------------------------------------------------------------------
// k_enc_stats: dpn.minValue / maxValue / sum as DataPackCreator_VLT computes them (DataPackCreator_VLT.java:21-163).
//   INT / LONG      min, max, wrapping 64-bit sum
//   FLOAT / DOUBLE  the fold min = Math.min(v, min), max = Math.max(v, max): -0.0 < 0.0, and a NaN replaces whatever was
//                   there and stays until the next NaN, so both end as the last NaN of the pack when there is one; stored
//                   as Double.doubleToRawLongBits of the (widened) value (NumType.java:93-107).  The sum is accumulated row
//                   after row in the column's own precision (float sum / double sum), which no tree reduction reproduces:
//                   the rows are staged through LDS and one thread adds them in order.
// One workgroup of 256 threads per pack.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kStatThreads = 256, kStatTile = 2048;
template <typename U>
__global__ __launch_bounds__(kStatThreads) void k_enc_stats(const EncPack* __restrict__ packs, const int* __restrict__ list, EncMeta* __restrict__ metas,
                                                            const uint8_t* __restrict__ data) {
    using S = typename std::make_signed<U>::type;
    using F = typename std::conditional<sizeof(U) == 4, float, double>::type;
    __shared__ U s_tile[kStatTile];
    __shared__ U s_min[kStatThreads / 64], s_max[kStatThreads / 64];
    __shared__ unsigned long long s_sum[kStatThreads / 64];
    __shared__ int s_nan[kStatThreads / 64];
    const int pi = list[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const EncPack P = packs[pi];
    const int n = P.obj_count;
    const U* rows = (const U*)(data + P.src_off);
    const bool fp = P.data_type == IXR_FLOAT || P.data_type == IXR_DOUBLE;
    constexpr U sign = (U)1 << (sizeof(U) * 8 - 1);
    U kmin = ~(U)0, kmax = 0;  // order-preserving unsigned keys
    unsigned long long isum = 0;
    int last_nan = -1;
    F fsum = 0;
    for (int base = 0; base < n; base += kStatTile) {
#pragma unroll
        for (int q = 0; q < kStatTile / kStatThreads; q++) {
            const int i = base + q * kStatThreads + tid;
            if (i < n) {
                const U b = rows[i];
                if (fp) {
                    s_tile[q * kStatThreads + tid] = b;
                    const bool nan = sizeof(U) == 4 ? (b & 0x7fffffffu) > 0x7f800000u : (b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
                    if (nan) last_nan = i;
                    else {
                        const U k = (b & sign) ? ~b : (b | sign);
                        kmin = k < kmin ? k : kmin;
                        kmax = k > kmax ? k : kmax;
                    }
                } else {
                    const U k = b ^ sign;
                    kmin = k < kmin ? k : kmin;
                    kmax = k > kmax ? k : kmax;
                    isum += (unsigned long long)(long long)(S)b;
                }
            }
        }
        if (fp) {  // workgroup-uniform
            __syncthreads();
            if (tid == 0) {
                const int c = min(kStatTile, n - base);
                for (int k = 0; k < c; k++) {
                    F v;
                    if constexpr (sizeof(U) == 4) v = __uint_as_float(s_tile[k]);
                    else v = __longlong_as_double((long long)s_tile[k]);
                    fsum += v;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const U a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        isum += __shfl_xor(isum, d);
        last_nan = max(last_nan, __shfl_xor(last_nan, d));
    }
    if (lane == 0) { s_min[wave] = kmin; s_max[wave] = kmax; s_sum[wave] = isum; s_nan[wave] = last_nan; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kStatThreads / 64; w++) {
            kmin = s_min[w] < kmin ? s_min[w] : kmin;
            kmax = s_max[w] > kmax ? s_max[w] : kmax;
            isum += s_sum[w];
            last_nan = max(last_nan, s_nan[w]);
        }
        EncMeta& M = metas[pi];
        if (!fp) {
            M.min_value = (int64_t)(S)(kmin ^ sign);
            M.max_value = (int64_t)(S)(kmax ^ sign);
            M.sum = (int64_t)isum;
        } else {
            U bmin, bmax;
            if (last_nan >= 0) bmin = bmax = rows[last_nan];
            else {
                bmin = (kmin & sign) ? (kmin ^ sign) : ~kmin;
                bmax = (kmax & sign) ? (kmax ^ sign) : ~kmax;
            }
            if constexpr (sizeof(U) == 4) {
                M.min_value = __double_as_longlong((double)__uint_as_float(bmin));
                M.max_value = __double_as_longlong((double)__uint_as_float(bmax));
            } else {
                M.min_value = (int64_t)bmin;
                M.max_value = (int64_t)bmax;
            }
            M.sum = __double_as_longlong((double)fsum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_enc_dict
// ---------------------------------------------------------------------------------------------------------------
// fastutil's binarySearch as DictCodec uses it: plain < and > on the primitive, the first probe that is neither wins
template <typename T> __device__ __forceinline__ int plain_search(const T* e, int m, T key) {
    int lo = 0, hi = m - 1;
    while (lo <= hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        const T v = e[mid];
        if (v < key) lo = mid + 1;
        else if (v > key) hi = mid - 1;
        else return mid;
    }
    return -1;
}

constexpr int kDictSamples = 2048;  // dictionary keys kept in LDS for the first level of the id search
template <typename U>
__global__ __launch_bounds__(kEncThreads) void k_enc_dict(const EncPack* __restrict__ packs, const int* __restrict__ list, EncMeta* __restrict__ metas,
                                                          const uint8_t* __restrict__ data, const U* __restrict__ sorted, uint8_t* __restrict__ scratch) {
    __shared__ uint32_t s_w[kEncWaves + 1];
    __shared__ U s_keys[kDictSamples];
    const int pi = list[blockIdx.x];
    const EncPack P = packs[pi];
    EncMeta& M = metas[pi];
    const int n = P.obj_count, tid = threadIdx.x;
    const U* ks = sorted + P.key_off;
    const U* rows = (const U*)(data + P.src_off);
    U* ent = (U*)(scratch + P.s_entries);
    const bool fp = P.data_type == IXR_FLOAT || P.data_type == IXR_DOUBLE;
    constexpr U sign = (U)1 << (sizeof(U) * 8 - 1);
    const int first_nan = M.first_nan;
    uint32_t carry = 0;
    for (int base = 0; base < n; base += kEncThreads) {
        const int i = base + tid;
        const bool head = i < n && (i == 0 || ks[i] != ks[i - 1]);
        uint32_t tot;
        const uint32_t ex = wg_excl_scan(head ? 1u : 0u, s_w, &tot);
        if (head) {
            const U k = ks[i];
            U bits;
            if (!fp) bits = k ^ sign;
            else {
                bits = unkey_fp<U>(k);
                const bool nan = sizeof(U) == 4 ? (bits & 0x7fffffffu) > 0x7f800000u : (bits & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
                if (nan && first_nan < n) bits = rows[first_nan];  // the set keeps the first NaN it was given
            }
            ent[carry + ex] = bits;
        }
        carry += tot;
    }
    const int ec = (int)carry;
    const bool use_dict = ec <= P.max_entries;
    if (tid == 0) { M.entry_count = ec; M.use_dict = use_dict ? 1 : 0; }
    if (!use_dict) return;
    __syncthreads();  // the entries written above are read below by other threads of this workgroup
    // Every row's value is in the dictionary, at one place: any search finds the id the reference's binarySearch finds.
    // Two levels: every stride-th entry's key is kept in LDS, the rest of the search touches at most log2(stride) entries
    // in memory.  Only zeros and NaNs of float columns depend on the reference's probe order (a zero equals the zero of
    // the other sign, a NaN "equals" the first entry probed): those rows take the plain search.
    auto key_of = [&](U bits) -> U {
        if (!fp) return bits ^ sign;
        bool nan;
        if constexpr (sizeof(U) == 4) return fkey(bits, &nan);
        else return dkey(bits, &nan);
    };
    const int stride = (ec + kDictSamples - 1) / kDictSamples;
    const int ns = (ec + stride - 1) / stride;
    for (int j = tid; j < ns; j += kEncThreads) s_keys[j] = key_of(ent[(size_t)j * stride]);
    __syncthreads();
    uint16_t* ids = (uint16_t*)(scratch + P.s_ids);
    bool bad = false;
    for (int i = tid; i < n; i += kEncThreads) {
        const U b = rows[i];
        int id;
        bool special = false;
        if (fp) {
            const U mag = b & ~sign;
            special = mag == 0 || (sizeof(U) == 4 ? mag > (U)0x7f800000u : mag > (U)0x7ff0000000000000ull);
        }
        if (special) {
            if constexpr (sizeof(U) == 4) id = plain_search<float>((const float*)ent, ec, __uint_as_float(b));
            else id = plain_search<double>((const double*)ent, ec, __longlong_as_double((long long)b));
        } else {
            const U k = key_of(b);
            int lo = 0, hi = ns - 1;  // last sample <= k
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_keys[mid] <= k) lo = mid; else hi = mid - 1;
            }
            int a = lo * stride, z = min(a + stride, ec) - 1;
            id = -1;
            while (a <= z) {
                const int mid = (a + z) >> 1;
                const U v = key_of(ent[mid]);
                if (v < k) a = mid + 1;
                else if (v > k) z = mid - 1;
                else { id = mid; break; }
            }
        }
        if (id < 0) { bad = true; id = 0; }
        ids[i] = (uint16_t)id;
    }
    if (bad) M.pack_size = kEncBug;
}

// ---------------------------------------------------------------------------------------------------------------
// DELTA over n values (DeltaCodec.encodeInts :40-97 / encodeLongs :144-201), the whole workgroup on one stream:
//   u32 encodedLen | zz first | per 128 values: zz minDelta, per 32 values: u8 bw + ceil(c/8)*bw bytes of (delta - minDelta)
// Returns the bytes written (4 + encodedLen), or 0 when they would not fit into `cap` (nothing is written then).
// ---------------------------------------------------------------------------------------------------------------
template <typename U> struct DeltaLds {
    U rel[kEncWaves][128];
    U mn[512];
    uint8_t bw[512 * 4];
    uint32_t pos[512];
};
template <typename U>
__device__ uint32_t wg_delta_encode(const U* __restrict__ v, int n, uint8_t* __restrict__ dst, uint32_t cap, DeltaLds<U>& L, uint32_t* s_w) {
    using S = typename std::make_signed<U>::type;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int nb = (n + 127) >> 7;
    auto block_values = [&](int b, U& rel0, U& rel1, int& cnt, S& mn, int& bw) {
        const int i0 = b * 128 + lane * 2;
        cnt = min(128, n - b * 128);
        const bool ok0 = i0 < n, ok1 = i0 + 1 < n;
        const U v0 = ok0 ? v[i0] : (U)0, v1 = ok1 ? v[i0 + 1] : (U)0;
        const U prev = i0 > 0 && ok0 ? v[i0 - 1] : v0;  // the first value's delta is against itself
        const U d0 = v0 - prev, d1 = v1 - v0;
        S m = ok0 ? (S)d0 : std::numeric_limits<S>::max();
        if (ok1 && (S)d1 < m) m = (S)d1;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const S o = (S)__shfl_xor((U)m, s);
            if (o < m) m = o;
        }
        mn = m;
        rel0 = ok0 ? (U)(d0 - (U)m) : (U)0;
        rel1 = ok1 ? (U)(d1 - (U)m) : (U)0;
        U orv = rel0 | rel1;
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) orv |= __shfl_xor(orv, s);  // the 16 lanes of one mini-block
        bw = bit_width64((uint64_t)orv);
    };
    // sizes
    for (int b = wave; b < nb; b += kEncWaves) {
        U rel0, rel1; int cnt, bw; S mn;
        block_values(b, rel0, rel1, cnt, mn, bw);
        uint32_t size = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int bwm = __shfl(bw, m * 16);
            const int c = min(32, cnt - 32 * m);
            if (c > 0) size += 1u + (uint32_t)((c + 7) >> 3) * (uint32_t)bwm;
            if (lane == 0) L.bw[b * 4 + m] = (uint8_t)bwm;
        }
        if (lane == 0) { L.mn[b] = (U)mn; L.pos[b] = size + (uint32_t)uvarint_len(zigzag<U>((U)mn)); }
    }
    __syncthreads();
    uint32_t total;
    {
        const uint32_t mine = tid < nb ? L.pos[tid] : 0u;
        const uint32_t ex = wg_excl_scan(mine, s_w, &total);
        if (tid < nb) L.pos[tid] = ex;
    }
    const U first = v[0];
    const uint32_t flen = (uint32_t)uvarint_len(zigzag<U>(first));
    const uint32_t len = flen + total;
    if (4ull + len > (uint64_t)cap) return 0;
    if (tid == 0) { put_u32(dst, len); put_uvarint(dst + 4, zigzag<U>(first)); }
    __syncthreads();  // L.pos is final
    uint8_t* body = dst + 4 + flen;
    for (int b = wave; b < nb; b += kEncWaves) {
        U rel0, rel1; int cnt, bw; S mn;
        block_values(b, rel0, rel1, cnt, mn, bw);
        L.rel[wave][lane * 2] = rel0;
        L.rel[wave][lane * 2 + 1] = rel1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t* p = body + L.pos[b];
        const uint32_t hdr = (uint32_t)uvarint_len(zigzag<U>((U)mn));
        if (lane == 0) put_uvarint(p, zigzag<U>((U)mn));
        if (lane < 16) {  // one group of 8 values per lane
            const int m = lane >> 2, g = lane & 3;
            uint32_t off = hdr;
            for (int q = 0; q < m; q++) {
                const int c = min(32, cnt - 32 * q);
                off += 1u + (uint32_t)((c + 7) >> 3) * (uint32_t)L.bw[b * 4 + q];
            }
            const int c = min(32, cnt - 32 * m);
            const int bwm = L.bw[b * 4 + m];
            if (c > 0 && g == 0) p[off] = (uint8_t)bwm;
            const int first_v = 8 * g;
            if (first_v < c && bwm > 0) pack_group<U>(p + off + 1 + (uint32_t)g * (uint32_t)bwm, &L.rel[wave][32 * m + first_v], min(8, c - first_v), bwm);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next round overwrites L.rel[wave]
    }
    return 4u + len;
}

// ---------------------------------------------------------------------------------------------------------------
// RLE / bit-packing hybrid over the ids (RLEPackingHybridCodec.encodeInts :38-94):
//   u32 encodedLen | { uvarint run << 1, value in (bw+7)/8 bytes | uvarint count << 1 | 1, ceil(count/8)*bw bytes }*
// A run of equal ids becomes an RLE segment when it is >= 8 long and another value follows it; everything between two
// such runs, and whatever ends the input, is one bit-packed segment.  Returns bytes written or 0 when over `cap`.
// ---------------------------------------------------------------------------------------------------------------
__device__ uint32_t wg_rle_encode(const uint16_t* __restrict__ ids, int n, int bw, uint8_t* __restrict__ dst, uint32_t cap,
                                  uint16_t* __restrict__ run_of, uint32_t* __restrict__ heads, uint16_t* __restrict__ seg_of,
                                  uint32_t* __restrict__ segs, uint32_t* __restrict__ pos, uint32_t* s_w) {
    const int tid = threadIdx.x;
    const int bytew = (bw + 7) >> 3;
    // runs
    uint32_t carry = 0;
    for (int base = 0; base < n; base += kEncThreads) {
        const int i = base + tid;
        const bool head = i < n && (i == 0 || ids[i] != ids[i - 1]);
        uint32_t tot;
        const uint32_t ex = wg_excl_scan(head ? 1u : 0u, s_w, &tot);
        if (i < n) run_of[i] = (uint16_t)(carry + ex + (head ? 1u : 0u) - 1u);
        if (head) heads[carry + ex] = (uint32_t)i;
        carry += tot;
    }
    const int R = (int)carry;
    if (tid == 0) heads[R] = (uint32_t)n;
    __syncthreads();
    // segments
    carry = 0;
    for (int base = 0; base < R; base += kEncThreads) {
        const int r = base + tid;
        bool head = false, rle = false;
        if (r < R) {
            rle = heads[r + 1] - heads[r] >= 8u && r < R - 1;
            const bool prev_rle = r > 0 && heads[r] - heads[r - 1] >= 8u;
            head = r == 0 || rle || prev_rle;
        }
        uint32_t tot;
        const uint32_t ex = wg_excl_scan(head ? 1u : 0u, s_w, &tot);
        if (r < R) seg_of[r] = (uint16_t)(carry + ex + (head ? 1u : 0u) - 1u);
        if (head) segs[carry + ex] = heads[r] | (rle ? 0x80000000u : 0u);
        carry += tot;
    }
    const int NS = (int)carry;
    if (tid == 0) segs[NS] = (uint32_t)n;
    __syncthreads();
    // sizes and positions
    carry = 0;
    for (int base = 0; base < NS; base += kEncThreads) {
        const int s = base + tid;
        uint32_t size = 0;
        if (s < NS) {
            const uint32_t a = segs[s], c = (segs[s + 1] & 0x7fffffffu) - (a & 0x7fffffffu);
            size = (a >> 31) ? (uint32_t)uvarint_len((uint64_t)c << 1) + (uint32_t)bytew
                             : (uint32_t)uvarint_len(((uint64_t)c << 1) | 1) + ((c + 7) >> 3) * (uint32_t)bw;
        }
        uint32_t tot;
        const uint32_t ex = wg_excl_scan(size, s_w, &tot);
        if (s < NS) pos[s] = carry + ex;
        carry += tot;
    }
    const uint32_t total = carry;
    if (4ull + total > (uint64_t)cap) return 0;
    if (tid == 0) put_u32(dst, total);
    __syncthreads();
    uint8_t* body = dst + 4;
    // headers, run values
    for (int s = tid; s < NS; s += kEncThreads) {
        const uint32_t a = segs[s], st = a & 0x7fffffffu, c = (segs[s + 1] & 0x7fffffffu) - st;
        uint8_t* p = body + pos[s];
        if (a >> 31) {
            const int h = put_uvarint(p, (uint64_t)c << 1);
            const uint32_t val = ids[st];
            for (int k = 0; k < bytew; k++) p[h + k] = (uint8_t)(val >> (8 * k));
        } else {
            put_uvarint(p, ((uint64_t)c << 1) | 1);
        }
    }
    // bit-packed groups of 8 ids
    if (bw > 0) {
        for (int i = tid; i < n; i += kEncThreads) {
            const int s = seg_of[run_of[i]];
            const uint32_t a = segs[s], st = a & 0x7fffffffu;
            if ((a >> 31) || (((uint32_t)i - st) & 7u)) continue;
            const uint32_t en = segs[s + 1] & 0x7fffffffu, c = en - st;
            const uint32_t h = (uint32_t)uvarint_len(((uint64_t)c << 1) | 1);
            pack_group<uint16_t>(body + pos[s] + h + (((uint32_t)i - st) >> 3) * (uint32_t)bw, ids + i, (int)min(8u, en - (uint32_t)i), bw);
        }
    }
    return 4u + total;
}

// ---------------------------------------------------------------------------------------------------------------
// k_enc_pack
// ---------------------------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(kEncThreads) void k_enc_pack(const EncPack* __restrict__ packs, const int* __restrict__ list, EncMeta* __restrict__ metas,
                                                          const uint8_t* __restrict__ data, uint8_t* __restrict__ out, uint8_t* __restrict__ scratch) {
    __shared__ uint32_t s_w[kEncWaves + 1];
    __shared__ DeltaLds<U> s_delta;
    const int pi = list[blockIdx.x];
    const EncPack P = packs[pi];
    EncMeta& M = metas[pi];
    if (M.pack_size < 0) return;  // an earlier kernel reported an error
    const int n = P.obj_count, tid = threadIdx.x;
    uint8_t* dst = out + P.dst_off;
    const uint32_t cap = (uint32_t)P.dst_cap;
    if (!M.use_dict) {
        if (P.is_indexed) { if (tid == 0) M.pack_size = kEncImproper; return; }  // no DELTA fallback on the indexed path (:30-33)
        const uint32_t sz = wg_delta_encode<U>((const U*)(data + P.src_off), n, dst, cap, s_delta, s_w);
        if (tid == 0) { M.pack_size = sz ? (int32_t)sz : kEncOverflow; M.encode_type = IXR_ENC_DELTA; }
        return;
    }
    const int ec = M.entry_count;
    const uint32_t dict_size = (uint32_t)ec * (uint32_t)sizeof(U);
    const bool compressed = ec >= 64 && dict_size >= 256;  // DictStructUtil.java:26-27
    const U* ent = (const U*)(scratch + P.s_entries);
    if (9ull + dict_size > (uint64_t)cap) { if (tid == 0) M.pack_size = kEncOverflow; return; }
    if (tid == 0) {
        put_u32(dst, 4u + dict_size + 4u + 4u * (uint32_t)n);  // size of the uncompressed DictStruct
        put_u32(dst + 4, (uint32_t)ec);
        dst[8] = compressed ? 1 : 0;
    }
    uint32_t esz;
    if (compressed) {
        esz = wg_delta_encode<U>(ent, ec, dst + 9, cap - 9, s_delta, s_w);
        if (!esz) { if (tid == 0) M.pack_size = kEncOverflow; return; }
    } else {
        const uint8_t* eb = (const uint8_t*)ent;
        for (uint32_t i = tid; i < dict_size; i += kEncThreads) dst[9 + i] = eb[i];
        esz = dict_size;
    }
    const int bw = bit_width64((uint64_t)(uint32_t)(ec - 1));
    const uint32_t rsz = wg_rle_encode((const uint16_t*)(scratch + P.s_ids), n, bw, dst + 9 + esz, cap - 9 - esz,
                                       (uint16_t*)(scratch + P.s_run_of), (uint32_t*)(scratch + P.s_heads), (uint16_t*)(scratch + P.s_seg_of),
                                       (uint32_t*)(scratch + P.s_segs), (uint32_t*)(scratch + P.s_pos), s_w);
    if (tid == 0) { M.pack_size = rsz ? (int32_t)(9u + esz + rsz) : kEncOverflow; M.encode_type = IXR_ENC_DICT_COMPRESS; }
}

// ---------------------------------------------------------------------------------------------------------------
// k_enc_str: a string pack (StringsStruct: int32 offset[n+1] | bytes) -> DICT_COMPRESS, one workgroup per pack.
//   DictCodec.encodeStrings :379-433 (sorted set of the distinct strings, BytesUtil.compareBytes order: unsigned bytes,
//   then length), DictStructUtil.compress :25-81 (entries plain as a StringsStruct, or front coded by
//   DeltaCodec.encodeStrings :268-352 when there are >= 64 entries in >= 256 bytes), ids as RLE / bit-packing hybrid.
// Steps: the offsets are checked; a hash table finds one representative row per distinct string; a bitonic sort orders the
// representatives by their strings (8-byte prefix keys first, the strings on ties; compare-exchange pairs spread over the
// workgroup, one barrier per stage): slot e is
// dictionary entry e, and a row's id is the slot of its representative; entry lengths, common prefixes with the entry
// before and their running sums give every byte its place; the writers are those of the numeric packs.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSortPad = 0xFFFFFFFFu;  // a slot past the last row: sorts after every string
__device__ __forceinline__ int bytes_cmp(const uint8_t* a, int la, const uint8_t* b, int lb) {  // BytesUtil.compareBytes :7-19
    const int l = min(la, lb);
    int i = 0;
    for (; i + 8 <= l; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8);
        __builtin_memcpy(&y, b + i, 8);
        if (x != y) {
            const int sh = (__ffsll((long long)(x ^ y)) - 1) & ~7;  // the first byte that differs is the lowest one in a little-endian word
            return (int)((x >> sh) & 0xff) - (int)((y >> sh) & 0xff);
        }
    }
    for (; i < l; i++) {
        const int c = (int)a[i] - (int)b[i];
        if (c) return c;
    }
    return la - lb;
}
struct StrRows {
    const int32_t* off;
    const uint8_t* body;
    __device__ __forceinline__ int cmp(uint32_t ra, uint32_t rb) const {
        const int32_t a0 = off[ra], a1 = off[ra + 1], b0 = off[rb], b1 = off[rb + 1];
        return bytes_cmp(body + a0, a1 - a0, body + b0, b1 - b0);
    }
    // the sort's total order: strings, equal strings by row, padding last
    __device__ __forceinline__ bool after(uint32_t ra, uint32_t rb) const {
        if (ra == rb || rb == kSortPad) return false;
        if (ra == kSortPad) return true;
        const int c = cmp(ra, rb);
        return c > 0 || (c == 0 && ra > rb);
    }
};

__global__ __launch_bounds__(kEncThreads) void k_enc_str(const EncPack* __restrict__ packs, const int* __restrict__ list, EncMeta* __restrict__ metas,
                                                         const uint8_t* __restrict__ data, uint8_t* __restrict__ out, uint8_t* __restrict__ scratch) {
    __shared__ uint32_t s_w[kEncWaves + 1];
    __shared__ DeltaLds<uint32_t> s_delta;
    __shared__ int s_bad;
    const int pi = list[blockIdx.x];
    const EncPack P = packs[pi];
    EncMeta& M = metas[pi];
    const int n = P.obj_count, tid = threadIdx.x;
    const int32_t* off = (const int32_t*)(data + P.src_off);
    const StrRows S{off, data + P.src_off + 4ull * ((uint32_t)n + 1)};
    const int64_t total = (int64_t)P.data_size - 4ll * (n + 1);
    uint8_t* dst = out + P.dst_off;
    const uint32_t cap = (uint32_t)P.dst_cap;
    // the offsets index the body: ascending, inside it, the last one its size (what the host encoder asks of its input)
    if (tid == 0) s_bad = 0;
    __syncthreads();
    {
        bool bad = false;
        for (int i = tid; i < n; i += kEncThreads) {
            const int32_t a = off[i], b = off[i + 1];
            if (a < 0 || b < a || (int64_t)b > total) bad = true;
        }
        if (tid == 0 && (int64_t)off[n] != total) bad = true;
        if (bad) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) M.pack_size = IXR_ERR_INVALID_ARGUMENT; return; }
#ifdef IXR_ENC_TIMING
    long long tk[6]; tk[0] = wall_clock64();
#define TICK(k) tk[k] = wall_clock64()
#else
#define TICK(k)
#endif
    // distinct strings: every row goes into an open-addressing table keyed by a hash of its bytes; the first row to claim
    // a slot represents its string, later rows with the same bytes take it as their representative
    uint32_t n2 = 2;
    while (n2 < (uint32_t)n) n2 <<= 1;
    uint32_t* table = (uint32_t*)(scratch + P.s_table);
    uint32_t* rep = (uint32_t*)(scratch + P.s_rep);
    const uint32_t tmask = 2 * n2 - 1;
    for (uint32_t i = tid; i < 2 * n2; i += kEncThreads) table[i] = kSortPad;
    __syncthreads();
    for (int i = tid; i < n; i += kEncThreads) {
        const int32_t a0 = off[i], len = off[i + 1] - a0;
        const uint8_t* sp = S.body + a0;
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)len;
        int k = 0;
        for (; k + 8 <= len; k += 8) {
            uint64_t w;
            __builtin_memcpy(&w, sp + k, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
        uint64_t w = 0;
        for (int q = 0; k + q < len; q++) w |= (uint64_t)sp[k + q] << (8 * q);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 32;
        uint32_t slot = (uint32_t)h & tmask;
        for (;;) {
            uint32_t cur = atomicCAS(&table[slot], kSortPad, (uint32_t)i);
            if (cur == kSortPad) cur = (uint32_t)i;
            if (cur == (uint32_t)i || S.cmp(cur, (uint32_t)i) == 0) { rep[i] = cur; break; }
            slot = (slot + 1) & tmask;  // the table has at least twice as many slots as rows: a free one comes up
        }
    }
    __syncthreads();
    TICK(1);
    // the representatives, compacted in row order, are what gets sorted.  Each carries the first 8 bytes of its string as a
    // big-endian number (zero padded): where two such keys differ they decide the order, and the strings themselves are
    // read only on ties (equal strings never meet: every string is here once).  Keys of 16 bytes were not faster.
    uint32_t* perm = (uint32_t*)(scratch + P.s_perm);
    uint64_t* keys = (uint64_t*)(scratch + P.s_key);
    uint32_t carry = 0;
    for (int base = 0; base < n; base += kEncThreads) {
        const int i = base + tid;
        const bool mine = i < n && rep[i] == (uint32_t)i;
        uint32_t tot;
        const uint32_t ex = wg_excl_scan(mine ? 1u : 0u, s_w, &tot);
        if (mine) {
            const int32_t a0 = off[i], len = off[i + 1] - a0;
            const uint8_t* sp = S.body + a0;
            uint64_t key = 0;
            for (int q = 0; q < 8; q++) key = (key << 8) | (q < len ? (uint64_t)sp[q] : 0ull);
            perm[carry + ex] = (uint32_t)i;
            keys[carry + ex] = key;
        }
        carry += tot;
    }
    const int ec = (int)carry;
    uint32_t m2 = 2;
    while (m2 < (uint32_t)ec) m2 <<= 1;
    for (uint32_t i = ec + tid; i < m2; i += kEncThreads) { perm[i] = kSortPad; keys[i] = ~0ull; }
    __syncthreads();
    for (uint32_t k = 2; k <= m2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < m2 / 2; t += kEncThreads) {
                const uint32_t i = 2 * j * (t / j) + (t % j), l = i + j;
                const uint64_t ka = keys[i], kb = keys[l];
                const bool up = (i & k) == 0;
                bool after;  // does slot i's string sort after slot l's?
                uint32_t a = 0, b = 0;
                if (ka != kb) after = ka > kb;
                else { a = perm[i]; b = perm[l]; after = S.after(a, b); }
                if (after == up) {
                    if (ka != kb) { a = perm[i]; b = perm[l]; }
                    if (a != b) { perm[i] = b; perm[l] = a; keys[i] = kb; keys[l] = ka; }
                }
            }
            __syncthreads();
        }
    }
    TICK(2);
    // slot e of the sorted representatives is dictionary entry e; a row's id is the entry of its representative
    uint16_t* ids = (uint16_t*)(scratch + P.s_ids);
    uint16_t* rank = (uint16_t*)(scratch + P.s_rank);
    const uint32_t* ent_row = perm;
    for (int e = tid; e < ec; e += kEncThreads) rank[perm[e]] = (uint16_t)e;
    __syncthreads();
    for (int i = tid; i < n; i += kEncThreads) ids[i] = rank[rep[i]];
    __syncthreads();
    // entry lengths, prefixes, running sums
    uint32_t* pre = (uint32_t*)(scratch + P.s_pre);
    uint32_t* suf = (uint32_t*)(scratch + P.s_suf);
    uint32_t* ent_off = (uint32_t*)(scratch + P.s_ent_off);
    uint32_t* suf_off = (uint32_t*)(scratch + P.s_suf_off);
    uint32_t len_carry = 0, suf_carry = 0;
    for (int base = 0; base < ec; base += kEncThreads) {
        const int e = base + tid;
        uint32_t len = 0, p = 0;
        if (e < ec) {
            const uint32_t r = ent_row[e];
            const int32_t a0 = off[r];
            len = (uint32_t)(off[r + 1] - a0);
            if (e > 0) {
                const uint32_t q = ent_row[e - 1];
                const int32_t b0 = off[q];
                const uint32_t lim = min(len, (uint32_t)(off[q + 1] - b0));
                while (p < lim && S.body[a0 + p] == S.body[b0 + p]) p++;
            }
            pre[e] = p;
            suf[e] = len - p;
        }
        uint32_t tot;
        uint32_t ex = wg_excl_scan(len, s_w, &tot);
        if (e < ec) ent_off[e] = len_carry + ex;
        len_carry += tot;
        ex = wg_excl_scan(len - p, s_w, &tot);
        if (e < ec) suf_off[e] = suf_carry + ex;
        suf_carry += tot;
    }
    __syncthreads();
    TICK(3);
    const uint32_t body_bytes = len_carry, suf_bytes = suf_carry;
    const uint64_t dict_size = 4ull * ((uint32_t)ec + 1) + body_bytes;
    const bool compressed = ec >= 64 && dict_size >= 256;  // DictStructUtil.java:26-27
    if (tid == 0) {
        put_u32(dst, (uint32_t)(4u + dict_size + 4u + 4ull * (uint32_t)n));  // size of the uncompressed DictStruct
        put_u32(dst + 4, (uint32_t)ec);
        dst[8] = compressed ? 1 : 0;
    }
    uint32_t esz;
    if (!compressed) {
        if (9ull + dict_size > (uint64_t)cap) { if (tid == 0) M.pack_size = kEncOverflow; return; }
        uint8_t* idx = dst + 9;
        uint8_t* bytes = idx + 4ull * ((uint32_t)ec + 1);
        for (int e = tid; e <= ec; e += kEncThreads) put_u32(idx + 4ull * e, e < ec ? ent_off[e] : body_bytes);
        for (int e = tid; e < ec; e += kEncThreads) {
            const uint32_t r = ent_row[e];
            const uint8_t* sp = S.body + off[r];
            const uint32_t len = (uint32_t)(off[r + 1] - off[r]), at = ent_off[e];
            for (uint32_t k = 0; k < len; k++) bytes[at + k] = sp[k];
        }
        esz = (uint32_t)dict_size;
    } else {
        uint8_t* base = dst + 9;
        const uint32_t d1 = wg_delta_encode<uint32_t>(pre, ec, base + 4, cap - 13, s_delta, s_w);
        if (!d1) { if (tid == 0) M.pack_size = kEncOverflow; return; }
        __syncthreads();
        const uint32_t d2 = wg_delta_encode<uint32_t>(suf, ec, base + 4 + d1, cap - 13 - d1, s_delta, s_w);
        if (!d2 || 13ull + d1 + d2 + suf_bytes > (uint64_t)cap) { if (tid == 0) M.pack_size = kEncOverflow; return; }
        uint8_t* bytes = base + 4 + d1 + d2;
        for (int e = tid; e < ec; e += kEncThreads) {
            const uint32_t r = ent_row[e], p = pre[e], len = suf[e], at = suf_off[e];
            const uint8_t* sp = S.body + off[r] + p;
            for (uint32_t k = 0; k < len; k++) bytes[at + k] = sp[k];
        }
        if (tid == 0) put_u32(base, d1 + d2 + suf_bytes);
        esz = 4u + d1 + d2 + suf_bytes;
    }
    __syncthreads();
    TICK(4);
    const int bw = bit_width64((uint64_t)(uint32_t)(ec - 1));
    const uint32_t rsz = wg_rle_encode(ids, n, bw, dst + 9 + esz, cap - 9 - esz, (uint16_t*)(scratch + P.s_run_of), (uint32_t*)(scratch + P.s_heads),
                                       (uint16_t*)(scratch + P.s_seg_of), (uint32_t*)(scratch + P.s_segs), (uint32_t*)(scratch + P.s_pos), s_w);
#ifdef IXR_ENC_TIMING
    TICK(5);
    if (tid == 0) { M.min_value = ((tk[1] - tk[0]) << 32) | (tk[2] - tk[1]); M.max_value = ((tk[3] - tk[2]) << 32) | (tk[4] - tk[3]); M.sum = tk[5] - tk[4]; }
#endif
    if (tid == 0) { M.entry_count = ec; M.use_dict = 1; M.pack_size = rsz ? (int32_t)(9u + esz + rsz) : kEncOverflow; M.encode_type = IXR_ENC_DICT_COMPRESS; }
}

}  // namespace ixr

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
using namespace ixr;

namespace {
int max_entries(int n, float ratio) {
    volatile float prod = (float)n * ratio;  // int * float in float precision, then ceil in double (DictCodec.java:83)
    return (int)std::ceil((double)prod);
}
inline uint64_t align256(uint64_t v) { return (v + 255) & ~255ull; }
struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
};
// Lays the per-pack working arrays of `p` out from byte `off` of the group's scratch; returns the first byte after them.
uint64_t layout_pack(EncPack& p, uint64_t off) {
    const uint64_t nn = (uint64_t)p.obj_count;
    auto take = [&](uint64_t bytes) { const uint64_t at = off; off += align256(bytes); return at; };
    p.s_ids = take(nn * 2);
    p.s_run_of = take(nn * 2);
    p.s_seg_of = take(nn * 2);
    p.s_heads = take((nn + 1) * 4);
    p.s_segs = take((nn + 1) * 4);
    p.s_pos = take((nn + 1) * 4);
    p.s_entries = p.s_table = p.s_rep = p.s_rank = p.s_perm = p.s_key = p.s_pre = p.s_suf = p.s_ent_off = p.s_suf_off = 0;
    if (p.data_type == IXR_STRING) {
        uint64_t n2 = 2;
        while (n2 < nn) n2 <<= 1;
        p.s_table = take(n2 * 8);
        p.s_rep = take(nn * 4);
        p.s_rank = take(nn * 2);
        p.s_perm = take(n2 * 4);
        p.s_key = take(n2 * 8);
        p.s_pre = take(nn * 4);
        p.s_suf = take(nn * 4);
        p.s_ent_off = take((nn + 1) * 4);
        p.s_suf_off = take((nn + 1) * 4);
    } else {
        p.s_entries = take(nn * ((p.data_type == IXR_INT || p.data_type == IXR_FLOAT) ? 4 : 8));
    }
    return off;
}
template <typename U>
hipError_t sort_segments(void* temp, size_t& temp_bytes, U* in, U* out, unsigned total, unsigned segments, const unsigned* begins, const unsigned* ends, hipStream_t st) {
    return rocprim::segmented_radix_sort_keys(temp, temp_bytes, in, out, total, segments, begins, ends, 0, (unsigned)(sizeof(U) * 8), st);
}
}  // namespace

extern "C" int32_t ixr_encode_run(int32_t device, ixr_encode_desc* descs, int32_t n, const void* d_data, uint64_t data_bytes,
                                  void* d_out, uint64_t out_bytes, uint64_t scratch_budget, void* stream) {
    if (n < 0 || (n > 0 && (!descs || !d_data || !d_out))) return set_error(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return 0;
    if (((uintptr_t)d_data & 15) || ((uintptr_t)d_out & 15)) return set_error(IXR_ERR_INVALID_ARGUMENT, "misaligned device pointer");
    int32_t rc = ixr_init(device);
    if (rc) return rc;
    // what the kernels index with is checked here: rows inside the data arena, output slots inside the output arena
    std::vector<EncPack> hp((size_t)n);
    std::vector<EncMeta> hm((size_t)n);
    uint64_t per_pack_max = 0;
    std::vector<uint64_t> need((size_t)n);
    for (int i = 0; i < n; i++) {
        ixr_encode_desc& d = descs[i];
        d.pack_size = IXR_ERR_BUG;
        d.encode_type = 0;
        if (d.data_type < IXR_INT || d.data_type > IXR_STRING) return set_error(IXR_ERR_UNSUPPORTED_TYPE, "encode descriptor: unknown data type");
        const bool str = d.data_type == IXR_STRING;
        const int w = (d.data_type == IXR_INT || d.data_type == IXR_FLOAT) ? 4 : 8;
        if (d.obj_count < 1 || d.obj_count > IXR_PACK_MAX_ROWS || d.data_size < 0 ||
            (str ? (int64_t)d.data_size < 4 * ((int64_t)d.obj_count + 1) : (int64_t)d.obj_count * w != (int64_t)d.data_size))
            return set_error(IXR_ERR_INVALID_ARGUMENT, "encode descriptor: row count and data size do not agree");
        if ((d.src_off & 15) || d.src_off + (uint64_t)d.data_size > data_bytes) return set_error(IXR_ERR_INVALID_ARGUMENT, "encode descriptor: rows outside the data arena");
        const int32_t cap = ixr_compress_bound(d.obj_count, d.data_size);
        if ((d.dst_off & 15) || d.dst_off + (uint64_t)cap > out_bytes) return set_error(IXR_ERR_INVALID_ARGUMENT, "encode descriptor: output slot outside the output arena");
        EncPack& p = hp[(size_t)i];
        p.src_off = d.src_off; p.dst_off = d.dst_off; p.data_type = d.data_type; p.is_indexed = d.is_indexed ? 1 : 0;
        p.obj_count = d.obj_count; p.dst_cap = cap; p.data_size = d.data_size;
        p.max_entries = max_entries(d.obj_count, (d.data_type == IXR_INT && !d.is_indexed) ? 0.2f : 1.0f);  // PackCompressor_VLT.java:29-96
        need[(size_t)i] = layout_pack(p, 0) + (str ? 0 : 2 * align256((uint64_t)d.obj_count * w)) /* keys in, keys sorted */;
        per_pack_max = std::max(per_pack_max, need[(size_t)i]);
        hm[(size_t)i] = EncMeta{INT32_MAX, 0, 0, 0, 0, {0, 0, 0}, 0, 0, 0};
    }
    if (scratch_budget == 0) scratch_budget = 4ull << 30;
    if (scratch_budget < per_pack_max) return set_error(IXR_ERR_INVALID_ARGUMENT, "scratch budget smaller than one pack needs");
    if (hipSetDevice(device) != hipSuccess) return set_error(IXR_ERR_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    // groups of packs whose scratch fits the budget; a group is also the y extent of the per-pack grids, which ends at 65535
    constexpr int kMaxGroupPacks = 32768;
    struct Group { int first, count; uint64_t bytes; };
    std::vector<Group> groups;
    uint64_t max_group = 0;
    for (int i = 0; i < n;) {
        uint64_t bytes = 0;
        int j = i;
        while (j < n && j - i < kMaxGroupPacks && bytes + need[(size_t)j] <= scratch_budget) bytes += need[(size_t)j++];
        groups.push_back({i, j - i, bytes});
        max_group = std::max(max_group, bytes);
        i = j;
    }
    DevMem d_packs, d_metas, d_scratch, d_list, d_seg, d_temp;
    if (!d_packs.alloc(sizeof(EncPack) * (size_t)n) || !d_metas.alloc(sizeof(EncMeta) * (size_t)n) || !d_scratch.alloc(max_group + 256) ||
        !d_list.alloc(sizeof(int) * (size_t)n * 3) || !d_seg.alloc(sizeof(unsigned) * (size_t)n * 4))
        return set_error(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    // scratch layout per group: [keys32 in][keys32 sorted][keys64 in][keys64 sorted][per pack arrays]
    std::vector<int> lists((size_t)n * 3);
    std::vector<unsigned> segs((size_t)n * 4);
    struct GroupPlan { int c32, c64, cstr; uint64_t k32_in, k32_out, k64_in, k64_out; unsigned t32, t64; };
    std::vector<GroupPlan> plans;
    size_t temp_bytes = 0;
    for (const Group& g : groups) {
        GroupPlan gp{};
        uint64_t e32 = 0, e64 = 0;
        int* l32 = lists.data() + g.first;
        int* l64 = lists.data() + (size_t)n + g.first;
        int* lstr = lists.data() + 2 * (size_t)n + g.first;
        unsigned* b32 = segs.data() + g.first;
        unsigned* e32p = segs.data() + (size_t)n + g.first;
        unsigned* b64 = segs.data() + 2 * (size_t)n + g.first;
        unsigned* e64p = segs.data() + 3 * (size_t)n + g.first;
        for (int i = g.first; i < g.first + g.count; i++) {
            EncPack& p = hp[(size_t)i];
            if (p.data_type == IXR_STRING) {
                p.key_off = 0; lstr[gp.cstr++] = i;
            } else if (p.data_type == IXR_INT || p.data_type == IXR_FLOAT) {
                p.key_off = e32; b32[gp.c32] = (unsigned)e32; e32 += (uint64_t)p.obj_count; e32p[gp.c32] = (unsigned)e32; l32[gp.c32++] = i;
            } else {
                p.key_off = e64; b64[gp.c64] = (unsigned)e64; e64 += (uint64_t)p.obj_count; e64p[gp.c64] = (unsigned)e64; l64[gp.c64++] = i;
            }
        }
        gp.t32 = (unsigned)e32; gp.t64 = (unsigned)e64;
        uint64_t off = 0;
        gp.k32_in = off; off += align256(e32 * 4);
        gp.k32_out = off; off += align256(e32 * 4);
        gp.k64_in = off; off += align256(e64 * 8);
        gp.k64_out = off; off += align256(e64 * 8);
        for (int i = g.first; i < g.first + g.count; i++) off = layout_pack(hp[(size_t)i], off);
        // the key areas round up once each where `need` rounded up once per pack: the layout never exceeds the group's sum
        if (off > max_group) return set_error(IXR_ERR_BUG, "encode scratch layout exceeds its allocation");
        // temporary storage of the sort: the largest any group asks for
        size_t t = 0;
        if (gp.c32 && sort_segments<uint32_t>(nullptr, t, (uint32_t*)nullptr, (uint32_t*)nullptr, gp.t32, (unsigned)gp.c32, (const unsigned*)nullptr, (const unsigned*)nullptr, st) != hipSuccess)
            return set_error(IXR_ERR_HIP, "sort size query failed");
        temp_bytes = std::max(temp_bytes, t);
        t = 0;
        if (gp.c64 && sort_segments<uint64_t>(nullptr, t, (uint64_t*)nullptr, (uint64_t*)nullptr, gp.t64, (unsigned)gp.c64, (const unsigned*)nullptr, (const unsigned*)nullptr, st) != hipSuccess)
            return set_error(IXR_ERR_HIP, "sort size query failed");
        temp_bytes = std::max(temp_bytes, t);
        plans.push_back(gp);
    }
    if (!d_temp.alloc(temp_bytes + 256)) return set_error(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (hipMemcpyAsync(d_packs.p, hp.data(), sizeof(EncPack) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_metas.p, hm.data(), sizeof(EncMeta) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_list.p, lists.data(), sizeof(int) * (size_t)n * 3, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_seg.p, segs.data(), sizeof(unsigned) * (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)  // the host vectors are pageable memory
        return set_error(IXR_ERR_HIP, "H2D failed");
    const EncPack* P = (const EncPack*)d_packs.p;
    EncMeta* M = (EncMeta*)d_metas.p;
    uint8_t* S = (uint8_t*)d_scratch.p;
    const uint8_t* D = (const uint8_t*)d_data;
    uint8_t* O = (uint8_t*)d_out;
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const Group& g = groups[gi];
        const GroupPlan& gp = plans[gi];
        const int* l32 = (const int*)d_list.p + g.first;
        const int* l64 = (const int*)d_list.p + (size_t)n + g.first;
        const unsigned* sg = (const unsigned*)d_seg.p;
        if (gp.c32) {
            uint32_t* kin = (uint32_t*)(S + gp.k32_in);
            uint32_t* kout = (uint32_t*)(S + gp.k32_out);
            hipLaunchKernelGGL(k_enc_stats<uint32_t>, dim3(gp.c32), dim3(kStatThreads), 0, st, P, l32, M, D);
            hipLaunchKernelGGL(k_enc_keys<uint32_t>, dim3(IXR_PACK_TILES, gp.c32), dim3(1024), 0, st, P, l32, M, D, kin);
            size_t t = temp_bytes;
            if (sort_segments<uint32_t>(d_temp.p, t, kin, kout, gp.t32, (unsigned)gp.c32, sg + g.first, sg + (size_t)n + g.first, st) != hipSuccess)
                return set_error(IXR_ERR_HIP, "segmented sort failed");
            hipLaunchKernelGGL(k_enc_dict<uint32_t>, dim3(gp.c32), dim3(kEncThreads), 0, st, P, l32, M, D, (const uint32_t*)kout, S);
            hipLaunchKernelGGL(k_enc_pack<uint32_t>, dim3(gp.c32), dim3(kEncThreads), 0, st, P, l32, M, D, O, S);
        }
        if (gp.c64) {
            uint64_t* kin = (uint64_t*)(S + gp.k64_in);
            uint64_t* kout = (uint64_t*)(S + gp.k64_out);
            hipLaunchKernelGGL(k_enc_stats<uint64_t>, dim3(gp.c64), dim3(kStatThreads), 0, st, P, l64, M, D);
            hipLaunchKernelGGL(k_enc_keys<uint64_t>, dim3(IXR_PACK_TILES, gp.c64), dim3(1024), 0, st, P, l64, M, D, kin);
            size_t t = temp_bytes;
            if (sort_segments<uint64_t>(d_temp.p, t, kin, kout, gp.t64, (unsigned)gp.c64, sg + 2 * (size_t)n + g.first, sg + 3 * (size_t)n + g.first, st) != hipSuccess)
                return set_error(IXR_ERR_HIP, "segmented sort failed");
            hipLaunchKernelGGL(k_enc_dict<uint64_t>, dim3(gp.c64), dim3(kEncThreads), 0, st, P, l64, M, D, (const uint64_t*)kout, S);
            hipLaunchKernelGGL(k_enc_pack<uint64_t>, dim3(gp.c64), dim3(kEncThreads), 0, st, P, l64, M, D, O, S);
        }
        if (gp.cstr) {
            const int* lstr = (const int*)d_list.p + 2 * (size_t)n + g.first;
            hipLaunchKernelGGL(k_enc_str, dim3(gp.cstr), dim3(kEncThreads), 0, st, P, lstr, M, D, O, S);
        }
        if (hipGetLastError() != hipSuccess) return set_error(IXR_ERR_HIP, "kernel launch failed");
    }
    if (hipMemcpyAsync(hm.data(), d_metas.p, sizeof(EncMeta) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return set_error(IXR_ERR_HIP, "D2H failed");
    int bad = 0;
    for (int i = 0; i < n; i++) {
        descs[i].pack_size = hm[(size_t)i].pack_size;
        descs[i].encode_type = hm[(size_t)i].encode_type;
        descs[i].min_value = hm[(size_t)i].min_value;
        descs[i].max_value = hm[(size_t)i].max_value;
        descs[i].sum = hm[(size_t)i].sum;
        bad += descs[i].pack_size < 0;
    }
    return bad;
}
