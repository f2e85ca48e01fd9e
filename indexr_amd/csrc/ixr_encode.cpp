// ixr_encode.cpp -- host-side (CPU) VLT pack encoder: the write path that produces the bytes the GPU decoder reads.
//
// Mirrors the reference's codec choice and wire format
//   PackCompressor_VLT.compress           indexr-vlt-segment/.../segment/pack/PackCompressor_VLT.java:23-128
//   DictCompressCodec.encode              .../codec/dict/DictCompressCodec.java:40-69
//   DictCodec.encode* (sorted dictionary) .../codec/dict/DictCodec.java:79-131, 154-206, 229-281, 304-356, 379-433
//   DictStructUtil.compress               .../codec/dict/DictStructUtil.java:25-81
//   DeltaCodec.encode*                    .../codec/delta/DeltaCodec.java:40-97, 144-201, 268-352
//   RLEPackingHybridCodec.encodeInts      .../codec/rle/RLEPackingHybridCodec.java:38-94
//   bit packing, varints                  .../codec/pack/IntPacker.java:35-52, .../codec/VarInteger.java:25-118
// It is an independent implementation (word-wise bit accumulator, std::sort based dictionaries); the test-suite checks
// its output byte for byte against the CPU oracle's encoder.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ixr_gpu.h"

namespace {

struct Sink {
    std::vector<uint8_t> buf;
    void u8(uint32_t b) { buf.push_back((uint8_t)b); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) buf.push_back((uint8_t)(v >> (8 * i))); }
    void bytes(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; buf.insert(buf.end(), b, b + n); }
    void uvarint(uint64_t v) {
        while (v >= 0x80) { buf.push_back((uint8_t)(v | 0x80)); v >>= 7; }
        buf.push_back((uint8_t)v);
    }
    size_t reserve_u32() { size_t at = buf.size(); u32(0); return at; }
    void patch_u32(size_t at, uint32_t v) { for (int i = 0; i < 4; i++) buf[at + i] = (uint8_t)(v >> (8 * i)); }
};

inline int bit_width(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }

// n values of `bw` bits each, value i at bit i*bw; the run is padded with zero bits to ceil(n/8)*bw bytes.
void pack_bits(Sink& s, const uint64_t* v, int n, int bw) {
    if (n == 0 || bw == 0) return;
    const size_t total = (size_t)((n + 7) >> 3) * (size_t)bw;
    const size_t at = s.buf.size();
    s.buf.resize(at + total, 0);
    uint8_t* out = s.buf.data() + at;
    const uint64_t mask = bw == 64 ? ~0ull : ((1ull << bw) - 1);
    unsigned __int128 acc = 0;
    int have = 0;
    size_t o = 0;
    for (int i = 0; i < n; i++) {
        acc |= (unsigned __int128)(v[i] & mask) << have;
        have += bw;
        while (have >= 8) { out[o++] = (uint8_t)acc; acc >>= 8; have -= 8; }
    }
    if (have > 0) out[o++] = (uint8_t)acc;
    // remaining bytes (padding of the last group of 8) are already zero
}

template <typename S> struct Wide;
template <> struct Wide<int32_t> { using U = uint32_t; static constexpr int bits = 32; };
template <> struct Wide<int64_t> { using U = uint64_t; static constexpr int bits = 64; };

template <typename S> void put_zigzag(Sink& s, S v) {
    using U = typename Wide<S>::U;
    U z = ((U)v << 1) ^ (U)(v >> (Wide<S>::bits - 1));
    s.uvarint(z);
}

// DELTA: u32 len | zz first | per 128 values: zz minDelta, per 32 values: u8 bw + packed (delta - minDelta)
template <typename S> void delta_encode(Sink& s, const S* v, int n) {
    using U = typename Wide<S>::U;
    if (n == 0) return;
    const size_t len_at = s.reserve_u32();
    const size_t start = s.buf.size();
    put_zigzag<S>(s, v[0]);
    U prev = (U)v[0];
    uint64_t rel[128];
    S d[128];
    for (int b0 = 0; b0 < n; b0 += 128) {
        const int cnt = std::min(128, n - b0);
        S mn = 0;
        for (int i = 0; i < cnt; i++) {
            U cur = (U)v[b0 + i];
            d[i] = (S)(U)(cur - prev);
            prev = cur;
            if (i == 0 || d[i] < mn) mn = d[i];
        }
        put_zigzag<S>(s, mn);
        for (int i = 0; i < cnt; i++) rel[i] = (uint64_t)(U)((U)d[i] - (U)mn);
        for (int m0 = 0; m0 < cnt; m0 += 32) {
            const int c = std::min(32, cnt - m0);
            uint64_t orv = 0;
            for (int i = 0; i < c; i++) orv |= rel[m0 + i];
            const int bw = bit_width(orv);
            s.u8((uint32_t)bw);
            pack_bits(s, rel + m0, c, bw);
        }
    }
    s.patch_u32(len_at, (uint32_t)(s.buf.size() - start));
}

// RLE / bit-packing hybrid over non-negative ids. A run is written only when it is >= 8 long and a different value
// follows it; everything else, including whatever run ends the input, goes into bit-packed chunks.
void rle_encode(Sink& s, const uint32_t* ids, int n, int bw) {
    if (n == 0) return;
    const size_t len_at = s.reserve_u32();
    const size_t start = s.buf.size();
    const int bytew = (bw + 7) >> 3;
    std::vector<uint64_t> tmp;
    auto literal = [&](int a, int b) {
        if (a == b) return;
        s.uvarint(((uint64_t)(uint32_t)(b - a) << 1) | 1);
        tmp.resize((size_t)(b - a));
        for (int i = a; i < b; i++) tmp[(size_t)(i - a)] = ids[i];
        pack_bits(s, tmp.data(), b - a, bw);
    };
    int lit_start = 0, i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && ids[j] == ids[i]) j++;
        if (j - i >= 8 && j < n) {
            literal(lit_start, i);
            s.uvarint((uint64_t)(uint32_t)(j - i) << 1);
            for (int k = 0; k < bytew; k++) s.u8(ids[i] >> (8 * k));
            lit_start = j;
        }
        i = j;
    }
    literal(lit_start, n);
    s.patch_u32(len_at, (uint32_t)(s.buf.size() - start));
}

struct StrRef { const uint8_t* p; int len; };
inline int cmp_bytes(const StrRef& a, const StrRef& b) {  // unsigned bytes, then length (BytesUtil.compareBytes)
    int l = std::min(a.len, b.len);
    int c = l ? memcmp(a.p, b.p, (size_t)l) : 0;
    return c ? c : a.len - b.len;
}

// front coding: u32 len | DELTA(prefixLen) | DELTA(suffixLen) | suffix bytes
void delta_encode_strings(Sink& s, const StrRef* str, int n) {
    if (n == 0) return;
    const size_t len_at = s.reserve_u32();
    const size_t start = s.buf.size();
    std::vector<int32_t> pre((size_t)n), suf((size_t)n);
    StrRef prev{nullptr, 0};
    for (int i = 0; i < n; i++) {
        int l = std::min(prev.len, str[i].len), p = 0;
        while (p < l && prev.p[p] == str[i].p[p]) p++;
        pre[(size_t)i] = p;
        suf[(size_t)i] = str[i].len - p;
        prev = str[i];
    }
    delta_encode<int32_t>(s, pre.data(), n);
    delta_encode<int32_t>(s, suf.data(), n);
    for (int i = 0; i < n; i++) s.bytes(str[i].p + pre[(size_t)i], (size_t)suf[(size_t)i]);
    s.patch_u32(len_at, (uint32_t)(s.buf.size() - start));
}

// total order of java.lang.Float.compare / Double.compare as an unsigned key: -0.0 < 0.0, NaN (all alike) greatest
inline uint32_t fkey(uint32_t b) {
    if ((b & 0x7fffffffu) > 0x7f800000u) b = 0x7fc00000u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline uint64_t dkey(uint64_t b) {
    if ((b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) b = 0x7ff8000000000000ull;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

int max_entries(int n, float ratio) {
    volatile float prod = (float)n * ratio;  // int * float in float precision, then ceil in double (DictCodec.java:83)
    return (int)std::ceil((double)prod);
}

// The lookup the reference does with fastutil's binarySearch: plain < and > on the primitive type, first hit wins.
template <typename T> int plain_search(const T* e, int m, T key) {
    int lo = 0, hi = m - 1;
    while (lo <= hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (e[mid] < key) lo = mid + 1;
        else if (e[mid] > key) hi = mid - 1;
        else return mid;
    }
    return -1;
}

// Open-addressing set/map over 64-bit keys used to find the distinct values of a pack before sorting them.
struct KeyMap {
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;  // -1 = empty slot
    uint64_t mask;
    explicit KeyMap(size_t n) {
        size_t cap = 16;
        while (cap < 2 * n) cap <<= 1;
        keys.assign(cap, 0);
        vals.assign(cap, -1);
        mask = cap - 1;
    }
    static inline uint64_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; return k; }
    // returns the slot of `k`, inserting it with value `v` when absent; *fresh tells which
    inline size_t upsert(uint64_t k, int32_t v, bool* fresh) {
        size_t i = (size_t)(mix(k) & mask);
        for (;;) {
            if (vals[i] < 0) { keys[i] = k; vals[i] = v; *fresh = true; return i; }
            if (keys[i] == k) { *fresh = false; return i; }
            i = (i + 1) & mask;
        }
    }
    inline int32_t get(uint64_t k) const {
        size_t i = (size_t)(mix(k) & mask);
        for (;;) {
            if (vals[i] < 0) return -1;
            if (keys[i] == k) return vals[i];
            i = (i + 1) & mask;
        }
    }
    inline void set(uint64_t k, int32_t v) { bool f; vals[upsert(k, v, &f)] = v; }
};

inline uint64_t hash_bytes(const uint8_t* p, int len) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
    int i = 0;
    for (; i + 8 <= len; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
    uint64_t w = 0;
    if (i < len) memcpy(&w, p + i, (size_t)(len - i));
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 32);
}

struct Dict {
    int entry_count = 0;
    std::vector<uint8_t> entries;  // numbers: raw LE array ; strings: StringsStruct
    std::vector<uint32_t> ids;
};

// returns false when the dictionary would exceed ceil(n * ratio) entries (DATA_IMPROPER)
// Distinct values of a numeric column as order-preserving unsigned keys, ascending in the dictionary's order.
// key(x): int32/int64 -> offset binary; float/double -> fkey/dkey (Float.compare order, NaNs collapsed).
template <typename Bits, typename KeyFn>
static bool numeric_dict(int n, const uint8_t* data, int maxe, KeyFn key, bool is_fp, Dict& d) {
    KeyMap map((size_t)n);
    std::vector<uint64_t> uniq;
    uniq.reserve(1024);
    std::vector<Bits> first_bits;  // bit pattern of the first row that produced each distinct key (matters for NaN payloads)
    std::vector<uint64_t> rowkey((size_t)n);
    for (int i = 0; i < n; i++) {
        Bits b;
        memcpy(&b, data + (size_t)i * sizeof(Bits), sizeof(Bits));
        const uint64_t k = key(b);
        rowkey[(size_t)i] = k;
        bool fresh;
        map.upsert(k, (int32_t)uniq.size(), &fresh);
        if (fresh) { uniq.push_back(k); first_bits.push_back(b); }
    }
    if ((int)uniq.size() > maxe) return false;
    std::vector<int32_t> order(uniq.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int32_t)i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return uniq[(size_t)a] < uniq[(size_t)b]; });
    std::vector<Bits> entries(uniq.size());
    for (size_t r = 0; r < order.size(); r++) {
        entries[r] = first_bits[(size_t)order[r]];
        map.set(uniq[(size_t)order[r]], (int32_t)r);
    }
    d.entry_count = (int)entries.size();
    d.entries.assign((const uint8_t*)entries.data(), (const uint8_t*)entries.data() + entries.size() * sizeof(Bits));
    for (int i = 0; i < n; i++) d.ids[(size_t)i] = (uint32_t)map.get(rowkey[(size_t)i]);
    if (is_fp) {
        // The reference looks rows up with a plain </> binary search on the primitive (fastutil binarySearch): zeros of
        // either sign compare equal there and a NaN compares equal to whatever is probed first. Redo those rows that way.
        for (int i = 0; i < n; i++) {
            Bits b;
            memcpy(&b, data + (size_t)i * sizeof(Bits), sizeof(Bits));
            bool special;
            if constexpr (sizeof(Bits) == 4) special = (b & 0x7fffffffu) == 0 || (b & 0x7fffffffu) > 0x7f800000u;
            else special = (b & 0x7fffffffffffffffull) == 0 || (b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
            if (!special) continue;
            int id;
            if constexpr (sizeof(Bits) == 4) { float f; memcpy(&f, &b, 4); id = plain_search((const float*)entries.data(), (int)entries.size(), f); }
            else { double f; memcpy(&f, &b, 8); id = plain_search((const double*)entries.data(), (int)entries.size(), f); }
            if (id < 0) return false;
            d.ids[(size_t)i] = (uint32_t)id;
        }
    }
    return true;
}

// returns false when the dictionary would exceed ceil(n * ratio) entries (DATA_IMPROPER)
bool build_dict(int data_type, float ratio, int n, const uint8_t* data, Dict& d) {
    const int maxe = max_entries(n, ratio);
    d.ids.resize((size_t)n);
    switch (data_type) {
        case IXR_INT:
            return numeric_dict<uint32_t>(n, data, maxe, [](uint32_t b) { return (uint64_t)(b ^ 0x80000000u); }, false, d);
        case IXR_LONG:
            return numeric_dict<uint64_t>(n, data, maxe, [](uint64_t b) { return b ^ 0x8000000000000000ull; }, false, d);
        case IXR_FLOAT:
            return numeric_dict<uint32_t>(n, data, maxe, [](uint32_t b) { return (uint64_t)fkey(b); }, true, d);
        case IXR_DOUBLE:
            return numeric_dict<uint64_t>(n, data, maxe, [](uint64_t b) { return dkey(b); }, true, d);
        case IXR_STRING: {
            const uint8_t* body = data + 4 * ((size_t)n + 1);
            auto off = [&](int i) { int32_t o; memcpy(&o, data + 4 * (size_t)i, 4); return o; };
            // distinct strings through a hash of the bytes (collisions resolved by comparing), then sort only those
            KeyMap map((size_t)n);
            std::vector<StrRef> uniq;
            std::vector<int32_t> row_u((size_t)n);
            for (int i = 0; i < n; i++) {
                StrRef r{body + off(i), off(i + 1) - off(i)};
                uint64_t h = hash_bytes(r.p, r.len);
                for (;;) {  // linear re-hash on a (rare) collision of different strings
                    bool fresh;
                    size_t slot = map.upsert(h, (int32_t)uniq.size(), &fresh);
                    if (fresh) { uniq.push_back(r); row_u[(size_t)i] = (int32_t)uniq.size() - 1; break; }
                    const StrRef& e = uniq[(size_t)map.vals[slot]];
                    if (e.len == r.len && (r.len == 0 || memcmp(e.p, r.p, (size_t)r.len) == 0)) { row_u[(size_t)i] = map.vals[slot]; break; }
                    h = KeyMap::mix(h + 0x9E3779B97F4A7C15ull);
                }
            }
            if ((int)uniq.size() > maxe) return false;
            std::vector<int32_t> order(uniq.size());
            for (size_t i = 0; i < order.size(); i++) order[i] = (int32_t)i;
            std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cmp_bytes(uniq[(size_t)a], uniq[(size_t)b]) < 0; });
            std::vector<int32_t> rank(uniq.size());
            for (size_t r = 0; r < order.size(); r++) rank[(size_t)order[r]] = (int32_t)r;
            for (int i = 0; i < n; i++) d.ids[(size_t)i] = (uint32_t)rank[(size_t)row_u[(size_t)i]];
            d.entry_count = (int)uniq.size();
            size_t total = 0;
            for (auto& r : uniq) total += (size_t)r.len;
            d.entries.resize(4 * (uniq.size() + 1) + total);
            uint8_t* idx = d.entries.data();
            uint8_t* b = idx + 4 * (uniq.size() + 1);
            int32_t o = 0;
            for (size_t i = 0; i < order.size(); i++) {
                const StrRef& e = uniq[(size_t)order[i]];
                memcpy(idx + 4 * i, &o, 4);
                if (e.len) memcpy(b + o, e.p, (size_t)e.len);
                o += e.len;
            }
            memcpy(idx + 4 * uniq.size(), &o, 4);
            return true;
        }
    }
    return false;
}

// DICT_COMPRESS pack: u32 dictStructSize | u32 entryCount | u8 entriesCompressed | entries | RLE ids
void dict_compress(Sink& s, int data_type, int n, const Dict& d) {
    const uint32_t dict_size = (uint32_t)d.entries.size();
    s.u32(4 + dict_size + 4 + 4 * (uint32_t)n);
    s.u32((uint32_t)d.entry_count);
    const bool compressed = d.entry_count >= 64 && dict_size >= 256;
    s.u8(compressed ? 1 : 0);
    if (!compressed) s.bytes(d.entries.data(), d.entries.size());
    else if (data_type == IXR_INT || data_type == IXR_FLOAT) {
        std::vector<int32_t> e((size_t)d.entry_count);
        memcpy(e.data(), d.entries.data(), d.entries.size());
        delta_encode<int32_t>(s, e.data(), d.entry_count);
    } else if (data_type == IXR_LONG || data_type == IXR_DOUBLE) {
        std::vector<int64_t> e((size_t)d.entry_count);
        memcpy(e.data(), d.entries.data(), d.entries.size());
        delta_encode<int64_t>(s, e.data(), d.entry_count);
    } else {
        const uint8_t* idx = d.entries.data();
        const uint8_t* body = idx + 4 * ((size_t)d.entry_count + 1);
        std::vector<StrRef> e((size_t)d.entry_count);
        for (int i = 0; i < d.entry_count; i++) {
            int32_t a, b;
            memcpy(&a, idx + 4 * (size_t)i, 4);
            memcpy(&b, idx + 4 * (size_t)(i + 1), 4);
            e[(size_t)i] = {body + a, b - a};
        }
        delta_encode_strings(s, e.data(), d.entry_count);
    }
    rle_encode(s, d.ids.data(), n, bit_width((uint64_t)(uint32_t)(d.entry_count - 1)));
}

void delta_pack(Sink& s, int data_type, int n, const uint8_t* data) {
    if (data_type == IXR_INT || data_type == IXR_FLOAT) {
        std::vector<int32_t> v((size_t)n);
        memcpy(v.data(), data, (size_t)n * 4);
        delta_encode<int32_t>(s, v.data(), n);
    } else if (data_type == IXR_LONG || data_type == IXR_DOUBLE) {
        std::vector<int64_t> v((size_t)n);
        memcpy(v.data(), data, (size_t)n * 8);
        delta_encode<int64_t>(s, v.data(), n);
    } else {
        const uint8_t* body = data + 4 * ((size_t)n + 1);
        std::vector<StrRef> rows((size_t)n);
        for (int i = 0; i < n; i++) {
            int32_t a, b;
            memcpy(&a, data + 4 * (size_t)i, 4);
            memcpy(&b, data + 4 * (size_t)(i + 1), 4);
            rows[(size_t)i] = {body + a, b - a};
        }
        delta_encode_strings(s, rows.data(), n);
    }
}

}  // namespace

extern "C" int32_t ixr_compress_bound(int32_t obj_count, int32_t data_size) {
    int64_t size = (int64_t)data_size + ((int64_t)obj_count << 2) + 1024;  // Codec.outputBufferSize, DICT_COMPRESS branch
    return (int32_t)std::min<int64_t>(size, INT32_MAX);
}

extern "C" int32_t ixr_compress_pack(int32_t data_type, int32_t is_indexed, int32_t obj_count, const void* data_, int32_t data_size,
                                     void* out, int32_t out_limit, int32_t* encode_type) {
    if (!data_ || !out || obj_count <= 0 || obj_count > IXR_PACK_MAX_ROWS || data_size < 0) return IXR_ERR_INVALID_ARGUMENT;
    if (data_type < IXR_INT || data_type > IXR_STRING) return IXR_ERR_UNSUPPORTED_TYPE;
    const uint8_t* data = (const uint8_t*)data_;
    const int n = obj_count;
    if (data_type != IXR_STRING) {
        const int w = (data_type == IXR_INT || data_type == IXR_FLOAT) ? 4 : 8;
        if ((int64_t)n * w != data_size) return IXR_ERR_INVALID_ARGUMENT;
    } else {
        if (data_size < 4 * (n + 1)) return IXR_ERR_INVALID_ARGUMENT;
        int32_t total;
        memcpy(&total, data + 4 * (size_t)n, 4);
        if ((int64_t)4 * (n + 1) + total != data_size) return IXR_ERR_INVALID_ARGUMENT;
    }
    // INT, not indexed: dictionary only when <= ceil(n * 0.2f) distinct values, else DELTA; everything else: dictionary
    // with ratio 1 (which always succeeds), PackCompressor_VLT.java:29-113.
    const float ratio = (data_type == IXR_INT && !is_indexed) ? 0.2f : 1.0f;
    Sink s;
    s.buf.reserve((size_t)data_size / 2 + 1024);
    Dict d;
    int type;
    if (build_dict(data_type, ratio, n, data, d)) {
        dict_compress(s, data_type, n, d);
        type = IXR_ENC_DICT_COMPRESS;
    } else {
        if (is_indexed) return IXR_ERR_DATA_IMPROPER;
        delta_pack(s, data_type, n, data);
        type = IXR_ENC_DELTA;
    }
    if (s.buf.size() > (size_t)out_limit) return IXR_ERR_BUFFER_OVERFLOW;
    memcpy(out, s.buf.data(), s.buf.size());
    if (encode_type) *encode_type = type;
    return (int32_t)s.buf.size();
}
