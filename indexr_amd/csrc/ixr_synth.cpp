// ixr_synth.cpp -- deterministic synthetic columns and multi-threaded pack building (host only). See include/ixr_synth.h.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ixr_synth.h"

namespace {

inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {  // xoshiro256**
    uint64_t s[4];
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix64(seed); }
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }  // [0, n)
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }            // [0, 1)
};

inline uint64_t pack_seed(uint64_t seed, int kind, uint32_t pack_id) {
    uint64_t x = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(kind + 1)) ^ (0x8CB92BA72F3D8DD7ull * ((uint64_t)pack_id + 1));
    return splitmix64(x);
}

const char kAlnum[] = "abcdefghijklmnopqrstuvwxyz0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";

// seed-dependent lookup tables shared by all packs
struct Tables {
    uint64_t seed;
    std::vector<std::string> cats;  // 32 categories
    std::vector<std::string> urls;  // 50000 URLs
    std::vector<double> zipf_cdf;   // Zipf(1.1) over the URLs
    std::vector<int64_t> lcard;     // 4096 sparse longs
    std::vector<uint32_t> zipf_bucket;  // first URL index whose cdf reaches k/4096: narrows the search per draw
    explicit Tables(uint64_t sd) : seed(sd) {
        Rng r(sd ^ 0xA5A5A5A5DEADBEEFull);
        for (int i = 0; i < 32; i++) {
            int len = 3 + (int)r.below(10);
            std::string s;
            for (int k = 0; k < len; k++) s.push_back((char)('a' + r.below(26)));
            cats.push_back(s);
        }
        std::vector<std::string> doms;
        for (int i = 0; i < 200; i++) {
            int len = 4 + (int)r.below(9);
            std::string s;
            for (int k = 0; k < len; k++) s.push_back((char)('a' + r.below(26)));
            doms.push_back(s);
        }
        urls.reserve(50000);
        for (int i = 0; i < 50000; i++) {
            std::string s = "www." + doms[r.below(200)] + ".com/";
            int target = 20 + (int)r.below(61);
            while ((int)s.size() < target) {
                uint32_t c = r.below(40);
                s.push_back(c < 36 ? kAlnum[c] : (c < 38 ? '/' : '-'));
            }
            urls.push_back(s);
        }
        zipf_cdf.resize(50000);
        double acc = 0;
        for (int i = 0; i < 50000; i++) { acc += 1.0 / std::pow((double)(i + 1), 1.1); zipf_cdf[(size_t)i] = acc; }
        for (auto& v : zipf_cdf) v /= acc;
        zipf_bucket.resize(4097);
        for (int k = 0; k <= 4096; k++)
            zipf_bucket[(size_t)k] = (uint32_t)(std::lower_bound(zipf_cdf.begin(), zipf_cdf.end(), (double)k / 4096.0) - zipf_cdf.begin());
        lcard.resize(4096);
        for (auto& v : lcard) v = (int64_t)(r.next() & ((1ull << 40) - 1));
    }
};

std::shared_ptr<const Tables> tables_for(uint64_t seed) {
    static std::mutex mu;
    static std::shared_ptr<const Tables> cur;
    std::lock_guard<std::mutex> lk(mu);
    if (!cur || cur->seed != seed) cur = std::make_shared<const Tables>(seed);
    return cur;
}

struct StrBuilder {  // StringsStruct: int32 offset[n+1] | bodies
    std::vector<int32_t> off;
    std::vector<uint8_t> body;
    void add(const char* p, size_t n) { off.push_back((int32_t)body.size()); body.insert(body.end(), p, p + n); }
    int32_t finish(void* out, int32_t cap) {
        off.push_back((int32_t)body.size());
        size_t total = off.size() * 4 + body.size();
        if (total > (size_t)cap) return IXR_ERR_BUFFER_OVERFLOW;
        memcpy(out, off.data(), off.size() * 4);
        if (!body.empty()) memcpy((uint8_t*)out + off.size() * 4, body.data(), body.size());  // an all-empty column has no body
        return (int32_t)total;
    }
};

}  // namespace

extern "C" int32_t ixr_synth_data_type(int32_t kind) {
    switch (kind) {
        case IXR_COL_I_SEQ: case IXR_COL_I_RAND: case IXR_COL_I_CAT: case IXR_COL_I_RUNS: return IXR_INT;
        case IXR_COL_L_TS: case IXR_COL_L_CARD: case IXR_COL_L_RAND: return IXR_LONG;
        case IXR_COL_F_CARD: return IXR_FLOAT;
        case IXR_COL_D_CARD: return IXR_DOUBLE;
        case IXR_COL_S_CAT: case IXR_COL_S_URL: case IXR_COL_S_UNIQ: case IXR_COL_S_EMPTY: return IXR_STRING;
        default: return IXR_ERR_INVALID_ARGUMENT;
    }
}

extern "C" int32_t ixr_synth_column(int32_t kind, uint64_t seed, uint32_t pack_id, int32_t rows, void* out, int32_t cap) {
    if (!out || rows <= 0 || rows > IXR_PACK_MAX_ROWS) return IXR_ERR_INVALID_ARGUMENT;
    const int dt = ixr_synth_data_type(kind);
    if (dt < 0) return dt;
    Rng r(pack_seed(seed, kind, pack_id));
    const int n = rows;
    if (dt != IXR_STRING) {
        const int w = (dt == IXR_INT || dt == IXR_FLOAT) ? 4 : 8;
        if ((int64_t)n * w > cap) return IXR_ERR_BUFFER_OVERFLOW;
    }
    switch (kind) {
        case IXR_COL_I_SEQ: {
            int32_t* v = (int32_t*)out;
            uint32_t cur = pack_id * 98304u;
            for (int i = 0; i < n; i++) { cur += r.below(4); v[i] = (int32_t)cur; }
            return n * 4;
        }
        case IXR_COL_I_RAND: {
            int32_t* v = (int32_t*)out;
            for (int i = 0; i < n; i++) v[i] = (int32_t)(uint32_t)(r.next() >> 32);
            return n * 4;
        }
        case IXR_COL_I_CAT: {
            int32_t* v = (int32_t*)out;
            for (int i = 0; i < n; i++) v[i] = (int32_t)(r.below(1000) * 1000003u - 500000000u);
            return n * 4;
        }
        case IXR_COL_I_RUNS: {
            int32_t* v = (int32_t*)out;
            int i = 0;
            const double lq = std::log(1.0 - 1.0 / 32.0);
            while (i < n) {
                int len = 1 + (int)(std::log(1.0 - r.unit()) / lq);
                int32_t val = (int32_t)(r.below(16) * 37u + 5u);
                for (int k = 0; k < len && i < n; k++) v[i++] = val;
            }
            return n * 4;
        }
        case IXR_COL_L_TS: {
            int64_t* v = (int64_t*)out;
            int64_t cur = 1500000000000ll + (int64_t)pack_id * 65536ll * 1000ll;
            for (int i = 0; i < n; i++) { cur += (int64_t)r.below(2001); v[i] = cur; }
            return n * 8;
        }
        case IXR_COL_L_CARD: {
            auto t = tables_for(seed);
            int64_t* v = (int64_t*)out;
            for (int i = 0; i < n; i++) v[i] = t->lcard[r.below(4096)];
            return n * 8;
        }
        case IXR_COL_L_RAND: {
            int64_t* v = (int64_t*)out;
            for (int i = 0; i < n; i++) v[i] = (int64_t)r.next();
            return n * 8;
        }
        case IXR_COL_F_CARD: {
            float* v = (float*)out;
            for (int i = 0; i < n; i++) v[i] = (float)((double)r.below(10000) / 10.0);
            return n * 4;
        }
        case IXR_COL_D_CARD: {
            double* v = (double*)out;
            for (int i = 0; i < n; i++) v[i] = (double)r.below(50000) / 10.0;
            return n * 8;
        }
        case IXR_COL_S_CAT: {
            auto t = tables_for(seed);
            StrBuilder b;
            for (int i = 0; i < n; i++) { const std::string& s = t->cats[r.below(32)]; b.add(s.data(), s.size()); }
            return b.finish(out, cap);
        }
        case IXR_COL_S_URL: {
            auto t = tables_for(seed);
            StrBuilder b;
            for (int i = 0; i < n; i++) {
                double u = r.unit();
                const size_t bk = (size_t)(u * 4096.0);
                auto lo = t->zipf_cdf.begin() + t->zipf_bucket[bk];
                auto hi = t->zipf_cdf.begin() + std::min<size_t>(t->zipf_bucket[bk + 1] + 1, t->zipf_cdf.size());
                size_t k = (size_t)(std::lower_bound(lo, hi, u) - t->zipf_cdf.begin());
                if (k >= t->urls.size()) k = t->urls.size() - 1;
                const std::string& s = t->urls[k];
                b.add(s.data(), s.size());
            }
            return b.finish(out, cap);
        }
        case IXR_COL_S_UNIQ: {
            StrBuilder b;
            char tmp[32];
            for (int i = 0; i < n; i++) {
                int len = (int)r.below(25);
                for (int k = 0; k < len; k++) tmp[k] = kAlnum[r.below(62)];
                b.add(tmp, (size_t)len);
            }
            return b.finish(out, cap);
        }
        case IXR_COL_S_EMPTY: {
            StrBuilder b;
            char tmp[32];
            for (int i = 0; i < n; i++) {
                int len = r.below(10) == 0 ? 1 + (int)r.below(16) : 0;
                for (int k = 0; k < len; k++) tmp[k] = (char)('a' + r.below(26));
                b.add(tmp, (size_t)len);
            }
            return b.finish(out, cap);
        }
    }
    return IXR_ERR_INVALID_ARGUMENT;
}

extern "C" uint64_t ixr_checksum(const void* data, uint64_t bytes) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t sum = 0;
    const uint64_t words = bytes >> 3;
    for (uint64_t i = 0; i < words; i++) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        sum += (w + 0x9E3779B97F4A7C15ull) * (2 * i + 1);
    }
    if (bytes & 7) {
        uint64_t w = 0;
        memcpy(&w, p + 8 * words, (size_t)(bytes & 7));
        sum += (w + 0x9E3779B97F4A7C15ull) * (2 * words + 1);
    }
    return sum;
}

extern "C" void ixr_synth_free(uint8_t* arena) { free(arena); }

extern "C" int32_t ixr_synth_build(const ixr_synth_req* reqs, int32_t n, uint64_t seed, int32_t threads, ixr_pack_desc* descs,
                                   uint8_t** arena, uint64_t* arena_bytes, uint64_t* out_bytes, uint64_t* checksums) {
    if (!reqs || n <= 0 || !descs || !arena || !arena_bytes) return IXR_ERR_INVALID_ARGUMENT;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    (void)tables_for(seed);  // build the tables once, outside the workers
    std::vector<std::vector<uint8_t>> packs((size_t)n);
    std::atomic<int> next(0), err(0);
    auto worker = [&]() {
        std::vector<uint8_t> data, cmp;
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n || err.load()) break;
            const ixr_synth_req& q = reqs[i];
            const int dt = ixr_synth_data_type(q.kind);
            if (dt < 0 || q.rows <= 0 || q.rows > IXR_PACK_MAX_ROWS) { err = IXR_ERR_INVALID_ARGUMENT; break; }
            size_t cap = dt == IXR_STRING ? 4 * ((size_t)q.rows + 1) + 96 * (size_t)q.rows : 8 * (size_t)q.rows;
            data.resize(cap);
            int32_t ds = ixr_synth_column(q.kind, seed, q.pack_id, q.rows, data.data(), (int32_t)cap);
            if (ds < 0) { err = ds; break; }
            int32_t bound = ixr_compress_bound(q.rows, ds);
            cmp.resize((size_t)bound);
            int32_t et = 0;
            int32_t cs = ixr_compress_pack(dt, q.is_indexed, q.rows, data.data(), ds, cmp.data(), bound, &et);
            if (cs < 0) { err = cs; break; }
            packs[(size_t)i].assign(cmp.begin(), cmp.begin() + cs);
            ixr_pack_desc& d = descs[i];
            memset(&d, 0, sizeof(d));
            d.pack_size = cs; d.data_size = ds; d.obj_count = q.rows; d.encode_type = et; d.data_type = dt;
            if (checksums) checksums[i] = ixr_checksum(data.data(), (uint64_t)ds);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (err.load()) return err.load();
    uint64_t in_off = 0, out_off = 0;
    for (int i = 0; i < n; i++) {
        descs[i].in_off = in_off;
        descs[i].out_off = out_off;
        in_off += ((uint64_t)descs[i].pack_size + 15) & ~15ull;
        out_off += ((uint64_t)descs[i].data_size + 15) & ~15ull;
    }
    const uint64_t total = in_off + IXR_IN_SLACK;
    uint8_t* a = (uint8_t*)calloc(1, (size_t)total);
    if (!a) return IXR_ERR_OUT_OF_MEMORY;
    for (int i = 0; i < n; i++) {
        memcpy(a + descs[i].in_off, packs[(size_t)i].data(), packs[(size_t)i].size());
        std::vector<uint8_t>().swap(packs[(size_t)i]);
    }
    *arena = a;
    *arena_bytes = total;
    if (out_bytes) *out_bytes = out_off;
    return 0;
}
