// ixr_ext.hpp -- host-side mirrors of the callers either side of the decode, in batch form, over the C ABI of libixrgpu:
//   PackExtIndexBatch   PackExtIndex.equal / in / greater / between / like for many packs at once
//                       (indexr-segment/src/main/java/io/indexr/segment/PackExtIndex.java:14-39); a pack gets the index
//                       VersionAdapter_VLT.createExtIndex would give it (VersionAdapter_VLT.java:167-181): ExtIndex_DictBits
//                       when dpn.isDictEncoded(), ExtIndex_SimpleBits otherwise
//   PackEncodeBatch     DataPackCreator_VLT.from(...) + DataPack.compress for many packs at once
//                       (DataPackCreator_VLT.java:21-213, DataPack.java:150-163)
//   PackForeachBatch    DataPack.foreach(BitMap position, XxxSetter) for many packs at once (DataPack.java:273-306,
//                       324-357, 375-408, 426-459, 708-762): packs are decoded and the picked rows compacted on the device,
//                       only those come back
#ifndef IXR_EXT_HPP
#define IXR_EXT_HPP
#include "ixr_host.hpp"

namespace indexr {

// io.indexr.util.BitMap as a result of a pack filter: the NONE / ALL constants or 1024 words, row r = bit r & 63 of word r >> 6.
struct BitMap {
    enum Kind { NONE = IXR_BITMAP_NONE, ALL = IXR_BITMAP_ALL, SOME = IXR_BITMAP_SOME };
    Kind kind = NONE;
    int rowCount = 0;
    std::vector<uint64_t> words;  // only for SOME
    bool get(int row) const {
        if (row < 0 || row >= rowCount || kind == NONE) return false;
        if (kind == ALL) return true;
        return (words[(size_t)row >> 6] >> (row & 63)) & 1u;
    }
    int cardinality() const {
        if (kind == NONE) return 0;
        if (kind == ALL) return rowCount;
        int c = 0;
        for (uint64_t w : words) c += __builtin_popcountll(w);
        return c;
    }
};

// One predicate per pack.  Numeric keys are the reference's `long numValue` (float / double keys: Double.doubleToRawLongBits
// of the key as a double), string keys its UTF8String bytes.
struct PackPredicate {
    int32_t op = IXR_FILTER_EQUAL;  // IXR_FILTER_*
    std::vector<int64_t> numValues;
    std::vector<std::string> strValues;
    static PackPredicate equal(int64_t v) { return {IXR_FILTER_EQUAL, {v}, {}}; }
    static PackPredicate equal(std::string v) { return {IXR_FILTER_EQUAL, {}, {std::move(v)}}; }
    static PackPredicate in(std::vector<int64_t> v) { return {IXR_FILTER_IN, std::move(v), {}}; }
    static PackPredicate in(std::vector<std::string> v) { return {IXR_FILTER_IN, {}, std::move(v)}; }
    static PackPredicate greater(int64_t v, bool acceptEqual) { return {acceptEqual ? IXR_FILTER_GREATER_EQUAL : IXR_FILTER_GREATER, {v}, {}}; }
    static PackPredicate greater(std::string v, bool acceptEqual) { return {acceptEqual ? IXR_FILTER_GREATER_EQUAL : IXR_FILTER_GREATER, {}, {std::move(v)}}; }
    static PackPredicate between(int64_t a, int64_t b) { return {IXR_FILTER_BETWEEN, {a, b}, {}}; }
    static PackPredicate between(std::string a, std::string b) { return {IXR_FILTER_BETWEEN, {}, {std::move(a), std::move(b)}}; }
    static PackPredicate like(std::string pattern) { return {IXR_FILTER_LIKE, {}, {std::move(pattern)}}; }
};

class PackExtIndexBatch {
public:
    explicit PackExtIndexBatch(int device = 0) : device_(device) {}

    int add(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData, PackPredicate pred) {
        if ((size_t)dpn.packSize != cmpData.size()) throw std::runtime_error("cmpData size != dpn.packSize");
        items_.push_back({dataType, dpn, cmpData, std::move(pred)});
        return (int)items_.size() - 1;
    }
    int size() const { return (int)items_.size(); }
    const BitMap& result(int i) const { return results_.at((size_t)i); }

    // Evaluates every predicate.  Throws where the reference throws (LIKE on a numeric column, a decode failure, the
    // checkState of ExtIndex_DictBits.between).
    void run() {
        results_.assign(items_.size(), BitMap());
        std::vector<int> dict, scan;
        for (int i = 0; i < (int)items_.size(); i++) (items_[(size_t)i].dpn.compress && items_[(size_t)i].dpn.isDictEncoded() ? dict : scan).push_back(i);
        if (!dict.empty()) run_part(dict, true);
        if (!scan.empty()) run_part(scan, false);
    }

private:
    struct Item { uint8_t dataType; DataPackNode dpn; ByteSlice cmp; PackPredicate pred; };
    struct Dev {  // device / pinned memory released on scope exit
        int device; std::vector<void*> dev; std::vector<void*> pinned; ixr_plan* plan = nullptr;
        explicit Dev(int d) : device(d) {}
        ~Dev() { if (plan) ixr_plan_destroy(plan); for (void* p : dev) ixr_dev_free(device, p); for (void* p : pinned) ixr_host_free_pinned(p); }
        void* alloc(uint64_t n) { void* p = ixr_dev_alloc(device, std::max<uint64_t>(n, 16)); if (!p) throw ixr_failure("ixr_dev_alloc", IXR_ERR_OUT_OF_MEMORY); dev.push_back(p); return p; }
        uint8_t* pin(uint64_t n) { void* p = ixr_host_alloc_pinned(std::max<uint64_t>(n, 16)); if (!p) throw ixr_failure("ixr_host_alloc_pinned", IXR_ERR_OUT_OF_MEMORY); pinned.push_back(p); return (uint8_t*)p; }
    };
    // keys of one predicate, appended to `values` at a multiple of 8: numbers as int64, strings as a StringsStruct
    static uint64_t append_keys(std::vector<uint8_t>& values, const Item& it, int32_t* n_values) {
        const uint64_t at = values.size();
        const bool str = it.dataType == ColumnType::STRING || it.pred.op == IXR_FILTER_LIKE;
        if (str) {
            const size_t k = it.pred.strValues.size();
            *n_values = (int32_t)k;
            int32_t off = 0;
            for (size_t i = 0; i <= k; i++) {
                const uint8_t* p = (const uint8_t*)&off;
                values.insert(values.end(), p, p + 4);
                if (i < k) off += (int32_t)it.pred.strValues[i].size();
            }
            for (auto& s : it.pred.strValues) values.insert(values.end(), s.begin(), s.end());
        } else {
            *n_values = (int32_t)it.pred.numValues.size();
            const uint8_t* p = (const uint8_t*)it.pred.numValues.data();
            values.insert(values.end(), p, p + 8 * it.pred.numValues.size());
        }
        while (values.size() & 7) values.push_back(0);
        if (*n_values < 1) throw std::runtime_error("predicate without a value");
        return at;
    }
    void run_part(const std::vector<int>& which, bool on_ids) {
        Dev m(device_);
        const int n = (int)which.size();
        std::vector<ixr_pack_desc> descs((size_t)n);
        uint64_t in_bytes = 0, out_bytes = 0;
        for (int k = 0; k < n; k++) {
            const Item& it = items_[(size_t)which[(size_t)k]];
            descs[(size_t)k] = it.dpn.desc(it.dataType, in_bytes, out_bytes);
            in_bytes += ((uint64_t)it.dpn.packSize + 15) & ~15ull;
            out_bytes += ((uint64_t)it.dpn.dataSize + 15) & ~15ull;
        }
        m.plan = ixr_plan_create(device_, descs.data(), n, 0);
        if (!m.plan) throw ixr_failure("ixr_plan_create", -1);
        const uint64_t in_need = ixr_plan_in_bytes(m.plan);
        uint8_t* h_in = m.pin(in_need);
        std::memset(h_in, 0, in_need);
        for (int k = 0; k < n; k++) std::memcpy(h_in + descs[(size_t)k].in_off, items_[(size_t)which[(size_t)k]].cmp.address(), items_[(size_t)which[(size_t)k]].cmp.size());
        void* d_in = m.alloc(in_need);
        int32_t rc = ixr_memcpy_h2d(device_, d_in, h_in, in_need);
        if (rc) throw ixr_failure("ixr_memcpy_h2d", rc);
        std::vector<uint8_t> values;
        std::vector<int32_t> res((size_t)n);
        const uint64_t bm_bytes = (uint64_t)n * IXR_BITMAP_BYTES;
        void* d_bm = m.alloc(bm_bytes);
        if (on_ids) {  // ExtIndex_DictBits
            std::vector<ixr_filter_desc> fd((size_t)n);
            for (int k = 0; k < n; k++) {
                const Item& it = items_[(size_t)which[(size_t)k]];
                fd[(size_t)k].op = it.pred.op;
                fd[(size_t)k].values_off = append_keys(values, it, &fd[(size_t)k].n_values);
            }
            values.resize(values.size() + 8);
            rc = ixr_plan_run_filter(m.plan, d_in, in_need, fd.data(), values.data(), values.size(), d_bm, bm_bytes, res.data(), nullptr);
            if (rc < 0) throw ixr_failure("ixr_plan_run_filter", rc);
        } else {  // ExtIndex_SimpleBits: column.pack(packId) decodes, then every row is looked at
            const uint64_t out_need = std::max<uint64_t>(ixr_plan_out_bytes(m.plan), 16);
            void* d_out = m.alloc(out_need);
            rc = ixr_plan_run(m.plan, d_in, in_need, d_out, out_need, nullptr);
            if (rc) throw ixr_failure("ixr_plan_run", rc);
            std::vector<int32_t> status((size_t)n);
            if (ixr_plan_status(m.plan, status.data(), nullptr) != 0) throw std::runtime_error("pack(s) failed to decode (consumed != packSize)");
            std::vector<ixr_scan_filter_desc> sd((size_t)n);
            for (int k = 0; k < n; k++) {
                const Item& it = items_[(size_t)which[(size_t)k]];
                ixr_scan_filter_desc& d = sd[(size_t)k];
                d.src_off = descs[(size_t)k].out_off;
                d.data_type = it.dataType; d.obj_count = it.dpn.objCount; d.data_size = it.dpn.dataSize;
                d.op = it.pred.op; d.reserved = 0;
                d.values_off = append_keys(values, it, &d.n_values);
            }
            values.resize(values.size() + 8);
            rc = ixr_scan_filter_run(device_, sd.data(), n, d_out, out_need, values.data(), values.size(), d_bm, bm_bytes, res.data(), nullptr);
            if (rc < 0) throw ixr_failure("ixr_scan_filter_run", rc);
        }
        uint8_t* h_bm = m.pin(bm_bytes);
        rc = ixr_memcpy_d2h(device_, h_bm, d_bm, bm_bytes);
        if (rc) throw ixr_failure("ixr_memcpy_d2h", rc);
        for (int k = 0; k < n; k++) {
            const Item& it = items_[(size_t)which[(size_t)k]];
            if (res[(size_t)k] == IXR_ERR_UNSUPPORTED_TYPE) throw std::runtime_error("column type " + std::to_string(it.dataType) + " is illegal in LIKE");
            if (res[(size_t)k] == IXR_ERR_BUG) throw std::runtime_error("checkState failed in between");  // ExtIndex_DictBits.java:165-166
            if (res[(size_t)k] < 0) throw ixr_failure("pack filter", res[(size_t)k]);
            BitMap& b = results_[(size_t)which[(size_t)k]];
            b.kind = (BitMap::Kind)res[(size_t)k];
            b.rowCount = it.dpn.objCount;
            if (b.kind == BitMap::SOME) {
                b.words.resize(1024);
                std::memcpy(b.words.data(), h_bm + (size_t)k * IXR_BITMAP_BYTES, IXR_BITMAP_BYTES);
            }
        }
    }
    int device_;
    std::vector<Item> items_;
    std::vector<BitMap> results_;
};

// Compresses many freshly created packs in one device call: what DPColumn does pack by pack when a pack fills
// (DPColumn.java:230-276).  add() takes a pack as DataPack::fromInts / fromLongs / ... / fromStrings made it; after run() every
// pack holds its compressed form and a DataPackNode with encodeType, packSize, compress and (numbers) min / max / sum as
// DataPackCreator_VLT computes them.
class PackEncodeBatch {
public:
    explicit PackEncodeBatch(int device = 0) : device_(device) {}
    int add(uint8_t dataType, bool isIndexed, const DataPack& pack) {
        if (pack.data().empty()) throw std::runtime_error("no data to compress");
        items_.push_back({dataType, isIndexed, pack.dpn(), pack.data(), ByteSlice()});
        return (int)items_.size() - 1;
    }
    int size() const { return (int)items_.size(); }
    const DataPackNode& dpn(int i) const { return items_.at((size_t)i).dpn; }
    const ByteSlice& cmpData(int i) const { return items_.at((size_t)i).cmp; }
    DataPack pack(int i) const { const Item& it = items_.at((size_t)i); return DataPack(it.data, it.cmp, it.dpn); }

    void run() {
        const int n = (int)items_.size();
        if (n == 0) return;
        std::vector<ixr_encode_desc> ed((size_t)n);
        uint64_t src = 0, dst = 0;
        for (int i = 0; i < n; i++) {
            const Item& it = items_[(size_t)i];
            ixr_encode_desc& d = ed[(size_t)i];
            d = ixr_encode_desc{};
            d.src_off = src; d.dst_off = dst; d.data_type = it.dataType; d.is_indexed = it.isIndexed ? 1 : 0;
            d.obj_count = it.dpn.objCount; d.data_size = (int32_t)it.data.size();
            src += ((uint64_t)it.data.size() + 15) & ~15ull;
            dst += ((uint64_t)ixr_compress_bound(d.obj_count, d.data_size) + 15) & ~15ull;
        }
        struct Mem { int dev; void *a = nullptr, *b = nullptr; uint8_t *ha = nullptr, *hb = nullptr;
                     ~Mem() { if (a) ixr_dev_free(dev, a); if (b) ixr_dev_free(dev, b); if (ha) ixr_host_free_pinned(ha); if (hb) ixr_host_free_pinned(hb); } } m{device_};
        m.ha = (uint8_t*)ixr_host_alloc_pinned(src + 16);
        m.hb = (uint8_t*)ixr_host_alloc_pinned(dst + 16);
        m.a = ixr_dev_alloc(device_, src + 16);
        m.b = ixr_dev_alloc(device_, dst + 16);
        if (!m.ha || !m.hb || !m.a || !m.b) throw ixr_failure("arena allocation", IXR_ERR_OUT_OF_MEMORY);
        std::memset(m.ha, 0, src + 16);
        for (int i = 0; i < n; i++) std::memcpy(m.ha + ed[(size_t)i].src_off, items_[(size_t)i].data.address(), items_[(size_t)i].data.size());
        int32_t rc = ixr_memcpy_h2d(device_, m.a, m.ha, src + 16);
        if (rc) throw ixr_failure("ixr_memcpy_h2d", rc);
        rc = ixr_encode_run(device_, ed.data(), n, m.a, src + 16, m.b, dst + 16, 0, nullptr);
        if (rc < 0) throw ixr_failure("ixr_encode_run", rc);
        if (rc > 0) throw std::runtime_error(std::to_string(rc) + " pack(s) failed to compress");  // Preconditions.checkState(cmpSize >= 0), PackCompressor_VLT.java:117
        rc = ixr_memcpy_d2h(device_, m.hb, m.b, dst + 16);
        if (rc) throw ixr_failure("ixr_memcpy_d2h", rc);
        for (int i = 0; i < n; i++) {
            Item& it = items_[(size_t)i];
            const ixr_encode_desc& d = ed[(size_t)i];
            it.cmp = ByteSlice::copyOf(m.hb + d.dst_off, (size_t)d.pack_size);
            it.dpn.packSize = d.pack_size;
            it.dpn.encodeType = d.encode_type;
            it.dpn.compress = true;
            if (it.dataType != ColumnType::STRING) { it.dpn.minValue = d.min_value; it.dpn.maxValue = d.max_value; it.dpn.sum = d.sum; }
        }
    }

private:
    struct Item { uint8_t dataType; bool isIndexed; DataPackNode dpn; ByteSlice data; ByteSlice cmp; };
    int device_;
    std::vector<Item> items_;
};

// Decodes many packs and hands out the rows a BitMap picks in each, as DataPack.foreach(BitMap position, XxxSetter) does
// pack by pack: after run(), foreachInt / Long / Float / Double / Bytes(i, setter) make the same setter calls, in the same
// order, and return the same count.  BitMap::ALL is the reference's BitMap.ALL (every row), NONE picks nothing.
class PackForeachBatch {
public:
    explicit PackForeachBatch(int device = 0) : device_(device) {}
    int add(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData, const BitMap& position) {
        if ((size_t)dpn.packSize != cmpData.size()) throw std::runtime_error("cmpData size != dpn.packSize");
        if (position.kind == BitMap::SOME && position.words.size() != 1024) throw std::runtime_error("a BitMap has 1024 words");
        items_.push_back({dataType, dpn, cmpData, position, 0, 0});
        return (int)items_.size() - 1;
    }
    int size() const { return (int)items_.size(); }
    int count(int i) const { return items_.at((size_t)i).count; }

    void run() {
        const int n = (int)items_.size();
        if (n == 0) return;
        struct Dev { int device; std::vector<void*> dev; ixr_plan* plan = nullptr;
                     ~Dev() { if (plan) ixr_plan_destroy(plan); for (void* p : dev) ixr_dev_free(device, p); }
                     void* alloc(uint64_t b) { void* p = ixr_dev_alloc(device, std::max<uint64_t>(b, 16)); if (!p) throw ixr_failure("ixr_dev_alloc", IXR_ERR_OUT_OF_MEMORY); dev.push_back(p); return p; } } m{device_, {}};
        std::vector<ixr_pack_desc> descs((size_t)n);
        std::vector<ixr_select_desc> sel((size_t)n);
        uint64_t in_bytes = 0, out_bytes = 0, dst = 0;
        for (int k = 0; k < n; k++) {
            Item& it = items_[(size_t)k];
            descs[(size_t)k] = it.dpn.desc(it.dataType, in_bytes, out_bytes);
            in_bytes += ((uint64_t)it.dpn.packSize + 15) & ~15ull;
            out_bytes += ((uint64_t)it.dpn.dataSize + 15) & ~15ull;
            ixr_select_desc& d = sel[(size_t)k];
            d.src_off = descs[(size_t)k].out_off; d.dst_off = dst; d.bitmap_off = (uint64_t)k * IXR_BITMAP_BYTES;
            d.data_type = it.dataType; d.obj_count = it.dpn.objCount; d.data_size = it.dpn.dataSize;
            d.all = it.position.kind == BitMap::ALL ? 1 : 0;
            it.out_off = dst;
            dst += (ixr_select_bound(&d) + 255) & ~255ull;
        }
        m.plan = ixr_plan_create(device_, descs.data(), n, 0);
        if (!m.plan) throw ixr_failure("ixr_plan_create", -1);
        const uint64_t in_need = ixr_plan_in_bytes(m.plan), out_need = std::max<uint64_t>(ixr_plan_out_bytes(m.plan), 16);
        std::vector<uint8_t> h_in((size_t)in_need, 0), h_bm((size_t)n * IXR_BITMAP_BYTES, 0);
        for (int k = 0; k < n; k++) {
            const Item& it = items_[(size_t)k];
            std::memcpy(h_in.data() + descs[(size_t)k].in_off, it.cmp.address(), it.cmp.size());
            if (it.position.kind == BitMap::SOME) std::memcpy(h_bm.data() + (size_t)k * IXR_BITMAP_BYTES, it.position.words.data(), IXR_BITMAP_BYTES);
        }
        void *d_in = m.alloc(in_need), *d_out = m.alloc(out_need), *d_bm = m.alloc(h_bm.size()), *d_sel = m.alloc(dst);
        int32_t rc = ixr_memcpy_h2d(device_, d_in, h_in.data(), in_need);
        if (!rc) rc = ixr_memcpy_h2d(device_, d_bm, h_bm.data(), h_bm.size());
        if (rc) throw ixr_failure("ixr_memcpy_h2d", rc);
        rc = ixr_plan_run(m.plan, d_in, in_need, d_out, out_need, nullptr);
        if (rc) throw ixr_failure("ixr_plan_run", rc);
        std::vector<int32_t> status((size_t)n), counts((size_t)n);
        if (ixr_plan_status(m.plan, status.data(), nullptr) != 0) throw std::runtime_error("pack(s) failed to decode (consumed != packSize)");
        rc = ixr_select_run(device_, sel.data(), n, d_out, out_need, d_bm, h_bm.size(), d_sel, std::max<uint64_t>(dst, 16), counts.data(), nullptr);
        if (rc < 0) throw ixr_failure("ixr_select_run", rc);
        if (rc > 0) throw std::runtime_error(std::to_string(rc) + " pack(s) with string offsets outside the pack");
        picked_.assign((size_t)dst, 0);
        rc = ixr_memcpy_d2h(device_, picked_.data(), d_sel, dst);
        if (rc) throw ixr_failure("ixr_memcpy_d2h", rc);
        for (int k = 0; k < n; k++) items_[(size_t)k].count = counts[(size_t)k];
    }

    template <typename F> int foreachInt(int i, F setter) const { return numbers<int32_t>(i, ColumnType::INT, setter); }
    template <typename F> int foreachLong(int i, F setter) const { return numbers<int64_t>(i, ColumnType::LONG, setter); }
    template <typename F> int foreachFloat(int i, F setter) const { return numbers<float>(i, ColumnType::FLOAT, setter); }
    template <typename F> int foreachDouble(int i, F setter) const { return numbers<double>(i, ColumnType::DOUBLE, setter); }
    // setter(index, BytePiece): the piece points into this batch's memory
    template <typename F> int foreachBytes(int i, F setter) const {
        const Item& it = item(i, ColumnType::STRING);
        const size_t n = (size_t)it.dpn.objCount;
        const uint8_t* base = picked_.data() + it.out_off;
        DataPack::BytePiece bp;
        for (int k = 0; k < it.count; k++) {
            int32_t id, off, len;
            std::memcpy(&id, base + 4 * (size_t)k, 4);
            std::memcpy(&off, base + 4 * n + 4 * (size_t)k, 4);
            std::memcpy(&len, base + 8 * n + 4 * (size_t)k, 4);
            bp.addr = base + 12 * n + (size_t)off;
            bp.len = len;
            setter((int)id, bp);
        }
        return it.count;
    }

private:
    struct Item { uint8_t dataType; DataPackNode dpn; ByteSlice cmp; BitMap position; uint64_t out_off; int count; };
    const Item& item(int i, uint8_t type) const {
        const Item& it = items_.at((size_t)i);
        if (it.dataType != type) throw std::runtime_error("pack " + std::to_string(i) + " holds another data type");
        return it;
    }
    template <typename T, typename F> int numbers(int i, uint8_t type, F setter) const {
        const Item& it = item(i, type);
        const size_t n = (size_t)it.dpn.objCount;
        const uint8_t* base = picked_.data() + it.out_off;
        const uint8_t* vals = base + ((4 * n + 7) & ~(size_t)7);
        for (int k = 0; k < it.count; k++) {
            int32_t id;
            T v;
            std::memcpy(&id, base + 4 * (size_t)k, 4);
            std::memcpy(&v, vals + sizeof(T) * (size_t)k, sizeof(T));
            setter((int)id, v);
        }
        return it.count;
    }
    int device_;
    std::vector<Item> items_;
    std::vector<uint8_t> picked_;
};

}  // namespace indexr
#endif  // IXR_EXT_HPP
