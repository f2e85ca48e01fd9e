// ixr_host.hpp -- C++ host side above the C ABI of libixrgpu, shaped like the reference's pack reader API so that a
// caller (or a test) written against io.indexr.segment.pack reads the same here.  Header only; needs include/ixr_gpu.h
// and -lixrgpu.  Names, argument meaning and error behaviour follow (paths relative to the reference tree):
//   ByteSlice            indexr-common/src/main/java/io/indexr/io/ByteSlice.java
//   DataPackNode(_VLT)   indexr-vlt-segment/.../vlt/segment/pack/DataPackNode_VLT.java:12-102
//   PackCompressor       indexr-segment/.../segment/pack/PackCompressor.java:8-19
//   PackCompressor_VLT   indexr-vlt-segment/.../vlt/segment/pack/PackCompressor_VLT.java:23-213
//   VersionAdapter       indexr-segment/.../segment/storage/VersionAdapter.java:42-48 (the three decode methods + compressPack)
//   DataPack             indexr-segment/.../segment/pack/DataPack.java:90-186 (dictStruct, compress, decompress), accessors :196-529
//   DataPackCreator_VLT  indexr-vlt-segment/.../vlt/segment/pack/DataPackCreator_VLT.java:21-213 (min / max / sum / dataSize)
// Java exceptions become std::runtime_error with the reference's message where it has one.
#ifndef IXR_HOST_HPP
#define IXR_HOST_HPP
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "ixr_gpu.h"

namespace indexr {

// io.indexr.segment.ColumnType / DataType ids (DataType.java:4-8)
namespace ColumnType { enum : uint8_t { INT = IXR_INT, LONG = IXR_LONG, FLOAT = IXR_FLOAT, DOUBLE = IXR_DOUBLE, STRING = IXR_STRING }; }
// CodecType ids (CodecType.java:4-10)
namespace CodecType { enum : int32_t { PLAIN = 1, DELTA = 2, RLE = 3, SIMPLE_DICT = 4, DICT = 5, DICT_COMPRESS = 6 }; }
namespace Version { enum : int { VERSION_7_ID = 7, VERSION_8_ID = 8, LATEST_ID = 8 }; }

constexpr int MAX_COUNT = IXR_PACK_MAX_ROWS;  // DataPack.java:36-38
inline int rowCountToPackCount(int64_t rowCount) { return (int)((rowCount + MAX_COUNT - 1) >> 16); }  // DataPack.java:773-776
inline int packRowCount(int64_t rowCount, int packId) {  // DataPack.java:778-803
    const int packCount = rowCountToPackCount(rowCount);
    if (packId < packCount - 1) return MAX_COUNT;
    const int rest = (int)(rowCount & (MAX_COUNT - 1));
    return rest == 0 ? MAX_COUNT : rest;
}

// Off-heap byte range with optional ownership; fresh allocations are zero-filled like ByteBuffer.allocateDirect
// (ByteBufferUtil.java:31-36).
class ByteSlice {
public:
    ByteSlice() = default;
    static ByteSlice allocateDirect(size_t n) {
        ByteSlice s;
        s.own_ = std::shared_ptr<uint8_t>((uint8_t*)std::calloc(n ? n : 1, 1), std::free);
        if (!s.own_) throw std::bad_alloc();
        s.p_ = s.own_.get();
        s.n_ = n;
        return s;
    }
    static ByteSlice fromAddress(const void* p, size_t n) {  // autoFree = false (ByteSlice.java:214-216)
        ByteSlice s;
        s.p_ = (uint8_t*)p;
        s.n_ = n;
        return s;
    }
    static ByteSlice copyOf(const void* p, size_t n) {
        ByteSlice s = allocateDirect(n);
        if (n) std::memcpy(s.p_, p, n);
        return s;
    }
    uint8_t* address() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    int32_t getInt(size_t off) const { int32_t v; std::memcpy(&v, p_ + off, 4); return v; }
    int64_t getLong(size_t off) const { int64_t v; std::memcpy(&v, p_ + off, 8); return v; }
    void putInt(size_t off, int32_t v) { std::memcpy(p_ + off, &v, 4); }
    void putLong(size_t off, int64_t v) { std::memcpy(p_ + off, &v, 8); }
    void free() { own_.reset(); p_ = nullptr; n_ = 0; }

private:
    std::shared_ptr<uint8_t> own_;
    uint8_t* p_ = nullptr;
    size_t n_ = 0;
};

// The 73-byte little-endian DPN record of VLT segments.
struct DataPackNode {
    static constexpr int SIZE = 4 + 1 + 8 + 4 + 8 + 4 + 8 + 4 + 8 + 8 + 4 + 4 + 8;  // DataPackNode_VLT.java:17
    int32_t objCount = 0;
    bool compress = true;
    int64_t packAddr = 0;
    int32_t packSize = 0;
    int64_t indexAddr = 0;
    int32_t indexSize = 0;
    int64_t extIndexAddr = 0;
    int32_t extIndexSize = 0;
    int64_t minValue = 0, maxValue = 0;
    int32_t encodeType = 0;
    int32_t dataSize = 0;
    int64_t sum = 0;

    bool isDictEncoded() const { return encodeType == CodecType::DICT_COMPRESS; }  // DataPackNode_VLT.java:25

    void write(uint8_t* out) const {  // DataPackNode_VLT.java:58-79
        size_t o = 0;
        auto put = [&](const void* p, size_t n) { std::memcpy(out + o, p, n); o += n; };
        const uint8_t c = compress ? 1 : 0;
        put(&objCount, 4); put(&c, 1); put(&packAddr, 8); put(&packSize, 4); put(&indexAddr, 8); put(&indexSize, 4);
        put(&extIndexAddr, 8); put(&extIndexSize, 4); put(&minValue, 8); put(&maxValue, 8); put(&encodeType, 4);
        put(&dataSize, 4); put(&sum, 8);
    }
    static DataPackNode from(const uint8_t* in) {  // DataPackNode_VLT.java:81-102
        DataPackNode d;
        size_t o = 0;
        auto get = [&](void* p, size_t n) { std::memcpy(p, in + o, n); o += n; };
        uint8_t c;
        get(&d.objCount, 4); get(&c, 1); d.compress = (c & 1) == 1; get(&d.packAddr, 8); get(&d.packSize, 4);
        get(&d.indexAddr, 8); get(&d.indexSize, 4); get(&d.extIndexAddr, 8); get(&d.extIndexSize, 4); get(&d.minValue, 8);
        get(&d.maxValue, 8); get(&d.encodeType, 4); get(&d.dataSize, 4); get(&d.sum, 8);
        return d;
    }
    // what the batch decoder needs of a DPN
    ixr_pack_desc desc(uint8_t dataType, uint64_t in_off, uint64_t out_off) const {
        ixr_pack_desc p{};
        p.in_off = in_off; p.out_off = out_off; p.pack_size = packSize; p.data_size = dataSize; p.obj_count = objCount;
        p.encode_type = compress ? encodeType : IXR_ENC_RAW;
        p.data_type = dataType;
        return p;
    }
};

inline std::runtime_error ixr_failure(const char* what, int32_t rc) {
    return std::runtime_error(std::string(what) + " failed: code " + std::to_string(rc) + " (" + ixr_last_error() + ")");
}

// PackCompressor.java:8-19
class PackCompressor {
public:
    virtual ~PackCompressor() = default;
    virtual ByteSlice compress(uint8_t dataType, bool isIndexed, DataPackNode& dpn, const ByteSlice& data) = 0;
    virtual ByteSlice decompress(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) = 0;
    virtual ByteSlice getDictStruct(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) = 0;
    virtual ByteSlice getDataFromDictStruct(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& dictStruct) = 0;
};

// PackCompressor_VLT with the decode side on the GPU.  compress() runs on the host (it only exists to make inputs).
class PackCompressor_VLT_GPU : public PackCompressor {
public:
    explicit PackCompressor_VLT_GPU(int device = 0) : device_(device) {}

    ByteSlice compress(uint8_t dataType, bool isIndexed, DataPackNode& dpn, const ByteSlice& data) override {  // :23-128
        const int32_t bound = ixr_compress_bound(dpn.objCount, (int32_t)data.size());
        ByteSlice tmp = ByteSlice::allocateDirect((size_t)bound);
        int32_t et = 0;
        const int32_t sz = ixr_compress_pack(dataType, isIndexed ? 1 : 0, dpn.objCount, data.address(), (int32_t)data.size(),
                                             tmp.address(), bound, &et);
        if (sz < 0) throw ixr_failure("ixr_compress_pack", sz);
        dpn.encodeType = et;
        dpn.packSize = sz;
        dpn.compress = true;
        return ByteSlice::copyOf(tmp.address(), (size_t)sz);
    }
    ByteSlice decompress(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) override {  // :130-170
        check_type(dataType);
        if (dpn.encodeType != CodecType::DICT_COMPRESS && dpn.encodeType != CodecType::SIMPLE_DICT && dpn.encodeType != CodecType::DELTA)
            throw std::runtime_error("Illegal encode type");  // :146
        ByteSlice data = ByteSlice::allocateDirect((size_t)dpn.dataSize);  // :133
        const int32_t consumed = ixr_decode_pack(device_, dataType, dpn.encodeType, dpn.objCount, cmpData.address(),
                                                 (int32_t)cmpData.size(), data.address(), dpn.dataSize);
        if (consumed < 0) throw ixr_failure("ixr_decode_pack", consumed);
        if (consumed != dpn.packSize) throw std::runtime_error("consumed != packSize");  // Preconditions.checkState :168
        return data;
    }
    ByteSlice getDictStruct(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) override {  // :172-186
        check_type(dataType);
        if (cmpData.size() < 4) throw std::runtime_error("pack too small");
        const int32_t dictStructSize = cmpData.getInt(0);  // :177
        if (dictStructSize < 8) throw std::runtime_error("bad dictStructSize");
        ByteSlice ds = ByteSlice::allocateDirect((size_t)dictStructSize);
        const int32_t rc = ixr_get_dict_struct(device_, dataType, dpn.objCount, cmpData.address(), (int32_t)cmpData.size(),
                                               ds.address(), dictStructSize);
        if (rc < 0) throw ixr_failure("ixr_get_dict_struct", rc);
        return ds;
    }
    ByteSlice getDataFromDictStruct(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& dictStruct) override {  // :188-213
        check_type(dataType);
        ByteSlice data = ByteSlice::allocateDirect((size_t)dpn.dataSize);
        const int32_t rc = ixr_get_data_from_dict_struct(device_, dataType, dpn.objCount, dictStruct.address(),
                                                         (int32_t)dictStruct.size(), data.address(), dpn.dataSize);
        if (rc < 0) throw ixr_failure("ixr_get_data_from_dict_struct", rc);
        return data;
    }

private:
    static void check_type(uint8_t t) {
        if (t < ColumnType::INT || t > ColumnType::STRING) throw std::runtime_error("Unsupported data type");  // :166
    }
    int device_;
};

// The slice of VersionAdapter this path sits behind (VersionAdapter.java:42-48; VersionAdapter_VLT.java:147-157,220-234).
// Versions below 7 are Brighthouse-compressed packs, which this library does not decode.
class VersionAdapter {
public:
    explicit VersionAdapter(int device = 0) : vlt_(device) {}
    bool isCompress(int version) const { return version >= 0; }
    ByteSlice compressPack(int version, uint8_t dataType, bool isIndexed, DataPackNode& dpn, const ByteSlice& data) {
        return pc(version).compress(dataType, isIndexed, dpn, data);
    }
    ByteSlice decompressPack(int version, uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) {
        return pc(version).decompress(dataType, dpn, cmpData);
    }
    ByteSlice getDictStruct(int version, uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) {
        return pc(version).getDictStruct(dataType, dpn, cmpData);
    }
    ByteSlice getDataFromDictStruct(int version, uint8_t dataType, const DataPackNode& dpn, const ByteSlice& dictStruct) {
        return pc(version).getDataFromDictStruct(dataType, dpn, dictStruct);
    }

private:
    PackCompressor& pc(int version) {
        if (version != Version::VERSION_7_ID && version != Version::VERSION_8_ID)
            throw std::runtime_error("segment version " + std::to_string(version) + " is not a VLT version (BH-compressed packs are not supported)");
        return vlt_;
    }
    PackCompressor_VLT_GPU vlt_;
};

// NumType.floatToLong / doubleToLong (NumType.java:93-111): the uniform 64-bit form of a value
inline int64_t floatToLong(float f) { double d = (double)f; int64_t v; std::memcpy(&v, &d, 8); return v; }
inline int64_t doubleToLong(double d) { int64_t v; std::memcpy(&v, &d, 8); return v; }

// DataPack: decoded data + compressed data + (optional) DictStruct of one pack, with the reference's life cycle.
class DataPack {
public:
    DataPack(ByteSlice data, ByteSlice cmpData, DataPackNode dpn) : data_(std::move(data)), cmp_(std::move(cmpData)), dpn_(dpn) {}

    // --- creators (DataPackCreator_VLT.java:21-213): decoded layout + dpn.objCount / dataSize / min / max / sum
    static DataPack fromInts(const int32_t* v, int n) { return numbers(v, n, [](int32_t x) { return (int64_t)x; }, false); }
    static DataPack fromLongs(const int64_t* v, int n) { return numbers(v, n, [](int64_t x) { return x; }, false); }
    static DataPack fromFloats(const float* v, int n) { return numbers(v, n, [](float x) { return floatToLong(x); }, true); }
    static DataPack fromDoubles(const double* v, int n) { return numbers(v, n, [](double x) { return doubleToLong(x); }, true); }
    static DataPack fromStrings(const std::vector<std::string>& v) {
        const int n = (int)v.size();
        check_count(n);
        size_t total = 0;
        for (auto& s : v) {
            if (s.size() > 65535) throw std::runtime_error("string too long");  // DataType.java:10
            total += s.size();
        }
        ByteSlice data = ByteSlice::allocateDirect(4 * ((size_t)n + 1) + total);  // StringsStruct.java:12-17
        int32_t off = 0;
        uint8_t* body = data.address() + 4 * ((size_t)n + 1);
        for (int i = 0; i < n; i++) {
            data.putInt(4 * (size_t)i, off);
            if (!v[i].empty()) std::memcpy(body + off, v[i].data(), v[i].size());
            off += (int32_t)v[i].size();
        }
        data.putInt(4 * (size_t)n, off);
        DataPackNode dpn;
        dpn.objCount = n;
        dpn.dataSize = (int32_t)data.size();
        return DataPack(std::move(data), ByteSlice(), dpn);
    }

    // --- life cycle (DataPack.java:90-186)
    const DataPackNode& dpn() const { return dpn_; }
    int objCount() const { return dpn_.objCount; }
    const ByteSlice& data() const { return data_; }
    const ByteSlice& cmpData() const { return cmp_; }
    void compress(VersionAdapter& va, int version, uint8_t dataType, bool isIndexed) {  // DataPack.java:150-163
        if (data_.empty() && dpn_.dataSize) throw std::runtime_error("no data to compress");
        cmp_ = va.compressPack(version, dataType, isIndexed, dpn_, data_);
    }
    void decompress(VersionAdapter& va, int version, uint8_t dataType) {  // DataPack.java:166-186
        if (cmp_.empty()) throw std::runtime_error("no compressed data");
        if (!dpn_.compress) {
            data_ = cmp_;  // :180-182
        } else if (dpn_.isDictEncoded()) {
            dict_ = va.getDictStruct(version, dataType, dpn_, cmp_);                  // :172-176
            data_ = va.getDataFromDictStruct(version, dataType, dpn_, dict_);
        } else {
            data_ = va.decompressPack(version, dataType, dpn_, cmp_);                 // :178
        }
    }
    const ByteSlice& dictStruct(VersionAdapter& va, int version, uint8_t dataType) {  // DataPack.java:90-100
        if (dict_.empty() && dpn_.isDictEncoded()) dict_ = va.getDictStruct(version, dataType, dpn_, cmp_);
        return dict_;
    }
    void free() { data_.free(); cmp_.free(); dict_.free(); }  // DataPack.java:114-136
    void dropData() { data_.free(); dict_.free(); }

    // --- accessors of the decoded form (DataPack.java:196-529; NumOp.java:9-23)
    int32_t intValueAt(int i) const { return data_.getInt(4 * (size_t)idx(i)); }
    int64_t longValueAt(int i) const { return data_.getLong(8 * (size_t)idx(i)); }
    float floatValueAt(int i) const { float f; std::memcpy(&f, data_.address() + 4 * (size_t)idx(i), 4); return f; }
    double doubleValueAt(int i) const { double d; std::memcpy(&d, data_.address() + 8 * (size_t)idx(i), 8); return d; }
    std::string_view stringValueAt(int i) const {
        idx(i);
        const int32_t a = data_.getInt(4 * (size_t)i), b = data_.getInt(4 * (size_t)(i + 1));
        return std::string_view((const char*)data_.address() + 4 * ((size_t)dpn_.objCount + 1) + a, (size_t)(b - a));
    }
    // DataPack.uniformValAt (:196-214), version >= 1: ints and longs as they are, float / double as the raw bits of the
    // value widened to double
    int64_t uniformValAt(int i, uint8_t dataType) const {
        switch (dataType) {
            case ColumnType::INT: return intValueAt(i);
            case ColumnType::LONG: return longValueAt(i);
            case ColumnType::FLOAT: return floatToLong(floatValueAt(i));
            case ColumnType::DOUBLE: return doubleToLong(doubleValueAt(i));
            default: throw std::runtime_error("unsupported type " + std::to_string(dataType));
        }
    }
    // DataPack.rawValueAt(index, BytePiece) (:531-558): address and length of a string row, no copy
    struct BytePiece { const uint8_t* addr = nullptr; int len = 0; };
    void rawValueAt(int i, BytePiece& bytes) const {
        const std::string_view v = stringValueAt(i);
        bytes.addr = (const uint8_t*)v.data();
        bytes.len = (int)v.size();
    }
    // DataPack.foreach(start, count, XxxSetter) (:257-272, 308-323, 359-374, 410-425, 628-666): setter(index, value)
    template <typename F> void foreachInt(int start, int count, F setter) const { range(start, count); for (int i = start; i < start + count; i++) setter(i, intValueAt(i)); }
    template <typename F> void foreachLong(int start, int count, F setter) const { range(start, count); for (int i = start; i < start + count; i++) setter(i, longValueAt(i)); }
    template <typename F> void foreachFloat(int start, int count, F setter) const { range(start, count); for (int i = start; i < start + count; i++) setter(i, floatValueAt(i)); }
    template <typename F> void foreachDouble(int start, int count, F setter) const { range(start, count); for (int i = start; i < start + count; i++) setter(i, doubleValueAt(i)); }
    template <typename F> void foreachBytes(int start, int count, F setter) const {
        range(start, count);
        BytePiece bp;
        for (int i = start; i < start + count; i++) { rawValueAt(i, bp); setter(i, bp); }
    }
    // DataPack.foreach(BitMap position, XxxSetter) (:273-306 and siblings): the set rows below objCount, in order; returns
    // how many.  `words` is a row bitmap as the filters produce it (row r = bit r & 63 of word r >> 6); nullptr stands for
    // BitMap.ALL.  `at(i)` is one of the xxxValueAt accessors.
    template <typename At, typename F> int foreachAt(const uint64_t* words, At at, F setter) const {
        const int n = dpn_.objCount;
        if (!words) { for (int i = 0; i < n; i++) setter(i, at(i)); return n; }
        int count = 0;
        for (int w = 0; w * 64 < n; w++) {
            uint64_t bits = words[w];
            while (bits) {
                const int i = w * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                if (i >= n) return count;
                setter(i, at(i));
                count++;
            }
        }
        return count;
    }

private:
    static void check_count(int n) {
        if (n <= 0 || n > MAX_COUNT) throw std::runtime_error("row count out of range");
    }
    int idx(int i) const {
        if (i < 0 || i >= dpn_.objCount) throw std::out_of_range("row index");
        return i;
    }
    void range(int start, int count) const {
        if (start < 0 || count < 0 || start + count > dpn_.objCount) throw std::out_of_range("row range");
    }
    template <typename T, typename F> static DataPack numbers(const T* v, int n, F uniform, bool is_fp) {
        check_count(n);
        ByteSlice data = ByteSlice::copyOf(v, sizeof(T) * (size_t)n);
        DataPackNode dpn;
        dpn.objCount = n;
        dpn.dataSize = (int32_t)data.size();
        T mn = v[0], mx = v[0];
        if (is_fp) {
            // Math.min/max semantics (NaN wins, -0.0 < 0.0) are not needed for the decode path; plain comparisons suffice
            // for the statistics of NaN-free inputs.  float columns accumulate in float, double columns in double (:94,:131)
            T s = 0;
            for (int i = 0; i < n; i++) { mn = std::min(mn, v[i]); mx = std::max(mx, v[i]); s += v[i]; }
            dpn.sum = doubleToLong((double)s);
        } else {
            uint64_t s = 0;  // wrapping like Java's long
            for (int i = 0; i < n; i++) { mn = std::min(mn, v[i]); mx = std::max(mx, v[i]); s += (uint64_t)(int64_t)v[i]; }
            dpn.sum = (int64_t)s;
        }
        dpn.minValue = uniform(mn);
        dpn.maxValue = uniform(mx);
        return DataPack(std::move(data), ByteSlice(), dpn);
    }
    ByteSlice data_, cmp_, dict_;
    DataPackNode dpn_;
};

// Batch decode of many packs: the wrapper a CachedSegment-like prefetcher uses (CachedSegment.java:121-186 memoises
// pack(packId) per column; this decodes all requested packs of a scan in one plan).  Packs are staged in one pinned input
// arena, decoded on the device, returned in one pinned output arena; decoded(i) is a view valid until the batch dies.
class PackBatch {
public:
    explicit PackBatch(int device = 0) : device_(device) {}
    PackBatch(const PackBatch&) = delete;
    PackBatch& operator=(const PackBatch&) = delete;
    ~PackBatch() { release(); }

    int add(uint8_t dataType, const DataPackNode& dpn, const ByteSlice& cmpData) {
        if ((size_t)dpn.packSize != cmpData.size()) throw std::runtime_error("cmpData size != dpn.packSize");
        descs_.push_back(dpn.desc(dataType, in_bytes_, out_bytes_));
        srcs_.push_back(cmpData);
        in_bytes_ += ((uint64_t)dpn.packSize + 15) & ~15ull;
        out_bytes_ += ((uint64_t)dpn.dataSize + 15) & ~15ull;
        return (int)descs_.size() - 1;
    }
    // H2D, decode, D2H.  Throws if any pack fails; status(i) then tells which.
    void run() {
        release();
        const int n = (int)descs_.size();
        if (n == 0) return;
        plan_ = ixr_plan_create(device_, descs_.data(), n, 0);
        if (!plan_) throw ixr_failure("ixr_plan_create", -1);
        const uint64_t in_need = ixr_plan_in_bytes(plan_), out_need = std::max<uint64_t>(ixr_plan_out_bytes(plan_), 16);
        h_in_ = (uint8_t*)ixr_host_alloc_pinned(in_need);
        h_out_ = (uint8_t*)ixr_host_alloc_pinned(out_need);
        d_in_ = ixr_dev_alloc(device_, in_need);
        d_out_ = ixr_dev_alloc(device_, out_need);
        if (!h_in_ || !h_out_ || !d_in_ || !d_out_) throw ixr_failure("arena allocation", IXR_ERR_OUT_OF_MEMORY);
        std::memset(h_in_, 0, in_need);
        for (int i = 0; i < n; i++) std::memcpy(h_in_ + descs_[i].in_off, srcs_[i].address(), srcs_[i].size());
        int32_t rc = ixr_memcpy_h2d(device_, d_in_, h_in_, in_need);
        if (rc) throw ixr_failure("ixr_memcpy_h2d", rc);
        rc = ixr_plan_run(plan_, d_in_, in_need, d_out_, out_need, nullptr);
        if (rc) throw ixr_failure("ixr_plan_run", rc);
        status_.assign((size_t)n, 0);
        const int bad = ixr_plan_status(plan_, status_.data(), nullptr);
        rc = ixr_memcpy_d2h(device_, h_out_, d_out_, out_need);
        if (rc) throw ixr_failure("ixr_memcpy_d2h", rc);
        if (bad) throw std::runtime_error(std::to_string(bad) + " pack(s) failed to decode (consumed != packSize)");
    }
    int size() const { return (int)descs_.size(); }
    int32_t status(int i) const { return status_.at((size_t)i); }
    ByteSlice decoded(int i) const {
        if (!h_out_) throw std::runtime_error("run() first");
        return ByteSlice::fromAddress(h_out_ + descs_.at((size_t)i).out_off, (size_t)descs_[(size_t)i].data_size);
    }

private:
    void release() {
        if (plan_) ixr_plan_destroy(plan_);
        if (d_in_) ixr_dev_free(device_, d_in_);
        if (d_out_) ixr_dev_free(device_, d_out_);
        if (h_in_) ixr_host_free_pinned(h_in_);
        if (h_out_) ixr_host_free_pinned(h_out_);
        plan_ = nullptr; d_in_ = d_out_ = nullptr; h_in_ = h_out_ = nullptr;
    }
    int device_;
    std::vector<ixr_pack_desc> descs_;
    std::vector<ByteSlice> srcs_;
    std::vector<int32_t> status_;
    uint64_t in_bytes_ = 0, out_bytes_ = 0;
    ixr_plan* plan_ = nullptr;
    void *d_in_ = nullptr, *d_out_ = nullptr;
    uint8_t *h_in_ = nullptr, *h_out_ = nullptr;
};

}  // namespace indexr
#endif  // IXR_HOST_HPP
