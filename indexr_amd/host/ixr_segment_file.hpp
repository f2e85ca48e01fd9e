// ixr_segment_file.hpp -- IntegratedSegment files (version 8, VLT mode) -> batch descriptors for the GPU decoder.
//
// This is the step immediately before the decode path: open a segment file, read its meta block and the DPN table of the
// projected columns, pread the compressed packs into one pinned arena and hand `ixr_pack_desc[]` to ixr_plan_create.
// Layout follows the reference (paths relative to the reference tree):
//   file      "IXRSEG08" | u32 metaSize | SegmentMetaV1 | per column: DPNs | rsIndex | extIndex | packs | outer indexes
//             indexr-segment/.../storage/itg/IntegrateV1.java:23-34 (layout), :38-151 (write), :153-165 (read)
//   meta      IntegrateV1.java:167-259 (SegmentMetaV1), :261-425 (ColumnMetaV1), itg/ColumnNodeMeta.java:25-33,
//             indexr-common/.../util/ObjectSaver.java:21-49 (u32 size prefix), storage/Version.java:31-62 (magic)
//   DPN table storage/StorageColumn.java:47-67 (packCount records of dpnSize bytes at dpnOffset)
//   packs     StorageColumn.java:69-80 + itg/IntegratedColumn.java:117-119 (dpn.packAddr is relative to packOffset)
// Everything is little-endian.  The writer below exists to make test inputs: it emits the same container with EMPTY index
// sections (indexSize = extIndexSize = outerIndexSize = 0), which is all a decode-only reader looks at.
#ifndef IXR_SEGMENT_FILE_HPP
#define IXR_SEGMENT_FILE_HPP
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>

#include "ixr_host.hpp"

namespace indexr {

// SQL type ids stored in the column meta and the pack data type each maps to
namespace SQLType {
enum : int32_t { INT = 1, BIGINT = 2, FLOAT = 3, DOUBLE = 4, VARCHAR = 5, DATE = 6, TIME = 7, DATETIME = 8 };
inline uint8_t dataType(int32_t sqlType) {
    switch (sqlType) {
        case INT: case TIME: return ColumnType::INT;
        case BIGINT: case DATE: case DATETIME: return ColumnType::LONG;
        case FLOAT: return ColumnType::FLOAT;
        case DOUBLE: return ColumnType::DOUBLE;
        case VARCHAR: return ColumnType::STRING;
        default: throw std::runtime_error("Unsupported sql type " + std::to_string(sqlType));
    }
}
}  // namespace SQLType

constexpr int32_t kModeVLT = 10001;  // VLTPlugin.java:9; persisted in the file, IntegrateV1.java:218-220
constexpr const char* kMagicV8 = "IXRSEG08";

struct ColumnMetaV1 {  // IntegrateV1.java:261-425
    std::string name;
    int32_t sqlType = 0;
    bool isIndexed = false;
    int64_t dpnOffset = 0, indexOffset = 0, extIndexOffset = 0, outerIndexOffset = 0, outerIndexSize = 0, packOffset = 0;
    int metaSize() const { return 4 + (int)name.size() + 4 + 1 + 8 * 6; }  // version >= 8, :349-351
};

struct SegmentMetaV1 {  // IntegrateV1.java:167-259
    int32_t version = Version::VERSION_8_ID;
    int64_t sectionOffset = 8, sectionSize = 0, rowCount = 0;
    int32_t columnCount = 0, mode = kModeVLT;
    std::vector<std::pair<int64_t, int64_t>> columnNodeInfos;  // (minNumValue, maxNumValue)
    std::vector<ColumnMetaV1> columnInfos;

    int metaSize() const {  // :178-199 (version >= 6)
        int s = 8 + 8 + 8 + 4 + 16 * (int)columnNodeInfos.size() + 8 + 4 + 4;
        for (auto& c : columnInfos) s += c.metaSize();
        return s;
    }
    int packCount() const { return rowCountToPackCount(rowCount); }
};

namespace detail {
struct Cursor {
    const uint8_t* p;
    size_t n, o = 0;
    template <typename T> T get() {
        if (o + sizeof(T) > n) throw std::runtime_error("segment meta truncated");
        T v;
        std::memcpy(&v, p + o, sizeof(T));
        o += sizeof(T);
        return v;
    }
    std::string str(size_t len) {
        if (o + len > n) throw std::runtime_error("segment meta truncated");
        std::string s((const char*)p + o, len);
        o += len;
        return s;
    }
};
struct Out {
    std::vector<uint8_t> b;
    template <typename T> void put(T v) { const uint8_t* q = (const uint8_t*)&v; b.insert(b.end(), q, q + sizeof(T)); }
    void bytes(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
};
inline void pread_full(int fd, void* dst, size_t n, int64_t off) {
    uint8_t* d = (uint8_t*)dst;
    while (n) {
        const ssize_t r = ::pread(fd, d, n, (off_t)off);
        if (r <= 0) throw std::runtime_error("segment file: short read");
        d += r; n -= (size_t)r; off += r;
    }
}
}  // namespace detail

// One compressed pack per DPN of every projected column, staged for the batch decoder.
struct SegmentBatch {
    std::vector<ixr_pack_desc> descs;  // column-major: all packs of projection[0], then projection[1], ...
    std::vector<int> column_of;        // index into the projection, per desc
    std::vector<int> pack_id_of;
    uint8_t* arena = nullptr;          // pinned host memory (ixr_host_alloc_pinned), in_bytes long
    uint64_t in_bytes = 0, out_bytes = 0;
    SegmentBatch() = default;
    SegmentBatch(const SegmentBatch&) = delete;
    SegmentBatch& operator=(const SegmentBatch&) = delete;
    SegmentBatch(SegmentBatch&& o) noexcept { *this = std::move(o); }
    SegmentBatch& operator=(SegmentBatch&& o) noexcept {
        std::swap(descs, o.descs); std::swap(column_of, o.column_of); std::swap(pack_id_of, o.pack_id_of);
        std::swap(arena, o.arena); std::swap(in_bytes, o.in_bytes); std::swap(out_bytes, o.out_bytes);
        return *this;
    }
    ~SegmentBatch() { if (arena) ixr_host_free_pinned(arena); }
};

class IntegratedSegmentFile {
public:
    explicit IntegratedSegmentFile(const std::string& path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        try {
            char magic[8];
            detail::pread_full(fd_, magic, 8, 0);
            if (std::memcmp(magic, kMagicV8, 8) != 0) throw std::runtime_error("not an IXRSEG08 file (Version.check, Version.java:52-62)");
            uint32_t metaSize;
            detail::pread_full(fd_, &metaSize, 4, 8);  // ObjectSaver.load: u32 size, then the object
            if (metaSize < 44 || metaSize > (64u << 20)) throw std::runtime_error("implausible segment meta size");
            std::vector<uint8_t> buf(metaSize);
            detail::pread_full(fd_, buf.data(), metaSize, 12);
            detail::Cursor c{buf.data(), buf.size()};
            if (c.get<int64_t>() != -1) throw std::runtime_error("version 0 segment files are not supported");  // VERSION_FLAG :203-212
            meta_.version = c.get<int32_t>();
            if (meta_.version != Version::VERSION_8_ID) throw std::runtime_error("segment version " + std::to_string(meta_.version) + " != 8");
            meta_.sectionOffset = c.get<int64_t>();
            meta_.sectionSize = c.get<int64_t>();
            meta_.rowCount = c.get<int64_t>();
            meta_.columnCount = c.get<int32_t>();
            meta_.mode = c.get<int32_t>();
            if (meta_.columnCount < 0 || meta_.columnCount > 65536 || meta_.rowCount < 0) throw std::runtime_error("bad segment meta");
            for (int i = 0; i < meta_.columnCount; i++) {
                const int64_t mn = c.get<int64_t>(), mx = c.get<int64_t>();
                meta_.columnNodeInfos.emplace_back(mn, mx);
            }
            for (int i = 0; i < meta_.columnCount; i++) {
                ColumnMetaV1 m;
                const int32_t nameSize = c.get<int32_t>();
                if (nameSize < 0 || nameSize > 65536) throw std::runtime_error("bad column name size");
                m.name = c.str((size_t)nameSize);
                m.sqlType = c.get<int32_t>();
                m.isIndexed = c.get<uint8_t>() == 1;
                m.dpnOffset = c.get<int64_t>();
                m.indexOffset = c.get<int64_t>();
                m.extIndexOffset = c.get<int64_t>();
                m.outerIndexOffset = c.get<int64_t>();
                m.outerIndexSize = c.get<int64_t>();
                m.packOffset = c.get<int64_t>();
                meta_.columnInfos.push_back(std::move(m));
            }
            if (c.o != c.n) throw std::runtime_error("segment meta has trailing bytes");  // ObjectSaver.load asserts remaining()==0
        } catch (...) {
            ::close(fd_);
            throw;
        }
    }
    IntegratedSegmentFile(const IntegratedSegmentFile&) = delete;
    IntegratedSegmentFile& operator=(const IntegratedSegmentFile&) = delete;
    ~IntegratedSegmentFile() { if (fd_ >= 0) ::close(fd_); }

    const SegmentMetaV1& meta() const { return meta_; }
    bool isVLT() const { return meta_.mode == kModeVLT; }
    int columnId(const std::string& name) const {
        for (int i = 0; i < meta_.columnCount; i++) if (meta_.columnInfos[(size_t)i].name == name) return i;
        return -1;
    }
    // StorageColumn.loadDPNs (:47-67)
    std::vector<DataPackNode> loadDPNs(int colId) const {
        const ColumnMetaV1& m = meta_.columnInfos.at((size_t)colId);
        const int n = meta_.packCount();
        std::vector<uint8_t> raw((size_t)n * DataPackNode::SIZE);
        if (n) detail::pread_full(fd_, raw.data(), raw.size(), m.dpnOffset);
        std::vector<DataPackNode> dpns;
        for (int i = 0; i < n; i++) dpns.push_back(DataPackNode::from(raw.data() + (size_t)i * DataPackNode::SIZE));
        return dpns;
    }
    // StorageColumn.loadPack (:69-80): the pack in compressed form
    ByteSlice loadPack(int colId, const DataPackNode& dpn) const {
        ByteSlice b = ByteSlice::allocateDirect((size_t)dpn.packSize);
        if (dpn.packSize) detail::pread_full(fd_, b.address(), (size_t)dpn.packSize, meta_.columnInfos.at((size_t)colId).packOffset + dpn.packAddr);
        return b;
    }
    // All packs of the projected columns, read into one pinned arena; descriptors ready for ixr_plan_create.
    SegmentBatch stage(const std::vector<int>& projection) const {
        if (!isVLT()) throw std::runtime_error("segment mode " + std::to_string(meta_.mode) + " is not VLT (10001)");
        SegmentBatch b;
        uint64_t in_off = 0, out_off = 0;
        std::vector<int64_t> file_off;
        for (size_t pc = 0; pc < projection.size(); pc++) {
            const int colId = projection[pc];
            const ColumnMetaV1& m = meta_.columnInfos.at((size_t)colId);
            const uint8_t dt = SQLType::dataType(m.sqlType);
            const std::vector<DataPackNode> dpns = loadDPNs(colId);
            for (size_t k = 0; k < dpns.size(); k++) {
                const DataPackNode& d = dpns[k];
                if (d.objCount != packRowCount(meta_.rowCount, (int)k)) throw std::runtime_error("DPN objCount does not match the segment's row count");
                if (d.packSize < 0 || d.dataSize < 0 || d.packAddr < 0) throw std::runtime_error("bad DPN");
                b.descs.push_back(d.desc(dt, in_off, out_off));
                b.column_of.push_back((int)pc);
                b.pack_id_of.push_back((int)k);
                file_off.push_back(m.packOffset + d.packAddr);
                in_off += ((uint64_t)d.packSize + 15) & ~15ull;
                out_off += ((uint64_t)d.dataSize + 15) & ~15ull;
            }
        }
        b.in_bytes = in_off + IXR_IN_SLACK;
        b.out_bytes = out_off + 16;
        b.arena = (uint8_t*)ixr_host_alloc_pinned(b.in_bytes);
        if (!b.arena) throw ixr_failure("ixr_host_alloc_pinned", IXR_ERR_OUT_OF_MEMORY);
        std::memset(b.arena, 0, b.in_bytes);
        for (size_t i = 0; i < b.descs.size(); i++)
            if (b.descs[i].pack_size) detail::pread_full(fd_, b.arena + b.descs[i].in_off, (size_t)b.descs[i].pack_size, file_off[i]);
        return b;
    }

private:
    int fd_ = -1;
    SegmentMetaV1 meta_;
};

// Test-input writer: columns of already compressed packs -> an IXRSEG08 file with empty index sections.
class IntegratedSegmentWriter {
public:
    struct Column {
        std::string name;
        int32_t sqlType;
        bool isIndexed;
        std::vector<DataPack> packs;  // compressed (cmpData + dpn filled)
    };
    static void write(const std::string& path, int64_t rowCount, const std::vector<Column>& cols) {
        SegmentMetaV1 meta;
        meta.rowCount = rowCount;
        meta.columnCount = (int32_t)cols.size();
        const int packCount = meta.packCount();
        for (auto& c : cols) {
            if ((int)c.packs.size() != packCount) throw std::runtime_error("column " + c.name + ": pack count != rowCountToPackCount(rowCount)");
            int64_t mn = 0, mx = 0;
            for (size_t k = 0; k < c.packs.size(); k++) {
                const DataPackNode& d = c.packs[k].dpn();
                mn = k ? std::min(mn, d.minValue) : d.minValue;
                mx = k ? std::max(mx, d.maxValue) : d.maxValue;
            }
            meta.columnNodeInfos.emplace_back(mn, mx);
            ColumnMetaV1 m;
            m.name = c.name; m.sqlType = c.sqlType; m.isIndexed = c.isIndexed;
            meta.columnInfos.push_back(m);
        }
        // offsets (IntegrateV1.java:83-104): column data back to back after the meta block, outer indexes (empty) last
        int64_t off = meta.sectionOffset + 4 + meta.metaSize();
        std::vector<std::vector<DataPackNode>> dpns(cols.size());
        for (size_t ci = 0; ci < cols.size(); ci++) {
            ColumnMetaV1& m = meta.columnInfos[ci];
            int64_t packBytes = 0;
            for (auto& p : cols[ci].packs) {
                DataPackNode d = p.dpn();
                d.packAddr = packBytes;  // relative to packOffset
                d.indexAddr = 0; d.indexSize = 0; d.extIndexAddr = 0; d.extIndexSize = 0;
                packBytes += d.packSize;
                dpns[ci].push_back(d);
            }
            m.dpnOffset = off;
            m.indexOffset = m.dpnOffset + (int64_t)DataPackNode::SIZE * packCount;
            m.extIndexOffset = m.indexOffset;   // indexSize 0
            m.packOffset = m.extIndexOffset;    // extIndexSize 0
            off = m.packOffset + packBytes;
        }
        for (auto& m : meta.columnInfos) { m.outerIndexOffset = off; m.outerIndexSize = 0; }
        meta.sectionSize = off - meta.sectionOffset;

        detail::Out o;
        o.bytes(kMagicV8, 8);
        o.put<int32_t>(meta.metaSize());  // ObjectSaver.save
        o.put<int64_t>(-1);               // VERSION_FLAG
        o.put<int32_t>(meta.version);
        o.put<int64_t>(meta.sectionOffset);
        o.put<int64_t>(meta.sectionSize);
        o.put<int64_t>(meta.rowCount);
        o.put<int32_t>(meta.columnCount);
        o.put<int32_t>(meta.mode);
        for (auto& ni : meta.columnNodeInfos) { o.put<int64_t>(ni.first); o.put<int64_t>(ni.second); }
        for (auto& m : meta.columnInfos) {
            o.put<int32_t>((int32_t)m.name.size());
            o.bytes(m.name.data(), m.name.size());
            o.put<int32_t>(m.sqlType);
            o.put<uint8_t>(m.isIndexed ? 1 : 0);
            o.put<int64_t>(m.dpnOffset); o.put<int64_t>(m.indexOffset); o.put<int64_t>(m.extIndexOffset);
            o.put<int64_t>(m.outerIndexOffset); o.put<int64_t>(m.outerIndexSize); o.put<int64_t>(m.packOffset);
        }
        if ((int64_t)o.b.size() != meta.sectionOffset + 4 + meta.metaSize()) throw std::logic_error("meta size mismatch");
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + path);
        bool ok = std::fwrite(o.b.data(), 1, o.b.size(), f) == o.b.size();
        for (size_t ci = 0; ci < cols.size() && ok; ci++) {
            uint8_t rec[DataPackNode::SIZE];
            for (auto& d : dpns[ci]) { d.write(rec); ok = ok && std::fwrite(rec, 1, sizeof rec, f) == sizeof rec; }
            for (auto& p : cols[ci].packs) {
                const ByteSlice& c = p.cmpData();
                ok = ok && (c.size() == 0 || std::fwrite(c.address(), 1, c.size(), f) == c.size());
            }
        }
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) throw std::runtime_error("write failed: " + path);
    }
};

}  // namespace indexr
#endif  // IXR_SEGMENT_FILE_HPP
