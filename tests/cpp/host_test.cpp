// Harness for the C++ host layer (indexr_amd/host): reproduces the reference's callers of the decode path the way its own
// tests do (DataPackTest.java:30-65,106-169; DPSegmentTest / StorageColumn.pack loop), against libixrgpu.
//   host_test --cpu <tmpdir>   everything that needs no device: geometry, DPN records, creators, host compress,
//                              segment file write/read, CPU scan of the file with the oracle
//   host_test --gpu <tmpdir>   the decode path on the device: DataPack life cycle, DictStruct, batch, segment scan, errors
// The CPU oracle (oracle/) is linked here as the checker only.
#include <cinttypes>
#include <cstdio>
#include <limits>

#include "ixr_segment_file.hpp"
#include "ixr_synth.h"
extern "C" {
#include "ixr_oracle.h"
}

using namespace indexr;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond) do { g_checks++; if (!(cond)) { g_fail++; std::fprintf(stderr, "FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)
#define CHECK_THROWS(expr, needle) do { g_checks++; bool ok_ = false; try { (void)(expr); } catch (const std::exception& e_) { ok_ = std::string(e_.what()).find(needle) != std::string::npos; \
    if (!ok_) std::fprintf(stderr, "  wrong message: %s\n", e_.what()); } if (!ok_) { g_fail++; std::fprintf(stderr, "FAIL %s:%d  expected throw containing \"%s\"\n", __FILE__, __LINE__, needle); } } while (0)

static const int VERSION = Version::LATEST_ID;

// ---------------------------------------------------------------------------------------------------------------- helpers
static ByteSlice oracle_decode(uint8_t dt, const DataPackNode& dpn, const ByteSlice& cmp) {
    ByteSlice out = ByteSlice::allocateDirect((size_t)dpn.dataSize);
    if (!dpn.compress) { std::memcpy(out.address(), cmp.address(), cmp.size()); return out; }
    const int consumed = ixo_decompress_pack(dt, dpn.encodeType, dpn.objCount, cmp.address(), (int)cmp.size(), out.address(), dpn.dataSize);
    CHECK(consumed == dpn.packSize);
    return out;
}
static bool same(const ByteSlice& a, const ByteSlice& b) {
    return a.size() == b.size() && (a.size() == 0 || std::memcmp(a.address(), b.address(), a.size()) == 0);
}

struct Config1 {  // the 1 M-row mixed segment: INT, BIGINT, DOUBLE, VARCHAR
    static constexpr int64_t kRows = 1000000;
    std::vector<IntegratedSegmentWriter::Column> cols;
    std::vector<std::vector<ByteSlice>> source;  // decoded bytes per column and pack
};
static Config1 build_config1(VersionAdapter& va) {
    Config1 c;
    const int kinds[4] = {IXR_COL_I_CAT, IXR_COL_L_TS, IXR_COL_D_CARD, IXR_COL_S_URL};
    const int32_t sql[4] = {SQLType::INT, SQLType::BIGINT, SQLType::DOUBLE, SQLType::VARCHAR};
    const char* names[4] = {"i_cat", "l_ts", "d_card", "s_url"};
    const int packs = rowCountToPackCount(Config1::kRows);
    c.source.resize(4);
    for (int ci = 0; ci < 4; ci++) {
        IntegratedSegmentWriter::Column col{names[ci], sql[ci], false, {}};
        const uint8_t dt = SQLType::dataType(sql[ci]);
        for (int k = 0; k < packs; k++) {
            const int rows = packRowCount(Config1::kRows, k);
            std::vector<uint8_t> buf((size_t)rows * 100 + 64);
            const int32_t sz = ixr_synth_column(kinds[ci], 0xC0FFEE, (uint32_t)k, rows, buf.data(), (int32_t)buf.size());
            if (sz < 0) throw std::runtime_error("ixr_synth_column failed");
            DataPackNode dpn;
            dpn.objCount = rows;
            dpn.dataSize = sz;
            DataPack p(ByteSlice::copyOf(buf.data(), (size_t)sz), ByteSlice(), dpn);
            p.compress(va, VERSION, dt, false);
            c.source[(size_t)ci].push_back(p.data());
            col.packs.push_back(std::move(p));
        }
        c.cols.push_back(std::move(col));
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------------------- CPU part
static void test_geometry() {
    CHECK(packRowCount(65537, 1) == 1);           // DataPackTest.java:30-33
    CHECK(packRowCount(65536, 0) == 65536);
    CHECK(rowCountToPackCount(65536) == 1 && rowCountToPackCount(65537) == 2 && rowCountToPackCount(0) == 0);
    CHECK(rowCountToPackCount(1000000) == 16 && packRowCount(1000000, 15) == 16960);
    CHECK(DataPackNode::SIZE == 73);
}
static void test_dpn_record() {
    DataPackNode d;
    d.objCount = 65536; d.compress = true; d.packAddr = 0x1122334455LL; d.packSize = 777; d.indexAddr = 5; d.indexSize = 6;
    d.extIndexAddr = 7; d.extIndexSize = 8; d.minValue = -9; d.maxValue = std::numeric_limits<int64_t>::max();
    d.encodeType = CodecType::DICT_COMPRESS; d.dataSize = 262144; d.sum = -1;
    uint8_t rec[DataPackNode::SIZE];
    d.write(rec);
    // field offsets of the record: objCount@0 compress@4 packAddr@5 packSize@13 ... encodeType@57 dataSize@61 sum@65
    int32_t v;
    std::memcpy(&v, rec + 13, 4); CHECK(v == 777);
    std::memcpy(&v, rec + 57, 4); CHECK(v == CodecType::DICT_COMPRESS);
    std::memcpy(&v, rec + 61, 4); CHECK(v == 262144);
    CHECK(rec[4] == 1);
    const DataPackNode e = DataPackNode::from(rec);
    CHECK(e.objCount == d.objCount && e.compress && e.packAddr == d.packAddr && e.packSize == d.packSize && e.indexAddr == 5 &&
          e.indexSize == 6 && e.extIndexAddr == 7 && e.extIndexSize == 8 && e.minValue == -9 && e.maxValue == d.maxValue &&
          e.encodeType == d.encodeType && e.dataSize == d.dataSize && e.sum == -1 && e.isDictEncoded());
    const ixr_pack_desc pd = e.desc(ColumnType::INT, 32, 64);
    CHECK(pd.pack_size == 777 && pd.data_size == 262144 && pd.obj_count == 65536 && pd.encode_type == IXR_ENC_DICT_COMPRESS && pd.in_off == 32);
    DataPackNode raw = e; raw.compress = false;
    CHECK(raw.desc(ColumnType::INT, 0, 0).encode_type == IXR_ENC_RAW);
}
static void test_creators_and_compress(VersionAdapter& va) {
    const int32_t iv[] = {1, 2, 3, 5, 6, 7, 8, 9, -1111, 333, 0};
    DataPack p = DataPack::fromInts(iv, 11);
    CHECK(p.dpn().objCount == 11 && p.dpn().dataSize == 44 && p.dpn().minValue == -1111 && p.dpn().maxValue == 333 && p.dpn().sum == -737);
    CHECK(p.intValueAt(8) == -1111);
    p.compress(va, VERSION, ColumnType::INT, false);
    std::vector<uint8_t> ob(4096);
    int et = 0;
    const int osz = ixo_compress_pack(ColumnType::INT, 0, 11, p.data().address(), 44, ob.data(), (int)ob.size(), &et);
    CHECK(osz == p.dpn().packSize && et == p.dpn().encodeType && std::memcmp(ob.data(), p.cmpData().address(), (size_t)osz) == 0);
    const float fv[] = {1.1f, 1.1f};
    DataPack pf = DataPack::fromFloats(fv, 2);
    CHECK(pf.dpn().minValue == floatToLong(1.1f) && pf.floatValueAt(1) == 1.1f);
    DataPack ps = DataPack::fromStrings({"qwfwrfwe", "121343%$#22342342", "2211", std::string("sdfdsgkwkek00-\t\t\5\3\1\bsdaasds"), ""});
    CHECK(ps.dpn().dataSize == 4 * 6 + 8 + 17 + 4 + 27 && ps.stringValueAt(2) == "2211" && ps.stringValueAt(4).empty());
    CHECK_THROWS(DataPack::fromInts(iv, 0), "row count");
    CHECK_THROWS(p.intValueAt(11), "row index");
}
static void test_segment_file_cpu(VersionAdapter& va, const std::string& dir) {
    Config1 c = build_config1(va);
    const std::string path = dir + "/config1.ixrseg";
    IntegratedSegmentWriter::write(path, Config1::kRows, c.cols);
    IntegratedSegmentFile f(path);
    const SegmentMetaV1& m = f.meta();
    CHECK(m.version == 8 && m.mode == kModeVLT && m.rowCount == Config1::kRows && m.columnCount == 4 && m.sectionOffset == 8);
    CHECK(f.columnId("d_card") == 2 && f.columnId("nope") == -1);
    struct stat st;
    CHECK(::stat(path.c_str(), &st) == 0 && (int64_t)st.st_size == m.sectionOffset + m.sectionSize);
    // the CPU reader: StorageColumn.pack(packId) = loadPack + decompress, here with the oracle's decode
    for (int ci = 0; ci < 4; ci++) {
        const uint8_t dt = SQLType::dataType(m.columnInfos[(size_t)ci].sqlType);
        const std::vector<DataPackNode> dpns = f.loadDPNs(ci);
        CHECK((int)dpns.size() == 16);
        for (size_t k = 0; k < dpns.size(); k++) {
            const ByteSlice cmp = f.loadPack(ci, dpns[k]);
            CHECK(same(cmp, c.cols[(size_t)ci].packs[k].cmpData()));
            CHECK(same(oracle_decode(dt, dpns[k], cmp), c.source[(size_t)ci][k]));
        }
    }
    // a file that is not a segment
    const std::string junk = dir + "/junk.bin";
    FILE* j = std::fopen(junk.c_str(), "wb");
    std::fputs("IXRSEG07 not really", j);
    std::fclose(j);
    CHECK_THROWS(IntegratedSegmentFile(junk), "IXRSEG08");
}
static void test_no_device_fails_loudly(VersionAdapter& va) {
    if (ixr_device_count() > 0) return;
    const int32_t iv[] = {10, 10, 10, 10};
    DataPack p = DataPack::fromInts(iv, 4);
    p.compress(va, VERSION, ColumnType::INT, false);
    p.dropData();
    CHECK_THROWS(p.decompress(va, VERSION, ColumnType::INT), "failed");  // never a silent CPU decode
}

// ---------------------------------------------------------------------------------------------------------------- GPU part
template <typename T, typename Get> static void cmpNumber(const DataPack& p, const T* vals, int n, Get get) {
    CHECK(p.objCount() == n);
    for (int i = 0; i < n; i++) {
        const T got = get(p, i);
        CHECK(std::memcmp(&got, &vals[i], sizeof(T)) == 0);
    }
}
// DataPackTest.testNumber / testRaw (:35-65): compress, serialise dpn + pack, rebuild from bytes, decompress, compare
static DataPack through_bytes(VersionAdapter& va, DataPack& p, uint8_t dt, bool isIndexed) {
    p.compress(va, VERSION, dt, isIndexed);
    uint8_t rec[DataPackNode::SIZE];
    p.dpn().write(rec);
    std::vector<uint8_t> wire(p.cmpData().address(), p.cmpData().address() + p.cmpData().size());
    DataPack q(ByteSlice(), ByteSlice::copyOf(wire.data(), wire.size()), DataPackNode::from(rec));
    q.decompress(va, VERSION, dt);
    CHECK(same(q.data(), p.data()));
    return q;
}
static void test_pack_literals(VersionAdapter& va, bool isIndexed) {
    const int32_t iv[] = {1, 2, 3, 5, 6, 7, 8, 9, -1111, 333, 0};
    DataPack p = DataPack::fromInts(iv, 11);
    cmpNumber(through_bytes(va, p, ColumnType::INT, isIndexed), iv, 11, [](const DataPack& d, int i) { return d.intValueAt(i); });
    const int32_t iv1[] = {0};
    p = DataPack::fromInts(iv1, 1);
    cmpNumber(through_bytes(va, p, ColumnType::INT, isIndexed), iv1, 1, [](const DataPack& d, int i) { return d.intValueAt(i); });
    const int32_t iv2[] = {10, 10, 10, 10};
    p = DataPack::fromInts(iv2, 4);
    cmpNumber(through_bytes(va, p, ColumnType::INT, isIndexed), iv2, 4, [](const DataPack& d, int i) { return d.intValueAt(i); });
    const int64_t lv[] = {1, std::numeric_limits<int64_t>::max(), 3, std::numeric_limits<int64_t>::min(), 6, 7, 8, 9, 1111, 333, 0};
    p = DataPack::fromLongs(lv, 11);
    cmpNumber(through_bytes(va, p, ColumnType::LONG, isIndexed), lv, 11, [](const DataPack& d, int i) { return d.longValueAt(i); });
    const float fv[] = {1, 2, 3, 5, 6, 7, 8, 9, 1111.4666F, 333, 0};
    p = DataPack::fromFloats(fv, 11);
    cmpNumber(through_bytes(va, p, ColumnType::FLOAT, isIndexed), fv, 11, [](const DataPack& d, int i) { return d.floatValueAt(i); });
    const float fv2[] = {1.1f, 1.1f};
    p = DataPack::fromFloats(fv2, 2);
    cmpNumber(through_bytes(va, p, ColumnType::FLOAT, isIndexed), fv2, 2, [](const DataPack& d, int i) { return d.floatValueAt(i); });
    const double dv[] = {1, 2, 3, 5, std::numeric_limits<double>::denorm_min(), 7, std::numeric_limits<double>::max(), 9, 1111, -333.4444444, 0};
    p = DataPack::fromDoubles(dv, 11);
    cmpNumber(through_bytes(va, p, ColumnType::DOUBLE, isIndexed), dv, 11, [](const DataPack& d, int i) { return d.doubleValueAt(i); });
    const std::vector<std::string> sv = {"qwfwrfwe", "121343%$#22342342", "2211", std::string("sdfdsgkwkek00-\t\t\5\3\1\bsdaasds"), ""};
    DataPack ps = DataPack::fromStrings(sv);
    DataPack qs = through_bytes(va, ps, ColumnType::STRING, isIndexed);
    for (size_t i = 0; i < sv.size(); i++) CHECK(qs.stringValueAt((int)i) == sv[i]);
    const std::vector<std::string> sv2 = {"", "", "", ""};
    ps = DataPack::fromStrings(sv2);
    qs = through_bytes(va, ps, ColumnType::STRING, isIndexed);
    for (size_t i = 0; i < sv2.size(); i++) CHECK(qs.stringValueAt((int)i).empty());
    // test_str (:188-196)
    const std::vector<std::string> sv3 = {"121343%$#22342342", "#$%#%242rwef23r23dfsdf", "", "2211", std::string("sdfdsgkwkek00-\t\t\5\3\1\bsdaasds"), "",
        "323weewew", "[][][][[]]", "windows", "mac", "linux", "android", "windows", "mac", "linux", "android", "windows", "mac", "linux", "android", "qwfwrfwe"};
    ps = DataPack::fromStrings(sv3);
    qs = through_bytes(va, ps, ColumnType::STRING, isIndexed);
    for (size_t i = 0; i < sv3.size(); i++) CHECK(qs.stringValueAt((int)i) == sv3[i]);
}
static void test_dict_struct_and_errors(VersionAdapter& va) {
    std::vector<int32_t> v(5000);
    for (size_t i = 0; i < v.size(); i++) v[i] = (int32_t)((i * 2654435761u) % 97) * 13 - 300;
    DataPack p = DataPack::fromInts(v.data(), (int)v.size());
    p.compress(va, VERSION, ColumnType::INT, true);
    CHECK(p.dpn().isDictEncoded());
    const ByteSlice& ds = p.dictStruct(va, VERSION, ColumnType::INT);
    std::vector<uint8_t> ods(ds.size());
    const int rc = ixo_get_dict_struct(ColumnType::INT, p.objCount(), p.cmpData().address(), (int)p.cmpData().size(), ods.data(), (int)ods.size());
    CHECK(rc == p.dpn().packSize && std::memcmp(ods.data(), ds.address(), ds.size()) == 0);
    CHECK(ds.getInt(0) == 97);  // entryCount (DictStruct.java:14-55)
    // error behaviour of PackCompressor_VLT.decompress
    DataPackNode bad = p.dpn();
    bad.encodeType = CodecType::RLE;
    CHECK_THROWS(va.decompressPack(VERSION, ColumnType::INT, bad, p.cmpData()), "Illegal encode type");
    CHECK_THROWS(va.decompressPack(VERSION, 9, p.dpn(), p.cmpData()), "Unsupported data type");
    CHECK_THROWS(va.decompressPack(6, ColumnType::INT, p.dpn(), p.cmpData()), "not a VLT version");
    ByteSlice cut = ByteSlice::copyOf(p.cmpData().address(), p.cmpData().size() / 2);  // truncated pack
    DataPackNode half = p.dpn();
    half.packSize = (int32_t)cut.size();
    CHECK_THROWS(va.decompressPack(VERSION, ColumnType::INT, half, cut), "ixr_decode_pack failed");
}
static void test_segment_scan_gpu(VersionAdapter& va, const std::string& dir) {
    Config1 c = build_config1(va);
    const std::string path = dir + "/config1_gpu.ixrseg";
    IntegratedSegmentWriter::write(path, Config1::kRows, c.cols);
    IntegratedSegmentFile f(path);
    SegmentBatch b = f.stage({0, 1, 2, 3});
    CHECK(b.descs.size() == 64);
    ixr_plan* plan = ixr_plan_create(0, b.descs.data(), (int32_t)b.descs.size(), 0);
    CHECK(plan != nullptr);
    if (!plan) return;
    void* d_in = ixr_dev_alloc(0, b.in_bytes);
    void* d_out = ixr_dev_alloc(0, b.out_bytes);
    uint8_t* h_out = (uint8_t*)ixr_host_alloc_pinned(b.out_bytes);
    CHECK(d_in && d_out && h_out);
    CHECK(ixr_memcpy_h2d(0, d_in, b.arena, b.in_bytes) == 0);
    CHECK(ixr_plan_run(plan, d_in, b.in_bytes, d_out, b.out_bytes, nullptr) == 0);
    std::vector<int32_t> status(b.descs.size());
    CHECK(ixr_plan_status(plan, status.data(), nullptr) == 0);
    CHECK(ixr_memcpy_d2h(0, h_out, d_out, b.out_bytes) == 0);
    uint64_t rows = 0;
    for (size_t i = 0; i < b.descs.size(); i++) {
        const ixr_pack_desc& d = b.descs[i];
        CHECK(status[i] == d.pack_size);  // consumed == packSize
        const ByteSlice& src = c.source[(size_t)b.column_of[i]][(size_t)b.pack_id_of[i]];
        CHECK((size_t)d.data_size == src.size() && std::memcmp(h_out + d.out_off, src.address(), src.size()) == 0);
        rows += (uint64_t)d.obj_count;
    }
    CHECK(rows == 4ull * Config1::kRows);
    ixr_plan_destroy(plan);
    ixr_dev_free(0, d_in);
    ixr_dev_free(0, d_out);
    ixr_host_free_pinned(h_out);
    // the same through the per-pack wrapper and the PackBatch convenience class, on one column
    PackBatch pb(0);
    const std::vector<DataPackNode> dpns = f.loadDPNs(3);
    std::vector<ByteSlice> keep;
    for (auto& d : dpns) { keep.push_back(f.loadPack(3, d)); pb.add(ColumnType::STRING, d, keep.back()); }
    pb.run();
    for (int k = 0; k < pb.size(); k++) CHECK(pb.status(k) == dpns[(size_t)k].packSize && same(pb.decoded(k), c.source[3][(size_t)k]));
    DataPack one(ByteSlice(), keep[15], dpns[15]);
    one.decompress(va, VERSION, ColumnType::STRING);
    CHECK(same(one.data(), c.source[3][15]) && one.objCount() == 16960);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: host_test --cpu|--gpu <tmpdir>\n"); return 2; }
    const std::string mode = argv[1], dir = argv[2];
    try {
        VersionAdapter va(0);
        if (mode == "--cpu") {
            test_geometry();
            test_dpn_record();
            test_creators_and_compress(va);
            test_segment_file_cpu(va, dir);
            test_no_device_fails_loudly(va);
        } else if (mode == "--gpu") {
            if (ixr_init(0) != 0) { std::fprintf(stderr, "no device: %s\n", ixr_last_error()); return 3; }
            test_pack_literals(va, false);
            test_pack_literals(va, true);
            test_dict_struct_and_errors(va);
            test_segment_scan_gpu(va, dir);
        } else {
            return 2;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "unexpected exception: %s\n", e.what());
        return 1;
    }
    std::printf("%s: %d checks, %d failed\n", mode.c_str(), g_checks, g_fail);
    return g_fail ? 1 : 0;
}
