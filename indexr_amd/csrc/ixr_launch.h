// ixr_launch.h -- host-callable launch wrappers of the kernels in ixr_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ixr_device.h"

#define IXR_PACK_MAX_ROWS_ 65536
#define IXR_PACK_TILES (IXR_PACK_MAX_ROWS_ / 1024)

namespace ixr {

struct LaunchCtx {
    hipStream_t stream;
    const DevPack* packs;  // device array, all packs of the plan
    PackMeta* metas;       // device array, all packs of the plan
    const uint8_t* in;     // input arena
    uint8_t* out;          // output arena
    uint8_t* scratch;      // scratch base of the current group
    int dict_only = 0;     // 1: build dictionaries only; packs whose decode writes straight to the output arena are skipped
    int slice = 0;         // 1: sorted numeric dictionaries whose rows take them in order are not materialised (slice path of k_dict_num)
    int wave_walk_from = 0;  // framing walk: packs per launch from which every stream has a one-wave workgroup of its own; 0 = default
};

// ---- the batch that allocates nothing: descriptors travel as a kernel argument, every working array lives in one
// caller-supplied block of device memory (the library's static scratch array), laid out as
//   [DevPack x kStaticMaxPacks][PackMeta x kStaticMaxPacks][5 work lists x kStaticMaxPacks ints] | index area | bulk area
constexpr int kStaticMaxPacks = 64;
constexpr uint64_t kStaticMetaOff = 8192, kStaticListOff = 16384, kStaticHeader = 32768;
enum StaticList { SL_ALL = 0, SL_N32, SL_N64, SL_STR, SL_RAW, SL_COUNT };
struct SmallDesc {  // same layout as ixr_pack_desc
    uint64_t in_off, out_off;
    int32_t pack_size, data_size, obj_count, encode_type, data_type, reserved;
};
struct StaticBatch {
    SmallDesc d[kStaticMaxPacks];
    uint64_t dict_cap;  // bytes reserved for every pack's dictionary; 0: what the pack's rows can hold (see pack_scratch_layout)
    int32_t n;
};
// Lays the batch out behind the header; fills `packs` (offsets relative to the block's start + kStaticHeader), the work lists
// (`lists[c * kStaticMaxPacks + k]`) and their lengths when the pointers are given.  Returns the bytes of the block the batch
// needs.  Runs on the host (bound check, grid sizes) and on the device (the records themselves) with the same result.
__host__ __device__ inline uint64_t static_layout(const SmallDesc* d, int n, uint64_t dict_cap, DevPack* packs, int* lists, int* counts) {
    uint64_t index_total = 0;
    for (int i = 0; i < n; i++) {
        DevPack p;
        p.obj_count = d[i].obj_count; p.data_size = d[i].data_size; p.encode_type = d[i].encode_type;
        p.width = (d[i].data_type == 1 || d[i].data_type == 3) ? 32 : (d[i].data_type == 5 ? 0 : 64);
        index_total += pack_scratch_layout(p, dict_cap).index;
    }
    index_total = scratch_align16(index_total) + 64;  // the bulk area never starts at offset 0 (k_str_copy may read 15 bytes before a dictionary)
    uint64_t io = 0, bo = index_total;
    for (int c = 0; counts && c < SL_COUNT; c++) counts[c] = 0;
    for (int i = 0; i < n; i++) {
        DevPack p;
        p.in_off = d[i].in_off; p.out_off = d[i].out_off; p.pack_size = d[i].pack_size; p.data_size = d[i].data_size;
        p.obj_count = d[i].obj_count; p.encode_type = d[i].encode_type; p.data_type = d[i].data_type;
        p.width = (d[i].data_type == 1 || d[i].data_type == 3) ? 32 : (d[i].data_type == 5 ? 0 : 64);
        p.ds_off = 0;
        const ScratchNeed need = pack_scratch_layout(p, dict_cap);
        if (p.encode_type != 0) {
            p.s_idx[0] += io; p.s_idx[1] += io; p.s_restart += io;
            p.s_dict += bo; p.s_lens += bo; p.s_rows += bo;
        }
        io += need.index;
        bo += need.bulk;
        if (packs) packs[i] = p;
        if (counts) {
            const int c = p.encode_type == 0 ? SL_RAW : (p.width == 32 ? SL_N32 : (p.width == 64 ? SL_N64 : SL_STR));
            if (lists) { lists[SL_ALL * kStaticMaxPacks + counts[SL_ALL]] = i; lists[c * kStaticMaxPacks + counts[c]] = i; }
            counts[SL_ALL]++; counts[c]++;
        }
    }
    // the same spare bytes behind the last array as a plan's scratch allocation has: kernels read up to 32 bytes past a
    // dictionary and 16 past a window
    return kStaticHeader + bo + 64 + 256;
}
void launch_static_setup(hipStream_t stream, const StaticBatch& batch, uint8_t* block);
void launch_export_status(hipStream_t stream, const PackMeta* metas, int count, int32_t* d_status);

// records the calling thread's last error (what ixr_last_error returns) and hands `code` back
int32_t set_error(int32_t code, const char* message);

// `list` is a device array of `count` pack indices.
void launch_parse(const LaunchCtx& c, const int* list, int count);
void launch_delta(const LaunchCtx& c, const int* list, int count, int width, int streams);
void launch_dict_num(const LaunchCtx& c, const int* list, int count, int width);
void launch_copy_raw(const LaunchCtx& c, const int* list, int count);
void launch_str_dict_scan(const LaunchCtx& c, const int* list, int count);
void launch_front_serial(const LaunchCtx& c, const int* list, int count);
void launch_front_summary(const LaunchCtx& c, const int* list, int count, int use_ds);
void launch_front_combine(const LaunchCtx& c, const int* list, int count, int use_ds);
void launch_front_decode(const LaunchCtx& c, const int* list, int count, int use_ds);
void launch_row_scan(const LaunchCtx& c, const int* list, int count);
void launch_str_index(const LaunchCtx& c, const int* list, int count);
void launch_str_copy(const LaunchCtx& c, const int* list, int count);
void launch_str_bases(const LaunchCtx& c, const int* list, int count);  // k_str_index that leaves one offset per 1024-row tile
void launch_str_fused(const LaunchCtx& c, const int* list, int count);  // row index + body copy, one wave per tile
void launch_ds_meta(const LaunchCtx& c, const int* list, int count);
void launch_ds_rows(const LaunchCtx& c, const int* list, int count, int mode);  // 0: 32 bit, 1: 64 bit, 2: strings
void launch_checksum(const LaunchCtx& c, int count, unsigned long long* sums);  // all packs of the plan
void launch_dict_struct(const LaunchCtx& c, const int* list, int count);

// predicate evaluation on dictionary ids: one record per pack of the plan, in device memory
struct DevFilter {
    int32_t op, n_values;
    uint64_t values_off;  // into the device copy of the key buffer
    uint32_t match_off;   // into the matched-id array (one slot per key; LIKE: one per row of the pack)
    int32_t result;       // out: 0 none, 1 all, 2 bitmap, < 0 error
    uint32_t from, to;    // out: id range [from, to) for the range predicates
    uint32_t n_match;     // out: number of matched ids (EQUAL, IN)
    uint32_t pad;
};
void launch_filter_search(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint8_t* values, uint32_t* matches);
void launch_filter_like(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint8_t* values, uint32_t* matches);
void launch_filter_rows(const LaunchCtx& c, const int* list, int count, DevFilter* filters, const uint32_t* matches,
                        unsigned long long* bitmaps);

// predicates on the rows of decoded packs (ExtIndex_SimpleBits); descs / results are device arrays, results zeroed by the caller
struct ScanFilterDesc {
    uint64_t src_off, values_off;
    int32_t data_type, obj_count, data_size, op, n_values, reserved;
};
void launch_scan_filter(hipStream_t stream, const ScanFilterDesc* descs, int count, const uint8_t* decoded, const uint8_t* values,
                        unsigned long long* bitmaps, int32_t* results);

// rows of decoded packs picked by a bitmap (DataPack.foreach(BitMap, ...)); descs / counts are device arrays
struct SelectDesc {
    uint64_t src_off, dst_off, bitmap_off;
    int32_t data_type, obj_count, data_size, all;
};
void launch_select(hipStream_t stream, const SelectDesc* descs, int count, const uint8_t* decoded, const uint8_t* bitmaps, uint8_t* out,
                   int32_t* counts);

// engine hand-back (decoded pack -> column vector contents); descs / status are device arrays
struct HandbackDesc {
    uint64_t src_off, dst_off;
    int32_t obj_count, data_size, sql_type, row_offset, row_count, reserved;
};
void launch_handback(hipStream_t stream, const HandbackDesc* descs, int count, const uint8_t* decoded, uint8_t* out, int32_t* status);

}  // namespace ixr
