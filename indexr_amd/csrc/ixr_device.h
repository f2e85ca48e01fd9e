// ixr_device.h -- structures shared by the host side of libixrgpu and its HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ixr {

// Geometry of the VLT DELTA codec (indexr-vlt-segment/.../codec/delta/DeltaCodec.java:31-32) and of the RLE tiling.
constexpr int kBlock = 128;        // values per delta block (one zig-zag varint minDelta each)
constexpr int kMini = 32;          // values per mini-block (one width byte each)
constexpr int kRleTile = 1024;     // rows per RLE segment record / per wave tile in the id-expanding kernels
#ifndef IXR_STR_TILE_ROWS
#define IXR_STR_TILE_ROWS 256
#endif
constexpr int kStrTileRows = IXR_STR_TILE_ROWS;  // rows per workgroup tile in the string body copy
constexpr int kFrontBlock = 64;    // strings per block of the parallel front-coding decode

// Element stride of the per-pack string length arrays (prefixLen / suffixLen / sufOff): obj_count rounded up to 4 so that
// each array starts 16-byte aligned.
// >= obj_count + 9: one terminating entry plus the 7 entries an 8-wide metadata read may touch past the last string
constexpr uint32_t kInSlack = 16;  // readable bytes after every pack in the input arena (IXR_IN_SLACK of the C ABI)
__host__ __device__ inline uint32_t lens_stride(int32_t obj_count) { return ((uint32_t)obj_count + 12u) & ~3u; }

// What the parse kernel found a pack to be.
enum PackKind : int32_t {
    K_NONE = 0,
    K_RAW = 1,        // stored uncompressed: copy
    K_DELTA_NUM = 2,  // DELTA over int32/int64 bit patterns, decoded straight into the output
    K_DICT_NUM = 3,   // DICT_COMPRESS or SIMPLE_DICT, numeric
    K_DICT_STR = 4,   // DICT_COMPRESS or SIMPLE_DICT, strings
    K_DELTA_STR = 5,  // whole pack front coded (DELTA on strings)
};

// One pack as the kernels see it: the caller's descriptor plus scratch offsets (relative to the group's scratch base)
// laid out by the host from obj_count / data_size alone.
struct DevPack {
    uint64_t in_off;
    uint64_t out_off;
    int32_t pack_size;
    int32_t data_size;
    int32_t obj_count;
    int32_t encode_type;
    int32_t data_type;
    int32_t width;       // 32 / 64 for numbers, 0 for strings
    uint64_t s_idx[2];   // DELTA index of stream 0 / 1:  int64 minDelta[ceil(cap/128)] | uint32 mb[ceil(cap/32)]
    uint64_t s_dict;     // decoded dictionary, aligned: numbers entryCount*w ; strings a StringsStruct
    uint64_t dict_cap;   // bytes available at s_dict
    uint64_t s_restart;  // RLE restart records: uint4[ceil(cap/kRleTile)] = (payload offset, first row, row count, packed | run value << 1)
    uint64_t s_lens;     // strings: int32 prefixLen[S] | int32 suffixLen[S] | uint32 sufOff[S], S = lens_stride(obj_count) >= cap+9
    uint64_t s_rows;     // strings: uint32 srcOff[cap] (offset of each row's bytes inside the dictionary body)
    uint64_t ds_off;     // only for the getDictStruct run: where to write the DictStruct in the output arena
};

// Lays the scratch regions of one pack out from obj_count / data_size alone.  Two areas:
//   index area (kept for a whole plan, filled by one parse launch over all packs): DELTA indexes, per-tile RLE segment records
//   bulk area  (reused group after group): decoded dictionaries, string length arrays, per-row source offsets
// Offsets are written relative to each area's start; the caller rebases them.  Returns the bytes used in each.
struct ScratchNeed { uint64_t index, bulk; };
__host__ __device__ inline uint64_t scratch_align16(uint64_t v) { return (v + 15) & ~15ull; }
__host__ __device__ inline ScratchNeed pack_scratch_layout(DevPack& p, uint64_t dict_cap_override) {
    uint64_t oi = 0, ob = 0;
    const uint64_t cap = (uint64_t)p.obj_count;
    const uint64_t idx = scratch_align16(8 * ((cap + kBlock - 1) / kBlock) + 4 * ((cap + kMini - 1) / kMini));
    const uint64_t restart = 16 * ((cap + kRleTile - 1) / kRleTile);
    p.s_idx[0] = p.s_idx[1] = p.s_dict = p.s_restart = p.s_lens = p.s_rows = 0;
    p.dict_cap = 0;
    if (p.encode_type == 0 /* stored uncompressed */) return {0, 0};
    p.s_idx[0] = oi; oi += idx;
    const bool dict = p.encode_type == 6 /* DICT_COMPRESS */ || p.encode_type == 4 /* SIMPLE_DICT */;
    if (p.width) {
        if (dict) {
            p.dict_cap = dict_cap_override ? dict_cap_override : cap * (uint64_t)(p.width / 8);
            p.s_dict = ob; ob += scratch_align16(p.dict_cap + 16);
            p.s_restart = oi; oi += restart;
        }
    } else {
        p.s_idx[1] = oi; oi += idx;
        p.s_lens = ob; ob += scratch_align16(4ull * 3 * lens_stride(p.obj_count));
        if (dict) {
            p.dict_cap = dict_cap_override ? dict_cap_override : (uint64_t)p.data_size;
            p.s_dict = ob; ob += scratch_align16(p.dict_cap + 32);
            p.s_rows = ob; ob += scratch_align16(4 * cap + 16);
            p.s_restart = oi; oi += restart;
        }
    }
    return {oi, ob};
}

struct StreamDesc {
    uint32_t n;       // values in the stream (0 = absent)
    uint32_t in_off;  // offset of the stream's 4-byte length word, relative to the pack
    int64_t first;    // first value (zig-zag decoded)
};

// Written by the parse kernel, read by every later kernel.
struct PackMeta {
    int32_t status;         // bytes consumed, or a negative error code
    int32_t kind;           // PackKind
    uint32_t entry_count;   // dictionary entries
    uint32_t id_bw;         // bit width of the ids = bitWidth(entryCount - 1)
    uint32_t ids_off;       // offset of the RLE stream's length word in the pack
    uint32_t ids_len;       // its encodedLen
    uint32_t entries_off;   // offset of plain (uncompressed) entries in the pack
    uint32_t entries_plain; // 1: entries are stored plain at entries_off
    uint32_t sfx_off;       // front coding: offset of the suffix bytes in the pack
    uint32_t sfx_len;       // front coding: number of suffix bytes the headers promise
    uint32_t dict_struct_size; // DICT_COMPRESS: the 4-byte uncompressed DictStruct size at the head of the pack
    uint32_t max_len;       // strings: longest dictionary entry (filled by the dictionary scan)
    uint32_t front_blocks;  // front-coded dictionary: number of 64-string blocks decoded in parallel (0 = serial path)
    uint32_t front_stride;  // columns per block row of the block-summary array (max_len rounded up to 64)
    uint32_t sliced;        // numeric dictionary pack whose rows ask for the entries in order: no dictionary in scratch, only one
                            // record per 128-entry block (k_delta), the rows come from the slice path of k_dict_num
    StreamDesc s[2];
};

}  // namespace ixr
