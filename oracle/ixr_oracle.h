/*
 * ixr_oracle.h -- CPU oracle for IndexR VLT pack decode (and the matching encode).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links or calls this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  It is a plain-C restatement
 * of the reference's Java algorithm (file:line cited per function in ixr_oracle.c) and keeps the
 * reference's structure: byte-wise unpack8Values, two-pass delta (unpack, then running sum),
 * sequential RLE/varint walk, per-row dictionary gather, per-row string append.
 *
 * PARITY STATUS: the reference holds no golden encoded bytes and cannot be built here (no JDK), so
 * encoded-byte parity is "unpinned"; what pins this oracle is listed in oracle/README.md.
 */
#ifndef IXR_ORACLE_H
#define IXR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* io.indexr.data.DataType (indexr-common/.../data/DataType.java:4-8) */
enum { IXO_INT = 1, IXO_LONG = 2, IXO_FLOAT = 3, IXO_DOUBLE = 4, IXO_STRING = 5 };
/* io.indexr.vlt.codec.CodecType (indexr-vlt-segment/.../codec/CodecType.java:4-10) */
enum { IXO_PLAIN = 1, IXO_DELTA = 2, IXO_RLE = 3, IXO_SIMPLE_DICT = 4, IXO_DICT = 5, IXO_DICT_COMPRESS = 6 };
/* io.indexr.vlt.codec.ErrorCode (ErrorCode.java:4-7) */
enum { IXO_BUFFER_OVERFLOW = -1, IXO_DATA_CORRUPTED = -2, IXO_DATA_IMPROPER = -3, IXO_BUG = -100 };

/* ---- bit packing: IntPacker/LongPacker.packUnsafe / unpackUnsafe ---- */
int ixo_pack_ints(int bw, int n, const int32_t* in, uint8_t* out);
int ixo_unpack_ints(int bw, int n, const uint8_t* in, int32_t* out);
int ixo_pack_longs(int bw, int n, const int64_t* in, uint8_t* out);
int ixo_unpack_longs(int bw, int n, const uint8_t* in, int64_t* out);

/* ---- single codecs. Every decode takes the readable input size and returns bytes consumed or <0. ---- */
int ixo_delta_encode_ints(int n, const int32_t* in, uint8_t* out, int out_limit);
int ixo_delta_decode_ints(int n, const uint8_t* in, int in_limit, int32_t* out);
int ixo_delta_encode_longs(int n, const int64_t* in, uint8_t* out, int out_limit);
int ixo_delta_decode_longs(int n, const uint8_t* in, int in_limit, int64_t* out);
int ixo_delta_encode_strings(int n, const uint8_t* strings_struct, uint8_t* out, int out_limit);
int ixo_delta_decode_strings(int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit);
int ixo_rle_encode_ints(int bw, int n, const int32_t* in, uint8_t* out, int out_limit);
int ixo_rle_decode_ints(int bw, int n, const uint8_t* in, int in_limit, int32_t* out);

/* DictCodec.encode* -> uncompressed DictStruct; ratio is the dictEntryRatioThreshold (float). */
int ixo_dict_encode(int data_type, float ratio, int n, const uint8_t* data, uint8_t* out, int out_limit);
/* DictStruct.decode* : gather rows out of an uncompressed DictStruct. Returns bytes written or <0. */
int ixo_data_from_dict_struct(int data_type, int n, const uint8_t* dict_struct, int ds_size, uint8_t* out, int out_limit);
/* DictStructUtil.compress / decompress (the part after the 4-byte dictStructSize). */
int ixo_dict_struct_compress(int data_type, int n, const uint8_t* dict_struct, uint8_t* out, int out_limit);
int ixo_dict_struct_decompress(int data_type, int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit);
/* Size in bytes of an uncompressed DictStruct. */
int ixo_dict_struct_size(int data_type, const uint8_t* dict_struct);
/* SimpleDictCodec */
int ixo_simple_dict_encode(int data_type, int n, const uint8_t* data, uint8_t* out, int out_limit);

/* ---- pack level: PackCompressor_VLT ---- */
/* compress(): chooses the codec as the reference does; *encode_type receives the CodecType id. */
int ixo_compress_pack(int data_type, int is_indexed, int n, const uint8_t* data, int data_size,
                      uint8_t* out, int out_limit, int* encode_type);
/* Force a given codec (DELTA / SIMPLE_DICT / DICT_COMPRESS) -- for building test inputs. */
int ixo_compress_pack_as(int data_type, int encode_type, int n, const uint8_t* data, int data_size,
                         uint8_t* out, int out_limit);
/* decompress(): returns bytes consumed (the caller checks == packSize) or a negative ErrorCode. */
int ixo_decompress_pack(int data_type, int encode_type, int n, const uint8_t* in, int pack_size,
                        uint8_t* out, int data_size);
/* getDictStruct(): in = whole pack (starting at the 4-byte dictStructSize); returns consumed incl. those 4. */
int ixo_get_dict_struct(int data_type, int n, const uint8_t* in, int pack_size, uint8_t* out, int out_limit);

/* ---- batch driver for the CPU baseline: decode `count` packs with `threads` worker threads. ---- */
typedef struct {
    uint64_t in_off;      /* offset of the pack in the input arena */
    uint64_t out_off;     /* offset of the decoded pack in the output arena */
    int32_t  pack_size;
    int32_t  data_size;
    int32_t  obj_count;
    int32_t  encode_type; /* CodecType id; 0 = stored uncompressed (dpn.compress == false) */
    int32_t  data_type;
    int32_t  status;      /* out: consumed bytes or negative error */
} ixo_pack_desc;
int ixo_decode_batch(ixo_pack_desc* descs, int count, const uint8_t* in_arena, uint8_t* out_arena, int threads);

/* ---- engine hand-back: decoded pack -> Spark column vector contents ----
 * Restates DefaultPackReader.readByCopy / readBySet (indexr-spark/.../indexr/DefaultPackReader.java:164-241) over
 * DataPack.foreach (indexr-segment/.../pack/DataPack.java:257-458,627-706) for rows [row_offset, row_offset + row_count):
 *   INT/BIGINT/FLOAT/DOUBLE  the rows as they are;  DATE  int32 (int)(ms / 86400000);  DATETIME  int64 ms * 1000;
 *   VARCHAR  int32 arrayOffset[count] | int32 arrayLength[count] | bytes appended in row order (ColumnVector.putByteArray).
 * sql_type: 1 INT 2 BIGINT 3 FLOAT 4 DOUBLE 5 VARCHAR 6 DATE 7 TIME 8 DATETIME.  Returns bytes written, -1 when out_limit is
 * too small, -102 for a type the reader rejects (TIME: "Unsupported date type"), -1000 for bad arguments. */
int ixo_handback(int sql_type, int n, const uint8_t* decoded, int data_size, int row_offset, int row_count,
                 uint8_t* out, int out_limit);

/* ---- predicate evaluation on dictionary ids ----
 * Restates ExtIndex_DictBits (indexr-vlt-segment/.../vlt/segment/index/ExtIndex_DictBits.java:33-180) over the pack's
 * DictStruct: op 1 equal (:60-79), 2 in (:81-113), 3 greater (:115-145, acceptEqual = false), 4 greater with acceptEqual,
 * 5 between (:147-180), 6 like (:182-200; string packs only, else -102).  num_values: the reference's long numValue per key (INT/LONG: the value; FLOAT/DOUBLE: raw long bits
 * of the key as a double, :44,47); str_values: a StringsStruct of the keys.  bitmap: 1024 little-endian 64-bit words
 * (BitMap / DirectBitMap: row i is bit i & 63 of word i >> 6).  Returns 0 NONE, 1 ALL, 2 a bitmap; -100 where the reference's
 * Preconditions.checkState fails (:165-166); -3 when the pack is not DICT_COMPRESS; other negatives: decode errors. */
/* SQLLike.match(string, pattern) (indexr-common/.../util/SQLLike.java:12-94): 1 when the string matches. */
int ixo_like(const uint8_t* str, int str_len, const uint8_t* pattern, int pattern_len);
int ixo_filter(int data_type, int encode_type, int n, const uint8_t* pack, int pack_size, int op, int n_values,
               const int64_t* num_values, const uint8_t* str_values, uint64_t* bitmap);

/* ExtIndex_SimpleBits (indexr-segment/.../segment/index/ExtIndex_SimpleBits.java:25-331): the same predicates evaluated on
 * every row of a decoded pack (`data`, data_size bytes); keys as in ixo_filter.  Returns 0 (NONE) or 2 (bitmap), -102
 * for LIKE on a numeric pack, IXO_DATA_CORRUPTED for string offsets that leave the pack. */
int ixo_scan_filter(int data_type, int n, const uint8_t* data, int data_size, int op, int n_values,
                    const int64_t* num_values, const uint8_t* str_values, uint64_t* bitmap);

/* DataPack.foreach(BitMap position, XxxSetter) (indexr-segment/.../segment/pack/DataPack.java:273-306, 324-357, 375-408,
 * 426-459, 708-762): the setter calls for the set bits below n in rising order; bitmap == NULL is BitMap.ALL.  row_ids[k],
 * and per call either the value at values + k * width or str_off[k] / str_len[k] into the bytes appended to values.
 * Returns the number of calls, IXO_DATA_CORRUPTED for string offsets that leave the pack, -1000 for bad arguments. */
int ixo_select(int data_type, int n, const uint8_t* data, int data_size, const uint64_t* bitmap, int32_t* row_ids,
               uint8_t* values, int32_t* str_off, int32_t* str_len);

/* DataPackCreator_VLT.from (indexr-vlt-segment/.../segment/pack/DataPackCreator_VLT.java:21-163): dpn.minValue, maxValue
 * and sum of a numeric pack (floats: Math.min / Math.max fold, float / double running sum, raw bits of the widened value). */
int ixo_pack_stats(int data_type, int n, const uint8_t* data, int64_t* min_value, int64_t* max_value, int64_t* sum);

#ifdef __cplusplus
}
#endif
#endif
