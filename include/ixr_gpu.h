/*
 * ixr_gpu.h -- C ABI of libixrgpu: IndexR VLT pack decode on AMD Instinct MI355X (gfx950).
 *
 * This is the boundary a JVM binding (JNI / Panama), a C++ host or a ctypes caller links against.  Signatures use
 * plain pointers and sizes only.  Citations name the reference interface each entry point replaces; paths are
 * relative to the reference tree (shunfei/indexr v0.6.1).
 *
 * Conventions shared with the reference:
 *   - everything is little-endian, native order (indexr-common/.../util/ByteBufferUtil.java:31-36);
 *   - decode functions return the number of compressed bytes consumed, or a negative ErrorCode
 *     (indexr-vlt-segment/.../codec/Decoder.java:3-25, .../codec/ErrorCode.java:4-7); the caller checks
 *     consumed == dpn.packSize exactly as PackCompressor_VLT.decompress does (:168);
 *   - the caller allocates the output of exactly dpn.dataSize bytes (PackCompressor_VLT.java:133).
 * Threading: entry points may be called from any number of threads, and different plans are independent of each other.
 * ONE plan supports one run at a time: ixr_plan_run, ixr_plan_run_dict_structs, ixr_plan_run_filter and ixr_plan_status
 * of the same plan share its scratch memory, its device copy of the descriptors, its auxiliary stream and its result
 * records, so the caller orders them (same stream, or a synchronisation between runs on different streams) and never
 * calls them on one plan from two threads at once.  The reference has the same rule per DataPack (DPSegment.java:52).
 */
#ifndef IXR_GPU_H
#define IXR_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* io.indexr.segment.ColumnType / io.indexr.data.DataType ids (indexr-common/.../data/DataType.java:4-8) */
enum { IXR_INT = 1, IXR_LONG = 2, IXR_FLOAT = 3, IXR_DOUBLE = 4, IXR_STRING = 5 };
/* dpn.encodeType values that appear on disk (indexr-vlt-segment/.../codec/CodecType.java:4-10); 0 = the pack is stored
 * uncompressed (dpn.compress == false: DataPack.java:180-182, FAST mode SegmentMode.java:17) */
enum { IXR_ENC_RAW = 0, IXR_ENC_DELTA = 2, IXR_ENC_SIMPLE_DICT = 4, IXR_ENC_DICT_COMPRESS = 6 };
/* ErrorCode.java:4-7 plus library-level failures (<= -1000) */
enum {
    IXR_ERR_BUFFER_OVERFLOW = -1,
    IXR_ERR_DATA_CORRUPTED = -2,
    IXR_ERR_DATA_IMPROPER = -3,
    IXR_ERR_BUG = -100,
    IXR_ERR_ILLEGAL_ENCODE_TYPE = -101, /* RuntimeException("Illegal encode type") PackCompressor_VLT.java:146 */
    IXR_ERR_UNSUPPORTED_TYPE = -102,    /* RuntimeException("Unsupported data type") :166 */
    IXR_ERR_INVALID_ARGUMENT = -1000,
    IXR_ERR_NO_DEVICE = -1001, /* no usable gfx950 device / HIP runtime failure; never a silent CPU fallback */
    IXR_ERR_HIP = -1002,
    IXR_ERR_OUT_OF_MEMORY = -1003
};

/* Fixed geometry of the format (indexr-segment/.../pack/DataPack.java:36-38) */
#define IXR_PACK_MAX_ROWS 65536
/* Readable slack the library needs after the last byte of every compressed pack inside a device input arena. */
#define IXR_IN_SLACK 16
/* Alignment required of in_off / out_off inside the arenas. */
#define IXR_ARENA_ALIGN 16

/*
 * One pack of work: the subset of DataPackNode_VLT (indexr-vlt-segment/.../segment/pack/DataPackNode_VLT.java:12-102;
 * objCount@0, compress@4, packSize@13, encodeType@57, dataSize@61) the decoder needs, the column's data type (from the
 * segment schema) and where the pack's bytes live inside the caller's input / output arenas.
 */
typedef struct ixr_pack_desc {
    uint64_t in_off;      /* offset of the compressed pack in the input arena, multiple of IXR_ARENA_ALIGN */
    uint64_t out_off;     /* offset of the decoded pack in the output arena, multiple of IXR_ARENA_ALIGN */
    int32_t  pack_size;   /* dpn.packSize  */
    int32_t  data_size;   /* dpn.dataSize  */
    int32_t  obj_count;   /* dpn.objCount, 1..65536 */
    int32_t  encode_type; /* dpn.encodeType id, or IXR_ENC_RAW when dpn.compress is false */
    int32_t  data_type;   /* IXR_INT .. IXR_STRING */
    int32_t  reserved;
} ixr_pack_desc;

/* ---- library / device --------------------------------------------------------------------------------------- */
/* Bind the calling thread's work to `device`, check it is gfx950. Returns 0 or a negative error. Idempotent. */
int32_t ixr_init(int32_t device);
/* Human-readable text for the last error on this thread. */
const char* ixr_last_error(void);
/* Number of visible devices (0 when there is no GPU / HIP runtime). */
int32_t ixr_device_count(void);

/* Switches for A/B timing runs and for tests that must reach both variants of a kernel choice; no counterpart in the
 * reference, not for production use.  They start from the environment (IXR_SPLIT_STR, IXR_WAVE_WALK_FROM, IXR_SLICE, read
 * once per process) and apply to plans created afterwards:
 *   "split_str"       1: string rows by the two-kernel form (per-row offsets through scratch), 0: the fused kernel
 *   "slice"           1 (default): sorted numeric dictionaries larger than 64 KB whose rows take the entries in order are decoded
 *                     per 1024-row tile and never written to scratch, 0: they go through scratch like every other dictionary
 *   "wave_walk_from"  packs per launch from which every framing stream is followed by a workgroup of one wave; 0 = built-in choice
 * Returns 0, or IXR_ERR_INVALID_ARGUMENT for an unknown name. */
int32_t ixr_debug_switch(const char* name, int32_t value);

/* ---- device memory helpers (so a JVM caller needs no HIP binding of its own) ---------------------------------- */
void*   ixr_dev_alloc(int32_t device, uint64_t bytes);
void    ixr_dev_free(int32_t device, void* dptr);
void*   ixr_host_alloc_pinned(uint64_t bytes);
void    ixr_host_free_pinned(void* hptr);
int32_t ixr_memcpy_h2d(int32_t device, void* dst_dev, const void* src_host, uint64_t bytes);
int32_t ixr_memcpy_d2h(int32_t device, void* dst_host, const void* src_dev, uint64_t bytes);
int32_t ixr_dev_synchronize(int32_t device);
/* Streams for callers without a HIP binding of their own: a non-blocking hipStream_t as an opaque pointer, accepted by
 * every `stream` parameter below.  No counterpart in the reference (its decode is synchronous on the calling thread). */
void*   ixr_stream_create(int32_t device);
void    ixr_stream_destroy(int32_t device, void* stream);
int32_t ixr_stream_synchronize(int32_t device, void* stream);

/* ---- batch decode, device resident (the hot path) -------------------------------------------------------------
 * A plan is built once per set of pack descriptors (DPN metadata is loaded when a segment is opened:
 * indexr-segment/.../storage/StorageColumn.java:47-67); it owns the device copy of the descriptors, the work lists and
 * the scratch memory.  ixr_plan_run decodes every pack of the plan: compressed bytes are read from `d_in`
 * (in_off/pack_size), decoded packs are written to `d_out` (out_off/data_size).  It replaces, for many packs at once,
 *   DataPack.decompress              indexr-segment/.../pack/DataPack.java:166-186
 *   PackCompressor_VLT.decompress    indexr-vlt-segment/.../segment/pack/PackCompressor_VLT.java:130-170
 *   getDictStruct + getDataFromDictStruct (the dictionary path)   :172-213
 * `stream` is a hipStream_t (NULL = the default stream); the call only enqueues work.  The input arena must be
 * readable for IXR_IN_SLACK bytes past every pack (ixr_plan_in_bytes accounts for it).
 */
typedef struct ixr_plan ixr_plan;
/* scratch_budget_bytes: upper bound for the plan's scratch memory (0 = default 16 GiB); packs are processed in
 * groups that fit.  Returns NULL on error (see ixr_last_error). */
ixr_plan* ixr_plan_create(int32_t device, const ixr_pack_desc* descs, int32_t n_packs, uint64_t scratch_budget_bytes);
void      ixr_plan_destroy(ixr_plan* plan);
/* Minimum sizes of the arenas implied by the descriptors (max(in_off+pack_size)+slack, max(out_off+data_size)). */
uint64_t  ixr_plan_in_bytes(const ixr_plan* plan);
uint64_t  ixr_plan_out_bytes(const ixr_plan* plan);
uint64_t  ixr_plan_scratch_bytes(const ixr_plan* plan);
int32_t   ixr_plan_run(ixr_plan* plan, const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes, void* stream);
/* Copies the per-pack results of the last run to `status` (n_packs entries: bytes consumed, or a negative error);
 * synchronises with `stream`. Returns the number of packs whose status != pack_size. */
int32_t   ixr_plan_status(ixr_plan* plan, int32_t* status, void* stream);
/* Algorithmic bytes of the plan: sum(pack_size) and sum(data_size); rows = sum(obj_count). */
void      ixr_plan_totals(const ixr_plan* plan, uint64_t* in_bytes, uint64_t* out_bytes, uint64_t* rows);
/* Names and accumulated device time (ms, measured with HIP events on `stream` when profiling was enabled with
 * ixr_plan_set_profiling) of the kernels of the last run; returns the number of entries written (<= cap). */
void      ixr_plan_set_profiling(ixr_plan* plan, int32_t enabled);
/* A run keeps two streams busy: the numeric kernels of a group go to a stream of the plan's own beside the string kernels
 * on the caller's stream.  enabled = 0 keeps everything on the caller's stream, one kernel after the other (the default is
 * 1); a kernel's event time then is its time alone on the device, which is what kernel tuning wants to look at. */
void      ixr_plan_set_overlap(ixr_plan* plan, int32_t enabled);
int32_t   ixr_plan_kernel_times(ixr_plan* plan, const char** names, float* ms, int32_t* launches, int32_t cap);
/* Algorithmic bytes attributed to each kernel of the last run (same order and names as ixr_plan_kernel_times, valid after
 * ixr_plan_status): every compressed byte of a pack is attributed to the kernel that consumes it and every decoded byte
 * to the kernel that produces it, so the entries add up to sum(pack_size) + sum(data_size); traffic to and from scratch
 * memory is not counted. */
int32_t   ixr_plan_kernel_bytes(ixr_plan* plan, uint64_t* alg_bytes, int32_t cap);
/* The same decode with both arenas in HOST memory (the descriptors' offsets then are offsets into h_in / h_out): what a
 * reader pays whose packs are not resident on the device, i.e. the path of StorageColumn.pack -> DataPack.decompress
 * (indexr-segment/.../storage/StorageColumn.java:78-116) including the trip over PCIe.  The plan's groups (see
 * scratch_budget_bytes: a smaller budget gives more, smaller groups) go through two device staging buffers per direction:
 * the copy in of group g+1, the decode of group g and the copy out of group g-1 run at the same time on three streams of
 * the plan's own.  The call returns when all rows are in h_out; ixr_plan_status(plan, status, NULL) then reports per pack.
 * Both arenas should be pinned (ixr_host_alloc_pinned): pageable memory works, without the overlap.  A group's rows travel
 * back as ONE copy of the stretch of the output arena its packs span, so the packs must lie in plan order and, within a
 * group, back to back in the output arena: at most the 15 alignment bytes between two packs' rows (those bytes are
 * overwritten with unspecified values); in the input arena gaps of up to 4096 bytes are accepted.  Any other layout is
 * refused with IXR_ERR_INVALID_ARGUMENT before anything is copied (ixr_plan_run takes it).  On any error the call returns
 * only after the copies it has started have ended.  The staging buffers, sized for the largest group, stay with the plan
 * until it is destroyed. */
int32_t   ixr_plan_run_host(ixr_plan* plan, const void* h_in, uint64_t in_bytes, void* h_out, uint64_t out_bytes);

/* ---- batch decode that allocates nothing ----------------------------------------------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device memory at the point of the call: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Decodes 1..IXR_STATIC_MAX_PACKS packs exactly as
 * ixr_plan_create + ixr_plan_run would (same kernels, same arenas, same results), but without a plan: the descriptors
 * travel as a kernel argument and every working array (device records, framing indexes, decoded dictionaries, string
 * offsets) lives in an array of IXR_STATIC_SCRATCH_BYTES that is part of the library's code object; each GPU the library
 * runs on holds its copy until the process ends.  The call makes no hipMalloc / hipFree, creates no stream or event, and
 * only enqueues kernels on `stream`.
 *   d_status   device memory, n_packs x int32: receives what ixr_plan_status reports (bytes consumed or a negative error);
 *              read it after synchronising with `stream`
 * A batch whose working arrays exceed the array (about 5 MB per full VARCHAR pack of 3.6 MB, 0.6 MB per full numeric pack), or
 * of more than IXR_STATIC_MAX_PACKS packs, is rejected with IXR_ERR_INVALID_ARGUMENT (ixr_last_error says how much it
 * needs).  All calls on one GPU share that GPU's array, which keeps the last call's contents (the call clears what must
 * start at zero): calls on the same GPU must not overlap -- order them on one stream or synchronise between them, two
 * threads or two streams running it at once on one GPU race.  The first call on a device resolves the array's address
 * (ixr_init does the same); call ixr_init beforehand where even that must not happen inside the critical section.
 */
#define IXR_STATIC_MAX_PACKS 64
#define IXR_STATIC_SCRATCH_BYTES (32ull << 20)
int32_t   ixr_decode_batch_static(int32_t device, const ixr_pack_desc* descs, int32_t n_packs, const void* d_in,
                                  uint64_t in_bytes, void* d_out, uint64_t out_bytes, int32_t* d_status, void* stream);

/* ---- dictionary form on device ------------------------------------------------------------------------------------
 * Replaces PackCompressor_VLT.getDictStruct (:172-186) for every DICT_COMPRESS pack of the plan: writes the
 * uncompressed DictStruct (indexr-common/.../data/DictStruct.java:14-55:  entryCount | entries | 4n | ids) of pack i at
 * d_out + ds_off[i]; ds_off are byte offsets supplied by the caller (multiples of IXR_ARENA_ALIGN).  The DictStruct's own
 * size (dictStructSize, the pack's first 4 bytes) is only known on the device, so the region of a DICT_COMPRESS pack must
 * hold the largest DictStruct its descriptor allows, 8 + data_size + 4 * obj_count bytes, inside the arena; the call
 * rejects anything smaller.  Packs of other encodings are skipped: nothing of them is decoded or written, their ds_off
 * entries are not looked at beyond the alignment check (0 is fine), and the arena needs no room for them.
 */
int32_t   ixr_plan_run_dict_structs(ixr_plan* plan, const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                    const uint64_t* ds_off, void* stream);

/* ---- single pack, host memory in / host memory out ---------------------------------------------------------------
 * Same contract as the reference's Decoder.decodeInts|Longs|Floats|Doubles|Strings(valCount, inputAddr, outputAddr,
 * outputLimit) (Decoder.java:5-25) dispatched like PackCompressor_VLT.decompress (:130-170).  Copies the pack to the
 * device, decodes it there, copies the result back.  Meant for parity tests and small callers; engines should batch.
 * All three calls work out of a per-device pool of contexts (a stream, device and pinned staging buffers, a working block of
 * IXR_STATIC_SCRATCH_BYTES): a call allocates nothing once a context exists, any number of threads may call at once (each takes
 * a context of its own), and the calling thread POLLS its stream for up to one millisecond before it falls back to a blocking
 * wait (a decode takes tens to hundreds of microseconds; a blocking wait's wake-up alone can take longer), so a call occupies
 * its host core for its whole duration.  At most 8 idle contexts per device are kept until the process ends (about 34 MiB of
 * device memory and 2 MiB of pinned memory each; buffers that a large pack grew beyond 16 MiB are released after the call).
 * A pack whose working arrays exceed the block (very long strings) takes a one-pack plan instead, with its allocations.
 */
int32_t ixr_decode_pack(int32_t device, int32_t data_type, int32_t encode_type, int32_t obj_count,
                        const void* in, int32_t in_size, void* out, int32_t out_limit);
/* PackCompressor_VLT.getDictStruct (:172-186): `in` is the whole pack; out receives dictStructSize bytes. */
int32_t ixr_get_dict_struct(int32_t device, int32_t data_type, int32_t obj_count,
                            const void* in, int32_t in_size, void* out, int32_t out_limit);
/* PackCompressor_VLT.getDataFromDictStruct (:188-213) / DictStruct.decodeXxx (DictStruct.java:57-110): gather the rows
 * out of an uncompressed DictStruct. Returns bytes written or a negative error. */
int32_t ixr_get_data_from_dict_struct(int32_t device, int32_t data_type, int32_t obj_count,
                                      const void* dict_struct, int32_t ds_size, void* out, int32_t out_limit);

/* ---- predicate evaluation on dictionary ids ---------------------------------------------------------------------
 * Replaces, for every DICT_COMPRESS pack of the plan at once, the per-pack filters of ExtIndex_DictBits
 * (indexr-vlt-segment/src/main/java/io/indexr/vlt/segment/index/ExtIndex_DictBits.java): equal :60-79, in :81-113,
 * greater :115-145, between :147-180, like :182-200.  A key is looked up in the pack's sorted dictionary (searchEntry
 * :33-58, BinarySearch.java) and the rows' ids are compared; rows are never materialised.  LIKE matches its pattern
 * (SQLLike.match, indexr-common/src/main/java/io/indexr/util/SQLLike.java:12-94: % any run of bytes, _ one byte,
 * backslash escapes) against every dictionary entry once and marks the rows whose id matched, which selects the same
 * rows as the reference's match on every row's string; string packs only, a numeric pack answers
 * IXR_ERR_UNSUPPORTED_TYPE where the reference throws IllegalStateException (:198); the result is NONE or SOME, never
 * ALL (fixBitmapInPack, indexr-segment/src/main/java/io/indexr/segment/PackExtIndex.java:32-39).
 *   filters[i]   the predicate for pack i of the plan
 *   values       host buffer holding the keys, values_off bytes in (multiple of 8): numeric packs: n_values x int64, the
 *                reference's `long numValue` (INT/LONG: the value; FLOAT/DOUBLE: the raw long bits of the key as a
 *                double, :44,47); string packs, and the pattern of a LIKE: a StringsStruct of the n_values keys (int32
 *                off[n_values+1] | bytes)
 *   d_bitmaps    device memory, IXR_BITMAP_BYTES per pack: 1024 little-endian 64-bit words, row r = bit r & 63 of word
 *                r >> 6 (BitMap / DirectBitMap, indexr-common/.../util/DirectBitMap.java); all 1024 words are written
 *   results[i]   IXR_BITMAP_NONE / IXR_BITMAP_ALL (the reference returns its constants; the bitmap is all zeros / ones up to
 *                objCount) / IXR_BITMAP_SOME, or a negative error: IXR_ERR_DATA_IMPROPER for a pack that is not
 *                DICT_COMPRESS (the reference gives those ExtIndex_SimpleBits: decode them with ixr_plan_run and use
 *                ixr_scan_filter_run), IXR_ERR_BUG where the reference's
 *                Preconditions.checkState fails (between :165-166), decode errors as in ixr_plan_status
 * The call synchronises with `stream`.  Returns the number of packs with a negative result, or a negative library error.
 */
enum { IXR_FILTER_EQUAL = 1, IXR_FILTER_IN = 2, IXR_FILTER_GREATER = 3, IXR_FILTER_GREATER_EQUAL = 4, IXR_FILTER_BETWEEN = 5,
       IXR_FILTER_LIKE = 6 };
enum { IXR_BITMAP_NONE = 0, IXR_BITMAP_ALL = 1, IXR_BITMAP_SOME = 2 };
#define IXR_BITMAP_BYTES 8192
typedef struct ixr_filter_desc {
    int32_t  op;         /* IXR_FILTER_* */
    int32_t  n_values;   /* EQUAL, GREATER, GREATER_EQUAL, LIKE: 1; BETWEEN: 2; IN: >= 1 */
    uint64_t values_off; /* where this pack's keys start in `values`, multiple of 8 */
} ixr_filter_desc;
int32_t ixr_plan_run_filter(ixr_plan* plan, const void* d_in, uint64_t in_bytes, const ixr_filter_desc* filters,
                            const void* values, uint64_t values_bytes, void* d_bitmaps, uint64_t bitmap_bytes,
                            int32_t* results, void* stream);

/* ---- predicates on the rows of decoded packs ----------------------------------------------------------------------
 * What packs without a dictionary get in the reference: ExtIndex_SimpleBits
 * (indexr-segment/src/main/java/io/indexr/segment/index/ExtIndex_SimpleBits.java): equal :25-89, in :91-166, greater
 * :168-234, between :236-311, like :313-331 (VersionAdapter_VLT.createExtIndex picks it when !dpn.isDictEncoded(),
 * indexr-vlt-segment/.../VersionAdapter_VLT.java:167-181).  It reads every row of the decoded pack: INT compares
 * (int) numValue, LONG numValue, FLOAT (float) Double.longBitsToDouble(numValue), DOUBLE Double.longBitsToDouble(numValue)
 * with the language's ==, >, >=, <= (a NaN never matches, -0.0 equals 0.0); strings compare as unsigned bytes, then by
 * length (BytesUtil.compareBytes); LIKE is SQLLike.match on each row and rejects numeric packs
 * (IXR_ERR_UNSUPPORTED_TYPE).  Works on any decoded pack, dictionary encoded or not: the input is a decoded-pack arena
 * as ixr_plan_run leaves it (src_off = the pack's out_off).  Keys, bitmaps and the return value are as in
 * ixr_plan_run_filter; results[i] is IXR_BITMAP_NONE or IXR_BITMAP_SOME (never ALL: fixBitmapInPack,
 * PackExtIndex.java:32-39), IXR_ERR_DATA_CORRUPTED for string offsets that leave the pack, IXR_ERR_INVALID_ARGUMENT for a
 * data_size too small for obj_count rows.  The call synchronises with `stream`.
 */
typedef struct ixr_scan_filter_desc {
    uint64_t src_off;    /* decoded pack inside the decoded arena, multiple of 8 */
    uint64_t values_off; /* where this pack's keys start in `values`, multiple of 8 */
    int32_t  data_type;  /* IXR_INT ... IXR_STRING */
    int32_t  obj_count;  /* rows in the pack */
    int32_t  data_size;  /* dpn.dataSize of the pack */
    int32_t  op;         /* IXR_FILTER_* */
    int32_t  n_values;   /* as in ixr_filter_desc */
    int32_t  reserved;
} ixr_scan_filter_desc;
int32_t ixr_scan_filter_run(int32_t device, const ixr_scan_filter_desc* descs, int32_t n, const void* d_decoded,
                            uint64_t decoded_bytes, const void* values, uint64_t values_bytes, void* d_bitmaps,
                            uint64_t bitmap_bytes, int32_t* results, void* stream);

/* ---- the rows a bitmap picks ----------------------------------------------------------------------------------------
 * The batch form of DataPack.foreach(BitMap position, XxxSetter)
 * (indexr-segment/src/main/java/io/indexr/segment/pack/DataPack.java: int :273-306, long :324-357, float :375-408,
 * double :426-459, BytePiece :708-762): what a reader does with the bitmap a predicate produced.  The setter is called
 * for every set bit below objCount in rising row order (BitMap.ALL: every row), and the return value is the number of
 * calls; here the calls' arguments are laid out at dst_off, n = obj_count:
 *   numbers   int32 rowId[n] | value[n], the values starting at the next multiple of 8 after 4 * n
 *   strings   int32 rowId[n] | int32 arrayOffset[n] | int32 arrayLength[n] | the picked rows' bytes one after the other
 *             (arrayOffset counts from the start of those bytes)
 * of which the first counts[i] entries of each array are written.  Input: a decoded-pack arena as ixr_plan_run leaves it
 * and a bitmap arena as the two filter calls leave it (bits at or above obj_count are ignored, as the reference's loop
 * stops there).  dst_off must be a multiple of IXR_ARENA_ALIGN, with ixr_select_bound(desc) bytes behind it.  counts[i]
 * is the number of picked rows, IXR_ERR_DATA_CORRUPTED for string offsets that leave the pack, IXR_ERR_INVALID_ARGUMENT
 * for a data_size too small for obj_count rows.  The call synchronises with `stream`.  Returns the number of packs with
 * a negative count, or a negative library error.
 */
typedef struct ixr_select_desc {
    uint64_t src_off;    /* decoded pack inside the decoded arena, multiple of 8 */
    uint64_t dst_off;    /* output inside the select arena, multiple of IXR_ARENA_ALIGN */
    uint64_t bitmap_off; /* the pack's IXR_BITMAP_BYTES inside the bitmap arena, multiple of 8; unused when all = 1 */
    int32_t  data_type;  /* IXR_INT ... IXR_STRING */
    int32_t  obj_count;  /* rows in the pack */
    int32_t  data_size;  /* dpn.dataSize of the pack */
    int32_t  all;        /* 1: position == BitMap.ALL; 0: read the bitmap */
} ixr_select_desc;
uint64_t ixr_select_bound(const ixr_select_desc* desc);
int32_t  ixr_select_run(int32_t device, const ixr_select_desc* descs, int32_t n, const void* d_decoded,
                        uint64_t decoded_bytes, const void* d_bitmaps, uint64_t bitmap_bytes, void* d_out,
                        uint64_t out_bytes, int32_t* counts, void* stream);

/* ---- engine hand-back on device -----------------------------------------------------------------------------------
 * The step right after the decode in the Spark reader: decoded pack -> column vector contents for rows
 * [row_offset, row_offset + row_count), DefaultPackReader.readByCopy / readBySet
 * (indexr-spark/src/main/java/org/apache/spark/sql/execution/datasources/indexr/DefaultPackReader.java:164-241):
 *   INT / BIGINT / FLOAT / DOUBLE   the rows as they are (HackColumnVector.copyInts ..., HackColumnVector.java:22-104)
 *   DATE      int64 milliseconds -> int32 days, (int) (value / 86400000)          (:214-219)
 *   DATETIME  int64 milliseconds -> int64 microseconds, value * 1000 (wrapping)   (:220-225)
 *   VARCHAR   int32 arrayOffset[row_count] | int32 arrayLength[row_count] | the rows' bytes in row order, which is what
 *             ColumnVector.putByteArray(rowId, bytes, off, len) leaves in its offset / length / child byte vectors (:226-232)
 *   TIME      rejected like the reference does ("Unsupported date type", :233-234) -> IXR_ERR_UNSUPPORTED_TYPE
 * Input: a decoded-pack arena as ixr_plan_run leaves it (src_off = the pack's out_off).  Output offsets must be multiples
 * of IXR_ARENA_ALIGN; ixr_handback_bound gives the bytes a descriptor can produce.  status[i] receives the bytes written
 * for descriptor i or a negative error.  The call synchronises with `stream`.  Returns the number of failed descriptors
 * or a negative library error.
 */
enum { IXR_SQL_INT = 1, IXR_SQL_BIGINT = 2, IXR_SQL_FLOAT = 3, IXR_SQL_DOUBLE = 4, IXR_SQL_VARCHAR = 5, IXR_SQL_DATE = 6,
       IXR_SQL_TIME = 7, IXR_SQL_DATETIME = 8 };
typedef struct ixr_handback_desc {
    uint64_t src_off;    /* decoded pack inside the decoded arena */
    uint64_t dst_off;    /* output inside the hand-back arena, multiple of IXR_ARENA_ALIGN */
    int32_t  obj_count;  /* rows in the pack */
    int32_t  data_size;  /* dpn.dataSize of the pack */
    int32_t  sql_type;   /* IXR_SQL_* */
    int32_t  row_offset; /* first row handed back */
    int32_t  row_count;  /* number of rows handed back */
    int32_t  reserved;
} ixr_handback_desc;
uint64_t ixr_handback_bound(const ixr_handback_desc* desc);
int32_t  ixr_handback_run(int32_t device, const ixr_handback_desc* descs, int32_t n, const void* d_decoded,
                          uint64_t decoded_bytes, void* d_out, uint64_t out_bytes, int32_t* status, void* stream);

/* ---- host-side write path (needed to build inputs; CPU only) --------------------------------------------------------
 * PackCompressor_VLT.compress (:23-128): codec choice and encoding of one pack. `data` is the decoded pack
 * (numbers: raw LE array; strings: StringsStruct, indexr-common/.../data/StringsStruct.java:12-17).
 * Returns the compressed size or a negative error; *encode_type receives the CodecType id.
 */
int32_t ixr_compress_pack(int32_t data_type, int32_t is_indexed, int32_t obj_count, const void* data, int32_t data_size,
                          void* out, int32_t out_limit, int32_t* encode_type);
/* Upper bound for the compressed size (Codec.outputBufferSize, indexr-vlt-segment/.../codec/Codec.java:7-27). */
int32_t ixr_compress_bound(int32_t obj_count, int32_t data_size);

/* ---- write path on the device ---------------------------------------------------------------------------------------
 * The batch form of PackCompressor_VLT.compress (:23-128): the same codec choice (INT not indexed: dictionary when it has
 * at most ceil(n * 0.2f) entries, else DELTA; everything else: dictionary) and the same bytes as ixr_compress_pack, which
 * is the reference's output.  A string pack whose dictionary form would not fit its slot reports
 * IXR_ERR_BUFFER_OVERFLOW (the reference falls back to DELTA on strings there, :106-110; use ixr_compress_pack).
 *   d_data      device arena holding the decoded packs (numbers: raw little-endian arrays; strings: StringsStruct, whose
 *               offsets are checked on the device: IXR_ERR_INVALID_ARGUMENT in pack_size), src_off multiple of 16
 *   d_out       device arena; every pack gets a slot of ixr_compress_bound(obj_count, data_size) bytes at dst_off
 *               (multiple of 16); the compressed pack starts at dst_off
 *   descs[i]    pack_size receives the compressed size or a negative error, encode_type the CodecType id; min_value,
 *               max_value and sum receive what DataPackCreator_VLT.from (DataPackCreator_VLT.java:21-163) puts into the
 *               DataPackNode: INT / LONG: min, max and the wrapping 64-bit sum; FLOAT / DOUBLE: the Math.min / Math.max
 *               fold (-0.0 < 0.0; with NaNs in the pack both are the last NaN) as Double.doubleToRawLongBits of the
 *               widened value, and the raw bits of the row-after-row sum in the column's precision, widened to double
 *               (a NaN sum is some NaN: which payload survives an addition is the hardware's choice); string packs: 0
 * scratch_budget: device bytes the call may allocate for its working arrays (0 = 4 GiB); packs are processed in groups
 * that fit.  The call synchronises with `stream`.  Returns the number of packs that failed or a negative library error.
 */
typedef struct ixr_encode_desc {
    uint64_t src_off;
    uint64_t dst_off;
    int32_t  data_type;   /* IXR_INT ... IXR_STRING */
    int32_t  is_indexed;  /* the column's isIndexed flag */
    int32_t  obj_count;
    int32_t  data_size;   /* obj_count * 4 or * 8; strings: 4 * (obj_count + 1) + bytes */
    int32_t  pack_size;   /* out */
    int32_t  encode_type; /* out: IXR_ENC_DELTA or IXR_ENC_DICT_COMPRESS */
    int64_t  min_value;   /* out: dpn.minValue */
    int64_t  max_value;   /* out: dpn.maxValue */
    int64_t  sum;         /* out: dpn.sum */
} ixr_encode_desc;
int32_t ixr_encode_run(int32_t device, ixr_encode_desc* descs, int32_t n, const void* d_data, uint64_t data_bytes,
                       void* d_out, uint64_t out_bytes, uint64_t scratch_budget, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IXR_GPU_H */
