/*
 * ixr_synth.h -- deterministic synthetic column generator + pack builder used by bench.py and the tests.
 *
 * Not part of the reference's API: the reference has no data generator beyond its unit tests' random walks.  Columns follow
 * the shapes the benchmark configurations name (monotone ints, categorical ints, run-heavy ints, ms timestamps, sparse
 * longs, k/10 floats and doubles as in indexr-vlt-segment/src/test/java/io/indexr/vlt/codec/dict/DictCodecTest.java:191,
 * category / URL / unique / mostly-empty strings).  Every pack is generated from (seed, column kind, pack id) alone, so any
 * process regenerates identical inputs without shipping data.
 */
#ifndef IXR_SYNTH_H
#define IXR_SYNTH_H
#include <stdint.h>

#include "ixr_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    IXR_COL_I_SEQ = 0,   /* INT    monotone, step U[0,3]                         -> DELTA, ~2 bit            */
    IXR_COL_I_RAND = 1,  /* INT    uniform over the whole int32 range            -> DELTA, 32 bit            */
    IXR_COL_I_CAT = 2,   /* INT    uniform over 1000 values                      -> dictionary, 10-bit ids   */
    IXR_COL_I_RUNS = 3,  /* INT    16 values, geometric runs (mean 32)           -> dictionary, RLE runs     */
    IXR_COL_L_TS = 4,    /* LONG   ms timestamps, monotone, step U[0,2000]       -> dictionary of ~65536, delta-coded */
    IXR_COL_L_CARD = 5,  /* LONG   4096 values spread over 2^40                  -> dictionary, 12-bit ids   */
    IXR_COL_L_RAND = 6,  /* LONG   uniform 64 bit                                -> dictionary of 65536 (worst case) */
    IXR_COL_F_CARD = 7,  /* FLOAT  k/10, k < 10000                               -> dictionary               */
    IXR_COL_D_CARD = 8,  /* DOUBLE k/10, k < 50000                               -> dictionary               */
    IXR_COL_S_CAT = 9,   /* STRING 32 categories of 3..12 bytes                  -> plain dictionary         */
    IXR_COL_S_URL = 10,  /* STRING Zipf(1.1) over 50000 URLs of 20..80 bytes     -> front-coded dictionary   */
    IXR_COL_S_UNIQ = 11, /* STRING unique alphanumerics of 0..24 bytes           -> dictionary of all rows   */
    IXR_COL_S_EMPTY = 12,/* STRING 90 % empty, else 1..16 bytes                  -> dictionary               */
    IXR_COL_KINDS = 13
};

/* Data type (IXR_INT..IXR_STRING) of a column kind. */
int32_t ixr_synth_data_type(int32_t kind);

/* Writes the decoded pack (numbers: raw LE array; strings: StringsStruct) of `rows` rows. Returns its size in bytes, or a
 * negative error when `cap` is too small. */
int32_t ixr_synth_column(int32_t kind, uint64_t seed, uint32_t pack_id, int32_t rows, void* out, int32_t cap);

typedef struct ixr_synth_req {
    int32_t  kind;
    uint32_t pack_id;
    int32_t  rows;
    int32_t  is_indexed;
} ixr_synth_req;

/* Generates and compresses n packs with `threads` host threads.  descs[i] is filled completely: in_off / out_off are
 * assigned back to back (16-byte aligned) in request order.  The compressed arena (readable IXR_IN_SLACK bytes past the
 * last pack, zero padded) is malloc'ed and returned through *arena / *arena_bytes; release it with ixr_synth_free.
 * checksums[i] (optional) receives ixr_checksum of the decoded pack.  Returns 0 or a negative error. */
int32_t ixr_synth_build(const ixr_synth_req* reqs, int32_t n, uint64_t seed, int32_t threads, ixr_pack_desc* descs,
                        uint8_t** arena, uint64_t* arena_bytes, uint64_t* out_bytes, uint64_t* checksums);
void    ixr_synth_free(uint8_t* arena);

/* Position-weighted 64-bit checksum of a byte string: sum over its little-endian 64-bit words w_i (zero padded) of
 * (w_i + 0x9E3779B97F4A7C15) * (2i + 1)  mod 2^64.  The device version lets full-size outputs be verified in place. */
uint64_t ixr_checksum(const void* data, uint64_t bytes);
/* Checksums of all decoded packs of a plan, computed on the device from the output arena (n_packs entries). */
int32_t  ixr_plan_checksums(ixr_plan* plan, const void* d_out, uint64_t out_bytes, uint64_t* checksums, void* stream);

#ifdef __cplusplus
}
#endif
#endif
