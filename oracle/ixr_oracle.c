/*
 * ixr_oracle.c -- CPU oracle for IndexR VLT pack decode/encode.  TEST INFRASTRUCTURE ONLY (see ixr_oracle.h).
 *
 * Plain-C restatement of the reference Java, keeping its algorithmic structure.  Citations are relative to
 * /root/reference.  Where Java would throw (BufferUnderflow, ArrayIndexOutOfBounds, checkState) this code
 * returns IXO_DATA_CORRUPTED instead, so corrupted inputs can be used in tests without crashing.
 * Everything is little-endian (indexr-vlt-segment/.../codec/UnsafeUtil.java:28-32).
 */
#define _GNU_SOURCE
#include "ixr_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "_gen/packing_gen.c" /* iunpack8_tab / ipack8_tab / lunpack8_tab / lpack8_tab, from gen_packing.py */

/* ------------------------------------------------------------------------------------------------
 * little helpers
 * ---------------------------------------------------------------------------------------------- */
static inline int32_t ld32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline int64_t ld64(const uint8_t* p) { int64_t v; memcpy(&v, p, 8); return v; }
static inline void st32(uint8_t* p, int32_t v) { memcpy(p, &v, 4); }
static inline void st64(uint8_t* p, int64_t v) { memcpy(p, &v, 8); }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* VarInteger.bitWidth (indexr-vlt-segment/.../codec/VarInteger.java:13-19) */
static inline int bit_width32(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }
static inline int bit_width64(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }
/* VarInteger.bitWidthToByteWidth (:21-23) */
static inline int byte_width(int bw) { return (bw + 7) >> 3; }
/* PackingUtil.packingOutputBytes (indexr-vlt-segment/.../codec/pack/PackingUtil.java:14-21) */
static inline int packing_output_bytes(int n, int bw) { return (n == 0 || bw == 0) ? 0 : ((n + 7) >> 3) * bw; }

/* A bounded reader standing in for java.nio.ByteBuffer (position/limit). */
typedef struct { const uint8_t* p; int pos; int limit; int err; } rd_t;
static inline int rd_u8(rd_t* r) {
    if (r->pos >= r->limit) { r->err = 1; return 0; }
    return r->p[r->pos++];
}
/* A bounded writer standing in for the output ByteBuffer. */
typedef struct { uint8_t* p; int pos; int limit; int err; } wr_t;
static inline void wr_u8(wr_t* w, int b) {
    if (w->pos >= w->limit) { w->err = 1; return; }
    w->p[w->pos++] = (uint8_t)b;
}

/* VarInteger.readUnsignedVarInt (:68-77). Java int shifts use the low 5 bits of the count. */
static uint32_t rd_uvarint32(rd_t* r) {
    uint32_t value = 0; int i = 0; uint32_t b;
    while (((b = (uint32_t)rd_u8(r)) & 0x80) != 0) {
        if (r->err) return 0;
        value |= (b & 0x7F) << (i & 31);
        i += 7;
    }
    return value | (b << (i & 31));
}
/* VarInteger.readUnsignedVarLong (:87-96). */
static uint64_t rd_uvarint64(rd_t* r) {
    uint64_t value = 0; int i = 0; uint64_t b;
    while (((b = (uint64_t)rd_u8(r)) & 0x80) != 0) {
        if (r->err) return 0;
        value |= (b & 0x7F) << (i & 63);
        i += 7;
    }
    return value | (b << (i & 63));
}
/* VarInteger.readZigZagVarInt (:123-127) */
static int32_t rd_zz32(rd_t* r) {
    int32_t raw = (int32_t)rd_uvarint32(r);
    int32_t temp = ((int32_t)(((int32_t)((uint32_t)raw << 31)) >> 31) ^ raw) >> 1;
    return temp ^ (raw & (int32_t)(1u << 31));
}
/* VarInteger.readZigZagVarLong (:106-110) */
static int64_t rd_zz64(rd_t* r) {
    int64_t raw = (int64_t)rd_uvarint64(r);
    int64_t temp = ((int64_t)(((int64_t)((uint64_t)raw << 63)) >> 63) ^ raw) >> 1;
    return temp ^ (raw & (int64_t)(1ull << 63));
}
/* VarInteger.writeUnsignedVarInt (:79-85) */
static void wr_uvarint32(wr_t* w, uint32_t value) {
    while ((value & 0xFFFFFF80u) != 0) { wr_u8(w, (int)((value & 0x7F) | 0x80)); value >>= 7; }
    wr_u8(w, (int)(value & 0x7F));
}
/* VarInteger.writeUnsignedVarLong (:98-104) */
static void wr_uvarint64(wr_t* w, uint64_t value) {
    while ((value & 0xFFFFFFFFFFFFFF80ull) != 0) { wr_u8(w, (int)((value & 0x7F) | 0x80)); value >>= 7; }
    wr_u8(w, (int)(value & 0x7F));
}
/* VarInteger.writeZigZagVarInt / writeZigZagVarLong (:112-118) */
static void wr_zz32(wr_t* w, int32_t v) { wr_uvarint32(w, ((uint32_t)v << 1) ^ (uint32_t)(v >> 31)); }
static void wr_zz64(wr_t* w, int64_t v) { wr_uvarint64(w, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
/* VarInteger.readFixedByteWidthInt (:49-66) */
static int32_t rd_fixed(rd_t* r, int byteWidth) {
    switch (byteWidth) {
        case 0: return 0;
        case 1: return rd_u8(r);
        case 2: { int a = rd_u8(r), b = rd_u8(r); return a | (b << 8); }
        case 3: { int a = rd_u8(r), b = rd_u8(r), c = rd_u8(r); return a | ((b | (c << 8)) << 8); }
        case 4: { uint32_t a = rd_u8(r), b = rd_u8(r), c = rd_u8(r), d = rd_u8(r); return (int32_t)(a | (b << 8) | (c << 16) | (d << 24)); }
        default: r->err = 1; return 0;
    }
}
/* VarInteger.writeFixedByteWidthInt (:25-47) */
static void wr_fixed(wr_t* w, int byteWidth, int32_t v) {
    uint32_t u = (uint32_t)v;
    for (int i = 0; i < byteWidth; i++) wr_u8(w, (int)((u >> (8 * i)) & 0xFF));
}

/* ------------------------------------------------------------------------------------------------
 * IntPacker / LongPacker (indexr-vlt-segment/.../codec/pack/IntPacker.java:35-70, LongPacker.java:35-71)
 * ---------------------------------------------------------------------------------------------- */
int ixo_pack_ints(int bw, int n, const int32_t* in, uint8_t* out) {
    if (bw < 0 || bw > 32) return IXO_BUG;
    int step = n >> 3;
    for (int s = 0; s < step; s++) ipack8_tab[bw](in + (s << 3), out + s * bw);
    int tail = n & 7;
    if (tail) {
        int32_t tmp[8] = {0};
        int tailStart = (n >> 3) << 3;
        for (int i = 0; i < n - tailStart; i++) tmp[i] = in[tailStart + i];
        ipack8_tab[bw](tmp, out + (n >> 3) * bw);
    }
    return packing_output_bytes(n, bw);
}
int ixo_unpack_ints(int bw, int n, const uint8_t* in, int32_t* out) {
    if (bw < 0 || bw > 32) return IXO_BUG;
    int step = n >> 3;
    for (int s = 0; s < step; s++) iunpack8_tab[bw](in + s * bw, out + (s << 3));
    int tail = n & 7;
    if (tail) {
        int32_t tmp[8] = {0};
        int tailStart = (n >> 3) << 3;
        iunpack8_tab[bw](in + (n >> 3) * bw, tmp);
        for (int i = 0; i < n - tailStart; i++) out[tailStart + i] = tmp[i];
    }
    return packing_output_bytes(n, bw);
}
int ixo_pack_longs(int bw, int n, const int64_t* in, uint8_t* out) {
    if (bw < 0 || bw > 64) return IXO_BUG;
    int step = n >> 3;
    for (int s = 0; s < step; s++) lpack8_tab[bw](in + (s << 3), out + s * bw);
    int tail = n & 7;
    if (tail) {
        int64_t tmp[8] = {0};
        int tailStart = (n >> 3) << 3;
        for (int i = 0; i < n - tailStart; i++) tmp[i] = in[tailStart + i];
        lpack8_tab[bw](tmp, out + (n >> 3) * bw);
    }
    return packing_output_bytes(n, bw);
}
int ixo_unpack_longs(int bw, int n, const uint8_t* in, int64_t* out) {
    if (bw < 0 || bw > 64) return IXO_BUG;
    int step = n >> 3;
    for (int s = 0; s < step; s++) lunpack8_tab[bw](in + s * bw, out + (s << 3));
    int tail = n & 7;
    if (tail) {
        int64_t tmp[8] = {0};
        int tailStart = (n >> 3) << 3;
        lunpack8_tab[bw](in + (n >> 3) * bw, tmp);
        for (int i = 0; i < n - tailStart; i++) out[tailStart + i] = tmp[i];
    }
    return packing_output_bytes(n, bw);
}

/* ------------------------------------------------------------------------------------------------
 * DeltaCodec (indexr-vlt-segment/.../codec/delta/DeltaCodec.java)
 * ---------------------------------------------------------------------------------------------- */
#define BLOCK_SPLIT 128
#define MINI_SPLIT 32

/* encodeIntsMiniBlock :82-97 */
static void enc_ints_mini(int from, int to, const int32_t* delta, wr_t* w) {
    uint32_t mask = 0;
    for (int i = from; i < to; i++) mask |= (uint32_t)delta[i];
    int bw = bit_width32(mask);
    wr_u8(w, bw);
    int bytes = packing_output_bytes(to - from, bw);
    if (w->pos + bytes > w->limit) { w->err = 1; return; }
    ixo_pack_ints(bw, to - from, delta + from, w->p + w->pos);
    w->pos += bytes;
}
/* encodeIntsBlock :57-80 */
static int32_t enc_ints_block(int32_t prev, int count, const int32_t* in, wr_t* w, int32_t* delta) {
    int32_t minDelta = INT32_MAX;
    for (int i = 0; i < count; i++) {
        int32_t v = in[i];
        int32_t d = (int32_t)((uint32_t)v - (uint32_t)prev);
        delta[i] = d;
        if (d < minDelta) minDelta = d;
        prev = v;
    }
    wr_zz32(w, minDelta);
    for (int i = 0; i < count; i++) delta[i] = (int32_t)((uint32_t)delta[i] - (uint32_t)minDelta);
    for (int i = 0; i < count; i += MINI_SPLIT) enc_ints_mini(i, imin(i + MINI_SPLIT, count), delta, w);
    return prev;
}
/* encodeInts :40-55 */
int ixo_delta_encode_ints(int n, const int32_t* in, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (out_limit < 4) return IXO_BUFFER_OVERFLOW;
    wr_t w = {out + 4, 0, out_limit - 4, 0};
    int32_t prev = in[0];
    wr_zz32(&w, prev);
    int32_t delta[BLOCK_SPLIT];
    for (int i = 0; i < n; i += BLOCK_SPLIT) {
        int bs = imin(i + BLOCK_SPLIT, n) - i;
        prev = enc_ints_block(prev, bs, in + i, &w, delta);
        if (w.err) return IXO_BUFFER_OVERFLOW;
    }
    if (w.err) return IXO_BUFFER_OVERFLOW;
    st32(out, w.pos);
    return 4 + w.pos;
}
/* decodeIntsMiniBlock :129-141 */
static void dec_ints_mini(int count, rd_t* r, int32_t* out) {
    int bw = (int8_t)rd_u8(r);
    if (r->err) return;
    if (bw < 0 || bw > 32) { r->err = 1; return; }
    int bytes = packing_output_bytes(count, bw);
    if (r->pos + bytes > r->limit) { r->err = 1; return; }
    ixo_unpack_ints(bw, count, r->p + r->pos, out);
    r->pos += bytes;
}
/* decodeIntsBlock :112-127 -- two passes: unpack the deltas in place, then the running sum */
static int32_t dec_ints_block(int32_t prev, int count, rd_t* r, int32_t* out) {
    int32_t minDelta = rd_zz32(r);
    for (int i = 0; i < count && !r->err; i += MINI_SPLIT) dec_ints_mini(imin(i + MINI_SPLIT, count) - i, r, out + i);
    if (r->err) return prev;
    for (int i = 0; i < count; i++) {
        int32_t v = (int32_t)((uint32_t)out[i] + (uint32_t)minDelta + (uint32_t)prev);
        out[i] = v;
        prev = v;
    }
    return prev;
}
/* decodeInts :100-110 (decodeFloats :253-255 is the same routine on the raw bits) */
int ixo_delta_decode_ints(int n, const uint8_t* in, int in_limit, int32_t* out) {
    if (n == 0) return 0;
    if (in_limit < 4) return IXO_DATA_CORRUPTED;
    int32_t encodedLen = ld32(in);
    if (encodedLen < 0 || (int64_t)encodedLen + 4 > in_limit) return IXO_DATA_CORRUPTED;
    rd_t r = {in + 4, 0, encodedLen, 0};
    int32_t prev = rd_zz32(&r);
    for (int i = 0; i < n && !r.err; i += BLOCK_SPLIT) {
        int bs = imin(i + BLOCK_SPLIT, n) - i;
        prev = dec_ints_block(prev, bs, &r, out + i);
    }
    if (r.err) return IXO_DATA_CORRUPTED;
    return 4 + r.pos;
}

/* encodeLongsMiniBlock :186-201 */
static void enc_longs_mini(int from, int to, const int64_t* delta, wr_t* w) {
    uint64_t mask = 0;
    for (int i = from; i < to; i++) mask |= (uint64_t)delta[i];
    int bw = bit_width64(mask);
    wr_u8(w, bw);
    int bytes = packing_output_bytes(to - from, bw);
    if (w->pos + bytes > w->limit) { w->err = 1; return; }
    ixo_pack_longs(bw, to - from, delta + from, w->p + w->pos);
    w->pos += bytes;
}
/* encodeLongsBlock :161-184 */
static int64_t enc_longs_block(int64_t prev, int count, const int64_t* in, wr_t* w, int64_t* delta) {
    int64_t minDelta = INT64_MAX;
    for (int i = 0; i < count; i++) {
        int64_t v = in[i];
        int64_t d = (int64_t)((uint64_t)v - (uint64_t)prev);
        delta[i] = d;
        if (d < minDelta) minDelta = d;
        prev = v;
    }
    wr_zz64(w, minDelta);
    for (int i = 0; i < count; i++) delta[i] = (int64_t)((uint64_t)delta[i] - (uint64_t)minDelta);
    for (int i = 0; i < count; i += MINI_SPLIT) enc_longs_mini(i, imin(i + MINI_SPLIT, count), delta, w);
    return prev;
}
/* encodeLongs :144-159 */
int ixo_delta_encode_longs(int n, const int64_t* in, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (out_limit < 4) return IXO_BUFFER_OVERFLOW;
    wr_t w = {out + 4, 0, out_limit - 4, 0};
    int64_t prev = in[0];
    wr_zz64(&w, prev);
    int64_t delta[BLOCK_SPLIT];
    for (int i = 0; i < n; i += BLOCK_SPLIT) {
        int bs = imin(i + BLOCK_SPLIT, n) - i;
        prev = enc_longs_block(prev, bs, in + i, &w, delta);
        if (w.err) return IXO_BUFFER_OVERFLOW;
    }
    if (w.err) return IXO_BUFFER_OVERFLOW;
    st32(out, w.pos);
    return 4 + w.pos;
}
/* decodeLongsMiniBlock :233-245 */
static void dec_longs_mini(int count, rd_t* r, int64_t* out) {
    int bw = (int8_t)rd_u8(r);
    if (r->err) return;
    if (bw < 0 || bw > 64) { r->err = 1; return; }
    int bytes = packing_output_bytes(count, bw);
    if (r->pos + bytes > r->limit) { r->err = 1; return; }
    ixo_unpack_longs(bw, count, r->p + r->pos, out);
    r->pos += bytes;
}
/* decodeLongsBlock :216-231 */
static int64_t dec_longs_block(int64_t prev, int count, rd_t* r, int64_t* out) {
    int64_t minDelta = rd_zz64(r);
    for (int i = 0; i < count && !r->err; i += MINI_SPLIT) dec_longs_mini(imin(i + MINI_SPLIT, count) - i, r, out + i);
    if (r->err) return prev;
    for (int i = 0; i < count; i++) {
        int64_t v = (int64_t)((uint64_t)out[i] + (uint64_t)minDelta + (uint64_t)prev);
        out[i] = v;
        prev = v;
    }
    return prev;
}
/* decodeLongs :204-214 (decodeDoubles :262-265 is the same routine on the raw bits).
 * `out` may be unaligned (a DictStruct puts entries at +4), so decode through memcpy. */
int ixo_delta_decode_longs(int n, const uint8_t* in, int in_limit, int64_t* out_) {
    if (n == 0) return 0;
    if (in_limit < 4) return IXO_DATA_CORRUPTED;
    int32_t encodedLen = ld32(in);
    if (encodedLen < 0 || (int64_t)encodedLen + 4 > in_limit) return IXO_DATA_CORRUPTED;
    rd_t r = {in + 4, 0, encodedLen, 0};
    int64_t prev = rd_zz64(&r);
    int64_t blk[BLOCK_SPLIT];
    uint8_t* out = (uint8_t*)out_;
    for (int i = 0; i < n && !r.err; i += BLOCK_SPLIT) {
        int bs = imin(i + BLOCK_SPLIT, n) - i;
        prev = dec_longs_block(prev, bs, &r, blk);
        if (!r.err) memcpy(out + (size_t)i * 8, blk, (size_t)bs * 8);
    }
    if (r.err) return IXO_DATA_CORRUPTED;
    return 4 + r.pos;
}

/* StringsStruct accessors (indexr-common/.../data/StringsStruct.java:35-59) */
static inline int ss_total_len(int count, const uint8_t* s) { return ((count + 1) << 2) + ld32(s + ((size_t)count << 2)); }

/* encodeStrings :268-352 (front coding) */
int ixo_delta_encode_strings(int n, const uint8_t* ss, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    int32_t* prefixLen = (int32_t*)malloc((size_t)n * 4);
    int32_t* deltaLen = (int32_t*)malloc((size_t)n * 4);
    const uint8_t* body = ss + ((size_t)(n + 1) << 2);
    const uint8_t* prevp = NULL; int prevlen = 0;
    for (int i = 0; i < n; i++) {
        int so = ld32(ss + ((size_t)i << 2)), eo = ld32(ss + ((size_t)(i + 1) << 2));
        const uint8_t* vp = body + so; int vlen = eo - so;
        int minLen = imin(prevlen, vlen), p = 0;
        while (p < minLen && vp[p] == prevp[p]) p++;
        prevp = vp; prevlen = vlen;
        prefixLen[i] = p;
        deltaLen[i] = vlen - p;
    }
    int ret;
    if (out_limit < 4) { ret = IXO_BUFFER_OVERFLOW; goto done; }
    uint8_t* dataOut = out + 4;
    int remain = out_limit - 4;
    int s1 = ixo_delta_encode_ints(n, prefixLen, dataOut, remain);
    if (s1 < 0) { ret = s1; goto done; }
    int s2 = ixo_delta_encode_ints(n, deltaLen, dataOut + s1, remain - s1);
    if (s2 < 0) { ret = s2; goto done; }
    int pos = s1 + s2, total = 0;
    for (int i = 0; i < n; i++) {
        int so = ld32(ss + ((size_t)i << 2));
        if (pos + deltaLen[i] > remain) { ret = IXO_BUFFER_OVERFLOW; goto done; }
        memcpy(dataOut + pos, body + so + prefixLen[i], (size_t)deltaLen[i]);
        pos += deltaLen[i];
        total += deltaLen[i];
    }
    st32(out, s1 + s2 + total);
    ret = 4 + s1 + s2 + total;
done:
    free(prefixLen); free(deltaLen);
    return ret;
}
/* decodeStrings :355-388, with StringsStruct.Builder.addString(pre, sub, pointer) (StringsStruct.java:135-158) */
int ixo_delta_decode_strings(int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (in_limit < 4) return IXO_DATA_CORRUPTED;
    int32_t* prefixLen = (int32_t*)malloc((size_t)n * 4);
    int32_t* deltaLen = (int32_t*)malloc((size_t)n * 4);
    int ret;
    const uint8_t* data = in + 4;
    int remain = in_limit - 4;
    int s1 = ixo_delta_decode_ints(n, data, remain, prefixLen);
    if (s1 < 0) { ret = s1; goto done; }
    int s2 = ixo_delta_decode_ints(n, data + s1, remain - s1, deltaLen);
    if (s2 < 0) { ret = s2; goto done; }
    int indexLen = (n + 1) << 2;
    if (indexLen > out_limit) { ret = IXO_BUFFER_OVERFLOW; goto done; }
    int pos = s1 + s2, total = 0, lastEnd = 0;
    const uint8_t* prevp = NULL; int prevlen = 0;
    for (int i = 0; i < n; i++) {
        int pl = prefixLen[i], dl = deltaLen[i];
        if (pl < 0 || dl < 0 || pl > prevlen || pos + dl > remain) { ret = IXO_DATA_CORRUPTED; goto done; }
        int start = lastEnd;
        if ((int64_t)indexLen + start + pl + dl > out_limit) { ret = IXO_BUFFER_OVERFLOW; goto done; }
        st32(out + ((size_t)i << 2), start);
        uint8_t* dst = out + indexLen + start;
        if (pl) memcpy(dst, prevp, (size_t)pl);
        if (dl) memcpy(dst + pl, data + pos, (size_t)dl);
        prevp = dst; prevlen = pl + dl;
        lastEnd = start + pl + dl;
        total += dl; pos += dl;
    }
    st32(out + ((size_t)n << 2), lastEnd);
    ret = 4 + s1 + s2 + total;
done:
    free(prefixLen); free(deltaLen);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * RLEPackingHybridCodec (indexr-vlt-segment/.../codec/rle/RLEPackingHybridCodec.java)
 * ---------------------------------------------------------------------------------------------- */
#define RLE_THRESHOLD 8
/* packingEncode :74-88 */
static void rle_packing_encode(int bw, int from, int to, const int32_t* in, wr_t* w) {
    if (from == to) return;
    int count = to - from;
    wr_uvarint32(w, ((uint32_t)count << 1) | 1);
    int bytes = packing_output_bytes(count, bw);
    if (w->pos + bytes > w->limit) { w->err = 1; return; }
    ixo_pack_ints(bw, count, in + from, w->p + w->pos);
    w->pos += bytes;
}
/* encodeInts :38-71, rleEncode :91-94 */
int ixo_rle_encode_ints(int bw, int n, const int32_t* in, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (out_limit < 4) return IXO_BUFFER_OVERFLOW;
    wr_t w = {out + 4, 0, out_limit - 4, 0};
    int bytew = byte_width(bw);
    int curPackingChunkStart = 0, curRepeatChunkStart = 0;
    int32_t curRepeatVal = in[0];
    for (int i = 0; i < n; i++) {
        int32_t v = in[i];
        if (v < 0) return IXO_DATA_IMPROPER;
        if (v != curRepeatVal) {
            int repeat = i - curRepeatChunkStart;
            if (repeat >= RLE_THRESHOLD) {
                rle_packing_encode(bw, curPackingChunkStart, curRepeatChunkStart, in, &w);
                wr_uvarint32(&w, (uint32_t)repeat << 1);
                wr_fixed(&w, bytew, curRepeatVal);
                curPackingChunkStart = i;
            }
            curRepeatChunkStart = i;
            curRepeatVal = v;
        }
    }
    rle_packing_encode(bw, curPackingChunkStart, n, in, &w);
    if (w.err) return IXO_BUFFER_OVERFLOW;
    st32(out, w.pos);
    return 4 + w.pos;
}
/* decodeInts :97-128 */
int ixo_rle_decode_ints(int bw, int n, const uint8_t* in, int in_limit, int32_t* out) {
    if (n == 0) return 0;
    if (bw < 0 || bw > 32 || in_limit < 4) return IXO_DATA_CORRUPTED;
    int32_t encodedLen = ld32(in);
    if (encodedLen < 0 || (int64_t)encodedLen + 4 > in_limit) return IXO_DATA_CORRUPTED;
    rd_t r = {in + 4, 0, encodedLen, 0};
    int bytew = byte_width(bw);
    int valIndex = 0;
    while (r.pos < r.limit) {
        uint32_t header = rd_uvarint32(&r);
        if (r.err) return IXO_DATA_CORRUPTED;
        if ((header & 1) == 0) {
            int repeat = (int)(header >> 1);
            int32_t val = rd_fixed(&r, bytew);
            if (r.err || repeat > n - valIndex) return IXO_DATA_CORRUPTED;
            for (int i = 0; i < repeat; i++) out[valIndex + i] = val;
            valIndex += repeat;
        } else {
            int count = (int)(header >> 1);
            int bytes = packing_output_bytes(count, bw);
            if (count > n - valIndex || r.pos + bytes > r.limit) return IXO_DATA_CORRUPTED;
            ixo_unpack_ints(bw, count, r.p + r.pos, out + valIndex);
            r.pos += bytes;
            valIndex += count;
        }
    }
    if (valIndex != n) return IXO_DATA_CORRUPTED; /* Preconditions.checkState(valIndex == valCount) :126 */
    return 4 + r.pos;
}

/* ------------------------------------------------------------------------------------------------
 * DictStruct (indexr-common/.../data/DictStruct.java:14-110)
 *   | entryCount(4) | dict | dataSize(4) | ids |
 * ---------------------------------------------------------------------------------------------- */
static inline int is_number(int t) { return t != IXO_STRING; }
static inline int num_shift(int t) { return (t == IXO_INT || t == IXO_FLOAT) ? 2 : 3; }

static int ds_dict_size(int data_type, const uint8_t* ds) {
    int entryCount = ld32(ds);
    if (is_number(data_type)) return entryCount << num_shift(data_type);
    return ss_total_len(entryCount, ds + 4);
}
int ixo_dict_struct_size(int data_type, const uint8_t* ds) {
    int dictSize = ds_dict_size(data_type, ds);
    return 4 + dictSize + 4 + ld32(ds + 4 + dictSize);
}

/* DictStruct.decodeInts/Longs/Floats/Doubles/Strings :57-110 (== DictCodec.decode* DictCodec.java:134-151 ...) */
int ixo_data_from_dict_struct(int data_type, int n, const uint8_t* ds, int ds_size, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (ds_size < 8) return IXO_DATA_CORRUPTED;
    int entryCount = ld32(ds);
    if (entryCount < 0) return IXO_DATA_CORRUPTED;
    const uint8_t* entries = ds + 4;
    if (is_number(data_type)) {
        int sh = num_shift(data_type);
        int64_t dictSize = (int64_t)entryCount << sh;
        if (4 + dictSize + 4 + ((int64_t)n << 2) > ds_size) return IXO_DATA_CORRUPTED;
        if (((int64_t)n << sh) > out_limit) return IXO_BUFFER_OVERFLOW;
        const uint8_t* ids = ds + 4 + dictSize + 4;
        for (int i = 0; i < n; i++) {
            uint32_t vid = (uint32_t)ld32(ids + ((size_t)i << 2));
            if (vid >= (uint32_t)entryCount) return IXO_DATA_CORRUPTED;
            if (sh == 2) st32(out + ((size_t)i << 2), ld32(entries + ((size_t)vid << 2)));
            else st64(out + ((size_t)i << 3), ld64(entries + ((size_t)vid << 3)));
        }
        return n << sh;
    } else {
        if (4 + ((int64_t)(entryCount + 1) << 2) > ds_size) return IXO_DATA_CORRUPTED;
        int dictSize = ss_total_len(entryCount, entries);
        if (dictSize < 0 || 4 + (int64_t)dictSize + 4 + ((int64_t)n << 2) > ds_size) return IXO_DATA_CORRUPTED;
        const uint8_t* ids = ds + 4 + dictSize + 4;
        const uint8_t* dbody = entries + ((size_t)(entryCount + 1) << 2);
        int indexLen = (n + 1) << 2;
        if (indexLen > out_limit) return IXO_BUFFER_OVERFLOW;
        int lastEnd = 0;
        for (int i = 0; i < n; i++) { /* StringsStruct.Builder.addString :120-131 */
            uint32_t vid = (uint32_t)ld32(ids + ((size_t)i << 2));
            if (vid >= (uint32_t)entryCount) return IXO_DATA_CORRUPTED;
            int so = ld32(entries + ((size_t)vid << 2)), eo = ld32(entries + ((size_t)(vid + 1) << 2));
            int len = eo - so;
            if (len < 0 || (int64_t)indexLen + lastEnd + len > out_limit) return IXO_BUFFER_OVERFLOW;
            st32(out + ((size_t)i << 2), lastEnd);
            memcpy(out + indexLen + lastEnd, dbody + so, (size_t)len);
            lastEnd += len;
        }
        st32(out + ((size_t)n << 2), lastEnd);
        return indexLen + lastEnd;
    }
}

/* ---- DictCodec.encode* (indexr-vlt-segment/.../codec/dict/DictCodec.java:79-131,154-206,229-281,304-356,379-433) ---- */
/* java.lang.Float.compare / Double.compare ordering: -0.0 < 0.0, NaN greatest, all NaN equal. */
static int java_float_compare(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    uint32_t ab, bb; memcpy(&ab, &a, 4); memcpy(&bb, &b, 4);
    if (a != a) ab = 0x7fc00000u;
    if (b != b) bb = 0x7fc00000u;
    return (ab == bb) ? 0 : ((int32_t)ab < (int32_t)bb ? -1 : 1);
}
static int java_double_compare(double a, double b) {
    if (a < b) return -1;
    if (a > b) return 1;
    uint64_t ab, bb; memcpy(&ab, &a, 8); memcpy(&bb, &b, 8);
    if (a != a) ab = 0x7ff8000000000000ull;
    if (b != b) bb = 0x7ff8000000000000ull;
    return (ab == bb) ? 0 : ((int64_t)ab < (int64_t)bb ? -1 : 1);
}
static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }
static int cmp_i64(const void* a, const void* b) { int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return (x > y) - (x < y); }
/* (value, first index) pairs so that, among compare-equal keys (NaNs), the first one inserted is kept */
typedef struct { float v; int idx; } fpair_t;
typedef struct { double v; int idx; } dpair_t;
static int cmp_fpair(const void* a, const void* b) {
    const fpair_t* x = (const fpair_t*)a; const fpair_t* y = (const fpair_t*)b;
    int c = java_float_compare(x->v, y->v);
    return c ? c : (x->idx > y->idx) - (x->idx < y->idx);
}
static int cmp_dpair(const void* a, const void* b) {
    const dpair_t* x = (const dpair_t*)a; const dpair_t* y = (const dpair_t*)b;
    int c = java_double_compare(x->v, y->v);
    return c ? c : (x->idx > y->idx) - (x->idx < y->idx);
}
/* BytesUtil.compareBytes (indexr-common/.../util/BytesUtil.java:7-19): unsigned bytes, then length */
static int compare_bytes(const uint8_t* a, int la, const uint8_t* b, int lb) {
    int len = imin(la, lb);
    for (int i = 0; i < len; i++) if (a[i] != b[i]) return (int)a[i] - (int)b[i];
    return la - lb;
}
typedef struct { const uint8_t* p; int len; } str_t;
static int cmp_str(const void* a, const void* b) {
    const str_t* x = (const str_t*)a; const str_t* y = (const str_t*)b;
    return compare_bytes(x->p, x->len, y->p, y->len);
}
static int max_dict_entries(int n, float ratio) {
    volatile float prod = (float)n * ratio; /* Java: int * float -> float, then Math.ceil(double) */
    return (int)ceil((double)prod);
}

int ixo_dict_encode(int data_type, float ratio, int n, const uint8_t* data, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    int maxEntries = max_dict_entries(n, ratio);
    int32_t* ids = (int32_t*)malloc((size_t)n * 4);
    int ret = IXO_BUG;
    int entryCount = 0, dictSize = 0;
    uint8_t* entriesOut = out + 4;
    switch (data_type) {
        case IXO_INT: {
            int32_t* e = (int32_t*)malloc((size_t)n * 4);
            memcpy(e, data, (size_t)n * 4);
            qsort(e, (size_t)n, 4, cmp_i32);
            int m = 0;
            for (int i = 0; i < n; i++) if (m == 0 || e[m - 1] != e[i]) e[m++] = e[i];
            if (m > maxEntries) { free(e); ret = IXO_DATA_IMPROPER; goto done; }
            for (int i = 0; i < n; i++) { /* IntArrays.binarySearch */
                int32_t v = ld32(data + ((size_t)i << 2));
                int lo = 0, hi = m - 1, found = -1;
                while (lo <= hi) { int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (e[mid] < v) lo = mid + 1; else if (e[mid] > v) hi = mid - 1; else { found = mid; break; } }
                ids[i] = found;
            }
            entryCount = m; dictSize = m << 2;
            if (4 + dictSize > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
            memcpy(entriesOut, e, (size_t)dictSize);
            free(e);
            break;
        }
        case IXO_LONG: {
            int64_t* e = (int64_t*)malloc((size_t)n * 8);
            memcpy(e, data, (size_t)n * 8);
            qsort(e, (size_t)n, 8, cmp_i64);
            int m = 0;
            for (int i = 0; i < n; i++) if (m == 0 || e[m - 1] != e[i]) e[m++] = e[i];
            if (m > maxEntries) { free(e); ret = IXO_DATA_IMPROPER; goto done; }
            for (int i = 0; i < n; i++) {
                int64_t v = ld64(data + ((size_t)i << 3));
                int lo = 0, hi = m - 1, found = -1;
                while (lo <= hi) { int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (e[mid] < v) lo = mid + 1; else if (e[mid] > v) hi = mid - 1; else { found = mid; break; } }
                ids[i] = found;
            }
            entryCount = m; dictSize = m << 3;
            if (4 + dictSize > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
            memcpy(entriesOut, e, (size_t)dictSize);
            free(e);
            break;
        }
        case IXO_FLOAT: {
            fpair_t* p = (fpair_t*)malloc((size_t)n * sizeof(fpair_t));
            for (int i = 0; i < n; i++) { memcpy(&p[i].v, data + ((size_t)i << 2), 4); p[i].idx = i; }
            qsort(p, (size_t)n, sizeof(fpair_t), cmp_fpair);
            float* e = (float*)malloc((size_t)n * 4);
            int m = 0;
            for (int i = 0; i < n; i++) if (m == 0 || java_float_compare(e[m - 1], p[i].v) != 0) e[m++] = p[i].v;
            free(p);
            if (m > maxEntries) { free(e); ret = IXO_DATA_IMPROPER; goto done; }
            for (int i = 0; i < n; i++) { /* FloatArrays.binarySearch: plain < and > */
                float v; memcpy(&v, data + ((size_t)i << 2), 4);
                int lo = 0, hi = m - 1, found = -1;
                while (lo <= hi) { int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (e[mid] < v) lo = mid + 1; else if (e[mid] > v) hi = mid - 1; else { found = mid; break; } }
                if (found < 0) { free(e); ret = IXO_BUG; goto done; }
                ids[i] = found;
            }
            entryCount = m; dictSize = m << 2;
            if (4 + dictSize > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
            memcpy(entriesOut, e, (size_t)dictSize);
            free(e);
            break;
        }
        case IXO_DOUBLE: {
            dpair_t* p = (dpair_t*)malloc((size_t)n * sizeof(dpair_t));
            for (int i = 0; i < n; i++) { memcpy(&p[i].v, data + ((size_t)i << 3), 8); p[i].idx = i; }
            qsort(p, (size_t)n, sizeof(dpair_t), cmp_dpair);
            double* e = (double*)malloc((size_t)n * 8);
            int m = 0;
            for (int i = 0; i < n; i++) if (m == 0 || java_double_compare(e[m - 1], p[i].v) != 0) e[m++] = p[i].v;
            free(p);
            if (m > maxEntries) { free(e); ret = IXO_DATA_IMPROPER; goto done; }
            for (int i = 0; i < n; i++) {
                double v; memcpy(&v, data + ((size_t)i << 3), 8);
                int lo = 0, hi = m - 1, found = -1;
                while (lo <= hi) { int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (e[mid] < v) lo = mid + 1; else if (e[mid] > v) hi = mid - 1; else { found = mid; break; } }
                if (found < 0) { free(e); ret = IXO_BUG; goto done; }
                ids[i] = found;
            }
            entryCount = m; dictSize = m << 3;
            if (4 + dictSize > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
            memcpy(entriesOut, e, (size_t)dictSize);
            free(e);
            break;
        }
        case IXO_STRING: {
            const uint8_t* body = data + ((size_t)(n + 1) << 2);
            str_t* e = (str_t*)malloc((size_t)n * sizeof(str_t));
            for (int i = 0; i < n; i++) {
                int so = ld32(data + ((size_t)i << 2)), eo = ld32(data + ((size_t)(i + 1) << 2));
                e[i].p = body + so; e[i].len = eo - so;
            }
            qsort(e, (size_t)n, sizeof(str_t), cmp_str);
            int m = 0;
            for (int i = 0; i < n; i++) if (m == 0 || cmp_str(&e[m - 1], &e[i]) != 0) e[m++] = e[i];
            if (m > maxEntries) { free(e); ret = IXO_DATA_IMPROPER; goto done; }
            for (int i = 0; i < n; i++) { /* Arrays.binarySearch(entries, key) via compareTo */
                int so = ld32(data + ((size_t)i << 2)), eo = ld32(data + ((size_t)(i + 1) << 2));
                str_t key = {body + so, eo - so};
                int lo = 0, hi = m - 1, found = -1;
                while (lo <= hi) { int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); int c = cmp_str(&e[mid], &key); if (c < 0) lo = mid + 1; else if (c > 0) hi = mid - 1; else { found = mid; break; } }
                ids[i] = found;
            }
            /* StringsStruct.from(entries, outputAddr + 4) :89-95 */
            int64_t total = 0;
            for (int i = 0; i < m; i++) total += e[i].len;
            int64_t ds = ((int64_t)(m + 1) << 2) + total;
            if (4 + ds > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
            int lastEnd = 0;
            uint8_t* ebody = entriesOut + ((size_t)(m + 1) << 2);
            for (int i = 0; i < m; i++) {
                st32(entriesOut + ((size_t)i << 2), lastEnd);
                memcpy(ebody + lastEnd, e[i].p, (size_t)e[i].len);
                lastEnd += e[i].len;
            }
            st32(entriesOut + ((size_t)m << 2), lastEnd);
            entryCount = m; dictSize = (int)ds;
            free(e);
            break;
        }
        default: goto done;
    }
    {
        int dataSize = n << 2;
        if ((int64_t)4 + dictSize + 4 + dataSize > out_limit) { ret = IXO_BUFFER_OVERFLOW; goto done; }
        st32(out, entryCount);
        st32(out + 4 + dictSize, dataSize);
        memcpy(out + 4 + dictSize + 4, ids, (size_t)dataSize);
        ret = 4 + dictSize + 4 + dataSize;
    }
done:
    free(ids);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * PlainCodec (indexr-vlt-segment/.../codec/plain/PlainCodec.java:18-77) folded into the callers below.
 * DictStructUtil (indexr-vlt-segment/.../codec/dict/DictStructUtil.java)
 *   | entryCount(4) | isDictEntryCompressed(1) | entries (Delta or Plain) | RLE-hybrid ids |
 * ---------------------------------------------------------------------------------------------- */
/* compress :25-81 */
int ixo_dict_struct_compress(int data_type, int n, const uint8_t* ds, uint8_t* out, int out_limit) {
    int entryCount = ld32(ds);
    int dictSize = ds_dict_size(data_type, ds);
    const uint8_t* entries = ds + 4;
    const uint8_t* ids = ds + 4 + dictSize + 4;
    int compressed = entryCount >= 64 && dictSize >= 256;
    if (out_limit < 5) return IXO_BUFFER_OVERFLOW;
    st32(out, entryCount);
    out[4] = (uint8_t)(compressed ? 1 : 0);
    uint8_t* eout = out + 5;
    int elimit = out_limit - 5;
    int es;
    if (compressed) {
        switch (data_type) {
            case IXO_INT: case IXO_FLOAT: {
                /* entries sit at +4: aligned for int32 */
                es = ixo_delta_encode_ints(entryCount, (const int32_t*)entries, eout, elimit);
                break;
            }
            case IXO_LONG: case IXO_DOUBLE: {
                int64_t* tmp = (int64_t*)malloc((size_t)entryCount * 8);
                memcpy(tmp, entries, (size_t)entryCount * 8);
                es = ixo_delta_encode_longs(entryCount, tmp, eout, elimit);
                free(tmp);
                break;
            }
            case IXO_STRING: es = ixo_delta_encode_strings(entryCount, entries, eout, elimit); break;
            default: return IXO_BUG;
        }
    } else {
        es = dictSize; /* PlainCodec.encode*: numbers n<<2|3, strings (n+1)*4 + total */
        if (es > elimit) return IXO_BUFFER_OVERFLOW;
        memcpy(eout, entries, (size_t)es);
    }
    if (es < 0) return es;
    int bw = bit_width32((uint32_t)(entryCount - 1));
    int32_t* idtmp = (int32_t*)malloc((size_t)n * 4);
    memcpy(idtmp, ids, (size_t)n * 4);
    int dsz = ixo_rle_encode_ints(bw, n, idtmp, eout + es, elimit - es);
    free(idtmp);
    /* The reference does not check dataEncodedSize (:75-80); a negative value cannot happen for valid ids. */
    if (dsz < 0) return dsz;
    return 4 + 1 + es + dsz;
}
/* decompress :83-142 */
int ixo_dict_struct_decompress(int data_type, int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit) {
    if (in_limit < 5 || out_limit < 8) return IXO_DATA_CORRUPTED;
    int entryCount = ld32(in);
    int compressed = in[4] != 0;
    if (entryCount < 0) return IXO_DATA_CORRUPTED;
    st32(out, entryCount);
    const uint8_t* ein = in + 5;
    int elimit = in_limit - 5;
    int consumed; int64_t dictSize;
    if (is_number(data_type)) {
        int sh = num_shift(data_type);
        dictSize = (int64_t)entryCount << sh;
        if (4 + dictSize + 4 + ((int64_t)n << 2) > out_limit) return IXO_BUFFER_OVERFLOW;
        if (compressed) {
            if (sh == 2) consumed = ixo_delta_decode_ints(entryCount, ein, elimit, (int32_t*)(out + 4));
            else consumed = ixo_delta_decode_longs(entryCount, ein, elimit, (int64_t*)(out + 4));
        } else {
            if (dictSize > elimit) return IXO_DATA_CORRUPTED;
            memcpy(out + 4, ein, (size_t)dictSize);
            consumed = (int)dictSize;
        }
    } else {
        if (compressed) {
            consumed = ixo_delta_decode_strings(entryCount, ein, elimit, out + 4, out_limit - 4);
        } else { /* PlainCodec.decodeStrings :73-77 */
            if (((int64_t)(entryCount + 1) << 2) > elimit) return IXO_DATA_CORRUPTED;
            int total = ss_total_len(entryCount, ein);
            if (total < 0 || total > elimit) return IXO_DATA_CORRUPTED;
            if (4 + (int64_t)total > out_limit) return IXO_BUFFER_OVERFLOW;
            memcpy(out + 4, ein, (size_t)total);
            consumed = total;
        }
        if (consumed < 0) return consumed;
        dictSize = ss_total_len(entryCount, out + 4);
        if (4 + dictSize + 4 + ((int64_t)n << 2) > out_limit) return IXO_BUFFER_OVERFLOW;
    }
    if (consumed < 0) return consumed;
    int bw = bit_width32((uint32_t)(entryCount - 1));
    st32(out + 4 + dictSize, n << 2);
    int32_t* idtmp = (int32_t*)malloc((size_t)n * 4 + 4);
    int dc = ixo_rle_decode_ints(bw, n, ein + consumed, elimit - consumed, idtmp);
    if (dc >= 0) memcpy(out + 4 + dictSize + 4, idtmp, (size_t)n * 4);
    free(idtmp);
    if (dc < 0) return dc;
    return 4 + 1 + consumed + dc;
}

/* ------------------------------------------------------------------------------------------------
 * SimpleDictCodec (indexr-vlt-segment/.../codec/dict/SimpleDictCodec.java)
 *   | entryCount(4) | plain entries | dataSize(4) | RLE-hybrid ids |
 * ---------------------------------------------------------------------------------------------- */
/* encode* :68-128 (numbers: sorted entries), :450-512 (strings: entries in first-seen order) */
int ixo_simple_dict_encode(int data_type, int n, const uint8_t* data, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    uint8_t* ds = (uint8_t*)malloc((size_t)out_limit + 64);
    int ret;
    int entryCount, dictSize;
    int32_t* ids = (int32_t*)malloc((size_t)n * 4);
    if (is_number(data_type)) {
        int dsz = ixo_dict_encode(data_type, 1.0f, n, data, ds, out_limit + 64);
        if (dsz < 0) { ret = dsz; goto done; }
        entryCount = ld32(ds);
        dictSize = entryCount << num_shift(data_type);
        if (4 + dictSize + 4 > out_limit) { ret = IXO_BUFFER_OVERFLOW; goto done; }
        st32(out, entryCount);
        memcpy(out + 4, ds + 4, (size_t)dictSize);
        memcpy(ids, ds + 4 + dictSize + 4, (size_t)n * 4);
    } else {
        /* Object2IntLinkedOpenHashMap: ids by first appearance. O(n * entries) scan is fine for tests. */
        const uint8_t* body = data + ((size_t)(n + 1) << 2);
        str_t* e = (str_t*)malloc((size_t)n * sizeof(str_t));
        int m = 0;
        for (int i = 0; i < n; i++) {
            int so = ld32(data + ((size_t)i << 2)), eo = ld32(data + ((size_t)(i + 1) << 2));
            str_t key = {body + so, eo - so};
            int found = -1;
            for (int k = 0; k < m; k++) if (e[k].len == key.len && memcmp(e[k].p, key.p, (size_t)key.len) == 0) { found = k; break; }
            if (found < 0) { found = m; e[m++] = key; }
            ids[i] = found;
        }
        int64_t total = 0;
        for (int i = 0; i < m; i++) total += e[i].len;
        int64_t dsz = ((int64_t)(m + 1) << 2) + total;
        if (4 + dsz + 4 > out_limit) { free(e); ret = IXO_BUFFER_OVERFLOW; goto done; }
        st32(out, m);
        uint8_t* eo_ = out + 4; uint8_t* ebody = eo_ + ((size_t)(m + 1) << 2);
        int lastEnd = 0;
        for (int i = 0; i < m; i++) { st32(eo_ + ((size_t)i << 2), lastEnd); memcpy(ebody + lastEnd, e[i].p, (size_t)e[i].len); lastEnd += e[i].len; }
        st32(eo_ + ((size_t)m << 2), lastEnd);
        free(e);
        entryCount = m; dictSize = (int)dsz;
    }
    {
        int bw = bit_width32((uint32_t)(entryCount - 1));
        int dataSize = ixo_rle_encode_ints(bw, n, ids, out + 4 + dictSize + 4, out_limit - (4 + dictSize + 4));
        if (dataSize < 0) { ret = dataSize; goto done; }
        st32(out + 4 + dictSize, dataSize);
        ret = 4 + dictSize + 4 + dataSize;
    }
done:
    free(ids); free(ds);
    return ret;
}
/* decode* :131-159, 228-256, 322-350, 419-447, 515-541 */
static int simple_dict_decode(int data_type, int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit) {
    if (n == 0) return 0;
    if (in_limit < 4) return IXO_DATA_CORRUPTED;
    int entryCount = ld32(in);
    if (entryCount < 0) return IXO_DATA_CORRUPTED;
    int64_t dictSize;
    if (is_number(data_type)) dictSize = (int64_t)entryCount << num_shift(data_type);
    else {
        if (4 + ((int64_t)(entryCount + 1) << 2) > in_limit) return IXO_DATA_CORRUPTED;
        dictSize = ss_total_len(entryCount, in + 4);
        if (dictSize < 0) return IXO_DATA_CORRUPTED;
    }
    if (4 + dictSize + 4 > in_limit) return IXO_DATA_CORRUPTED;
    int bw = bit_width32((uint32_t)(entryCount - 1));
    /* Rebuild an uncompressed DictStruct in a temporary and gather from it (same loops as the reference). */
    size_t dsbytes = (size_t)(4 + dictSize + 4) + (size_t)n * 4;
    uint8_t* ds = (uint8_t*)malloc(dsbytes + 8);
    memcpy(ds, in, (size_t)(4 + dictSize));
    st32(ds + 4 + dictSize, n << 2);
    int32_t* idtmp = (int32_t*)malloc((size_t)n * 4 + 4);
    int consumed = ixo_rle_decode_ints(bw, n, in + 4 + dictSize + 4, in_limit - (int)(4 + dictSize + 4), idtmp);
    int ret;
    if (consumed < 0) ret = consumed;
    else {
        memcpy(ds + 4 + dictSize + 4, idtmp, (size_t)n * 4);
        int w = ixo_data_from_dict_struct(data_type, n, ds, (int)dsbytes, out, out_limit);
        ret = w < 0 ? w : (int)(4 + dictSize + 4 + consumed);
    }
    free(idtmp); free(ds);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * DictCompressCodec (indexr-vlt-segment/.../codec/dict/DictCompressCodec.java)
 *   | dictStructSize(4) | compressed DictStruct |
 * ---------------------------------------------------------------------------------------------- */
/* Codec.outputBufferSize (indexr-vlt-segment/.../codec/Codec.java:7-27) */
static int output_buffer_size(int type, int n, int dataSize) {
    int64_t size;
    if (type == IXO_SIMPLE_DICT || type == IXO_DICT_COMPRESS || type == IXO_DICT) size = (int64_t)dataSize + ((int64_t)n << 2) + 1024;
    else if (n <= 64) size = (int64_t)dataSize << 1;
    else size = (int64_t)((double)dataSize * 1.5);
    if (size < 1024) size = 1024;
    return (int)size;
}
/* encode :40-69 */
static int dict_compress_encode(int data_type, float ratio, int n, const uint8_t* data, int data_size, uint8_t* out, int out_limit) {
    int dictInput = data_type == IXO_STRING ? ss_total_len(n, data) : data_size;
    int tmpSize = output_buffer_size(IXO_DICT, n, dictInput);
    uint8_t* tmp = (uint8_t*)malloc((size_t)tmpSize + 8);
    int size = ixo_dict_encode(data_type, ratio, n, data, tmp, tmpSize);
    if (size < 0) { free(tmp); return size; }
    if (out_limit < 4) { free(tmp); return IXO_BUFFER_OVERFLOW; }
    st32(out, ixo_dict_struct_size(data_type, tmp));
    int cs = ixo_dict_struct_compress(data_type, n, tmp, out + 4, out_limit - 4);
    free(tmp);
    if (cs < 0) return cs;
    return 4 + cs;
}
/* decode :71-85 */
static int dict_compress_decode(int data_type, int n, const uint8_t* in, int in_limit, uint8_t* out, int out_limit) {
    if (in_limit < 4) return IXO_DATA_CORRUPTED;
    int dictStructSize = ld32(in);
    if (dictStructSize < 8) return IXO_DATA_CORRUPTED;
    uint8_t* tmp = (uint8_t*)malloc((size_t)dictStructSize + 8);
    int consumed = ixo_dict_struct_decompress(data_type, n, in + 4, in_limit - 4, tmp, dictStructSize);
    int ret;
    if (consumed < 0) ret = consumed;
    else {
        int w = ixo_data_from_dict_struct(data_type, n, tmp, dictStructSize, out, out_limit);
        ret = w < 0 ? w : 4 + consumed;
    }
    free(tmp);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * PackCompressor_VLT (indexr-vlt-segment/.../segment/pack/PackCompressor_VLT.java)
 * ---------------------------------------------------------------------------------------------- */
static int delta_encode_any(int data_type, int n, const uint8_t* data, uint8_t* out, int out_limit) {
    switch (data_type) {
        case IXO_INT: case IXO_FLOAT: return ixo_delta_encode_ints(n, (const int32_t*)data, out, out_limit);
        case IXO_LONG: case IXO_DOUBLE: return ixo_delta_encode_longs(n, (const int64_t*)data, out, out_limit);
        case IXO_STRING: return ixo_delta_encode_strings(n, data, out, out_limit);
        default: return IXO_BUG;
    }
}
/* compress :23-128 */
int ixo_compress_pack(int data_type, int is_indexed, int n, const uint8_t* data, int data_size,
                      uint8_t* out, int out_limit, int* encode_type) {
    float ratio = (data_type == IXO_INT && !is_indexed) ? 0.2f : 1.0f;
    int bufSize = output_buffer_size(IXO_DICT_COMPRESS, n, data_size);
    uint8_t* buf = (uint8_t*)malloc((size_t)bufSize + 8);
    int type = IXO_DICT_COMPRESS;
    int cmpSize = dict_compress_encode(data_type, ratio, n, data, data_size, buf, bufSize);
    if (cmpSize <= 0 && !is_indexed) {
        type = IXO_DELTA;
        cmpSize = delta_encode_any(data_type, n, data, buf, bufSize);
    }
    if (cmpSize < 0) { free(buf); return cmpSize; }
    if (cmpSize > out_limit) { free(buf); return IXO_BUFFER_OVERFLOW; }
    memcpy(out, buf, (size_t)cmpSize);
    free(buf);
    if (encode_type) *encode_type = type;
    return cmpSize;
}
int ixo_compress_pack_as(int data_type, int encode_type, int n, const uint8_t* data, int data_size,
                         uint8_t* out, int out_limit) {
    switch (encode_type) {
        case IXO_DICT_COMPRESS: return dict_compress_encode(data_type, 1.0f, n, data, data_size, out, out_limit);
        case IXO_SIMPLE_DICT: return ixo_simple_dict_encode(data_type, n, data, out, out_limit);
        case IXO_DELTA: return delta_encode_any(data_type, n, data, out, out_limit);
        case 0:
            if (data_size > out_limit) return IXO_BUFFER_OVERFLOW;
            memcpy(out, data, (size_t)data_size);
            return data_size;
        default: return IXO_BUG;
    }
}
/* decompress :130-170.  encode_type 0 = pack stored uncompressed (DataPack.java:180-182: data = cmpData). */
int ixo_decompress_pack(int data_type, int encode_type, int n, const uint8_t* in, int pack_size,
                        uint8_t* out, int data_size) {
    switch (encode_type) {
        case 0:
            if (pack_size > data_size) return IXO_BUFFER_OVERFLOW;
            memcpy(out, in, (size_t)pack_size);
            return pack_size;
        case IXO_DICT_COMPRESS: return dict_compress_decode(data_type, n, in, pack_size, out, data_size);
        case IXO_SIMPLE_DICT: return simple_dict_decode(data_type, n, in, pack_size, out, data_size);
        case IXO_DELTA:
            switch (data_type) {
                case IXO_INT: case IXO_FLOAT:
                    if (((int64_t)n << 2) > data_size) return IXO_BUFFER_OVERFLOW;
                    return ixo_delta_decode_ints(n, in, pack_size, (int32_t*)out);
                case IXO_LONG: case IXO_DOUBLE:
                    if (((int64_t)n << 3) > data_size) return IXO_BUFFER_OVERFLOW;
                    return ixo_delta_decode_longs(n, in, pack_size, (int64_t*)out);
                case IXO_STRING: return ixo_delta_decode_strings(n, in, pack_size, out, data_size);
                default: return IXO_BUG;
            }
        default: return IXO_BUG; /* RuntimeException("Illegal encode type") :146 */
    }
}
/* getDictStruct :172-186 */
int ixo_get_dict_struct(int data_type, int n, const uint8_t* in, int pack_size, uint8_t* out, int out_limit) {
    if (pack_size < 4) return IXO_DATA_CORRUPTED;
    int dictStructSize = ld32(in);
    if (dictStructSize > out_limit) return IXO_BUFFER_OVERFLOW;
    int consumed = ixo_dict_struct_decompress(data_type, n, in + 4, pack_size - 4, out, dictStructSize);
    return consumed < 0 ? consumed : 4 + consumed;
}

/* ------------------------------------------------------------------------------------------------
 * Batch driver for the CPU baseline: one pack per task from a shared counter (the reference's natural
 * parallelism is one reader thread per engine task; packs are independent).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { ixo_pack_desc* d; int count; const uint8_t* in; uint8_t* out; int next; pthread_mutex_t mu; } batch_ctx;
static void* batch_worker(void* arg) {
    batch_ctx* c = (batch_ctx*)arg;
    for (;;) {
        int i = __atomic_fetch_add(&c->next, 1, __ATOMIC_RELAXED);
        if (i >= c->count) break;
        ixo_pack_desc* d = &c->d[i];
        d->status = ixo_decompress_pack(d->data_type, d->encode_type, d->obj_count, c->in + d->in_off, d->pack_size,
                                        c->out + d->out_off, d->data_size);
    }
    return NULL;
}
int ixo_decode_batch(ixo_pack_desc* descs, int count, const uint8_t* in_arena, uint8_t* out_arena, int threads) {
    batch_ctx c; c.d = descs; c.count = count; c.in = in_arena; c.out = out_arena; c.next = 0;
    if (threads <= 1) { batch_worker(&c); }
    else {
        if (threads > 256) threads = 256;
        pthread_t th[256];
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, batch_worker, &c);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    int bad = 0;
    for (int i = 0; i < count; i++) if (descs[i].status != descs[i].pack_size) bad++;
    return bad;
}

/* ---- engine hand-back (DefaultPackReader.java:164-241) ---- */
int ixo_handback(int sql_type, int n, const uint8_t* decoded, int data_size, int row_offset, int row_count,
                 uint8_t* out, int out_limit) {
    if (n <= 0 || row_offset < 0 || row_count < 0 || row_offset + (long long)row_count > n) return -1000;
    switch (sql_type) {
        case 1: case 3: {  /* INT, FLOAT: readByCopy, HackColumnVector.copyInts/copyFloats */
            if ((long long)n * 4 > data_size) return -1000;
            if ((long long)row_count * 4 > out_limit) return IXO_BUFFER_OVERFLOW;
            memcpy(out, decoded + (size_t)row_offset * 4, (size_t)row_count * 4);
            return row_count * 4;
        }
        case 2: case 4: {  /* BIGINT, DOUBLE */
            if ((long long)n * 8 > data_size) return -1000;
            if ((long long)row_count * 8 > out_limit) return IXO_BUFFER_OVERFLOW;
            memcpy(out, decoded + (size_t)row_offset * 8, (size_t)row_count * 8);
            return row_count * 8;
        }
        case 6: {  /* DATE: stored as long milliseconds, Spark wants int days: (int) (value / MILLIS_PER_DAY)  :214-219 */
            if ((long long)n * 8 > data_size) return -1000;
            if ((long long)row_count * 4 > out_limit) return IXO_BUFFER_OVERFLOW;
            for (int i = 0; i < row_count; i++) {
                int64_t v;
                memcpy(&v, decoded + (size_t)(row_offset + i) * 8, 8);
                const int64_t q = v / 86400000LL;           /* Java long division truncates toward zero, like C */
                const int32_t d = (int32_t)(uint32_t)(uint64_t)q;  /* (int) cast keeps the low 32 bits */
                memcpy(out + (size_t)i * 4, &d, 4);
            }
            return row_count * 4;
        }
        case 8: {  /* DATETIME: milliseconds -> microseconds, value * 1000 with Java's wrapping multiply  :220-225 */
            if ((long long)n * 8 > data_size) return -1000;
            if ((long long)row_count * 8 > out_limit) return IXO_BUFFER_OVERFLOW;
            for (int i = 0; i < row_count; i++) {
                uint64_t v;
                memcpy(&v, decoded + (size_t)(row_offset + i) * 8, 8);
                v *= 1000u;
                memcpy(out + (size_t)i * 8, &v, 8);
            }
            return row_count * 8;
        }
        case 5: {  /* VARCHAR: columnVector.putByteArray(id - offset, bytes, off, len) per row  :226-232 */
            if (4LL * (n + 1) > data_size) return -1000;
            int32_t a, b;
            memcpy(&a, decoded + 4 * (size_t)row_offset, 4);
            memcpy(&b, decoded + 4 * (size_t)(row_offset + row_count), 4);
            if (a < 0 || b < a || 4LL * (n + 1) + b > data_size) return -1000;
            if (8LL * row_count + (b - a) > out_limit) return IXO_BUFFER_OVERFLOW;
            int32_t appended = 0;  /* arrayData().appendBytes returns the position the bytes went to */
            for (int i = 0; i < row_count; i++) {
                int32_t s, e;
                memcpy(&s, decoded + 4 * (size_t)(row_offset + i), 4);
                memcpy(&e, decoded + 4 * (size_t)(row_offset + i + 1), 4);
                const int32_t len = e - s;
                memcpy(out + 4 * (size_t)i, &appended, 4);
                memcpy(out + 4 * (size_t)(row_count + i), &len, 4);
                memcpy(out + 8 * (size_t)row_count + appended, decoded + 4 * (size_t)(n + 1) + s, (size_t)len);
                appended += len;
            }
            return 8 * row_count + appended;
        }
        default:
            return -102;  /* IllegalStateException("Unsupported date type ...")  :233-234 */
    }
}

/* ---- predicate evaluation on dictionary ids (ExtIndex_DictBits.java) ---- */
/* searchEntry (:33-58) with BinarySearch.binarySearch{Ints,Longs,Floats,Doubles,Strings} (BinarySearch.java:27-47,
 * 100-120,162-181,229-248,295-316): plain < and > on the entry type; -(insertion point) - 1 when absent. */
static int search_entry(int data_type, const uint8_t* entries, int ec, int64_t num, const uint8_t* key, int key_len) {
    int from = 0, to = ec - 1;
    const uint8_t* body = entries + 4 * ((size_t)ec + 1);
    double dkey; memcpy(&dkey, &num, 8);
    const float fkey = (float)dkey;      /* (float) Double.longBitsToDouble(numValue) */
    const int32_t ikey = (int32_t)(uint32_t)(uint64_t)num; /* (int) numValue */
    while (from <= to) {
        const int mid = (int)(((unsigned)from + (unsigned)to) >> 1);
        int lt, gt;
        switch (data_type) {
            case IXO_INT: { int32_t v = ld32(entries + ((size_t)mid << 2)); lt = v < ikey; gt = v > ikey; break; }
            case IXO_LONG: { int64_t v = ld64(entries + ((size_t)mid << 3)); lt = v < num; gt = v > num; break; }
            case IXO_FLOAT: { float v; memcpy(&v, entries + ((size_t)mid << 2), 4); lt = v < fkey; gt = v > fkey; break; }
            case IXO_DOUBLE: { double v; memcpy(&v, entries + ((size_t)mid << 3), 8); lt = v < dkey; gt = v > dkey; break; }
            default: {
                const int32_t a = ld32(entries + 4 * (size_t)mid), b = ld32(entries + 4 * (size_t)(mid + 1));
                const int c = compare_bytes(body + a, b - a, key, key_len);
                lt = c < 0; gt = c > 0;
            }
        }
        if (lt) from = mid + 1; else if (gt) to = mid - 1; else return mid;
    }
    return -(from + 1);
}

/* SQLLike.match (indexr-common/src/main/java/io/indexr/util/SQLLike.java:12-94), the recursion as it stands: % any
 * sequence, _ one byte, \\ escapes the next pattern byte. */
static int like_match(const uint8_t* str, int sLen, const uint8_t* pat, int pLen, int sNdx, int pNdx) {
    if (pLen == 1 && pat[0] == '%') return 1;
    int nextIsNotWildcard = 0;
    for (;;) {
        if (sNdx >= sLen) {
            while (pNdx < pLen && pat[pNdx] == '%') pNdx++;
            return pNdx >= pLen;
        }
        if (pNdx >= pLen) return 0;
        const uint8_t p = pat[pNdx];
        if (!nextIsNotWildcard) {
            if (p == '\\') { pNdx++; nextIsNotWildcard = 1; continue; }
            if (p == '_') { sNdx++; pNdx++; continue; }
            if (p == '%') {
                uint8_t pNext = 0;
                if (pNdx + 1 < pLen) pNext = pat[pNdx + 1];
                if (pNext == '%') { pNdx++; continue; }
                pNdx++;
                for (int i = sLen; i >= sNdx; i--)
                    if (like_match(str, sLen, pat, pLen, i, pNdx)) return 1;
                return 0;
            }
        } else {
            nextIsNotWildcard = 0;
        }
        if (p != str[sNdx]) return 0;
        sNdx++;
        pNdx++;
    }
}
int ixo_like(const uint8_t* str, int str_len, const uint8_t* pattern, int pattern_len) {
    return like_match(str, str_len, pattern, pattern_len, 0, 0);
}

int ixo_filter(int data_type, int encode_type, int n, const uint8_t* pack, int pack_size, int op, int n_values,
               const int64_t* num_values, const uint8_t* str_values, uint64_t* bitmap) {
    if (encode_type != IXO_DICT_COMPRESS) return IXO_DATA_IMPROPER;  /* only dict-encoded packs get ExtIndex_DictBits */
    if (pack_size < 4 || n_values < 1) return -1000;
    const int ds_size = ld32(pack);
    if (ds_size < 8) return IXO_DATA_CORRUPTED;
    uint8_t* ds = (uint8_t*)calloc((size_t)ds_size + 16, 1);
    const int consumed = ixo_get_dict_struct(data_type, n, pack, pack_size, ds, ds_size);
    if (consumed < 0) { free(ds); return consumed; }
    const int ec = ld32(ds);
    const uint8_t* entries = ds + 4;
    const int dict_size = ixo_dict_struct_size(data_type, ds) - 4 - 4 - 4 * n;
    const uint8_t* ids = ds + 4 + dict_size + 4;
    memset(bitmap, 0, 8192);
    int result = 2;
#define KEY(i, kp, kl) do { if (data_type == IXO_STRING) { int32_t a_ = ld32(str_values + 4 * (size_t)(i)), b_ = ld32(str_values + 4 * (size_t)((i) + 1)); \
        kp = str_values + 4 * ((size_t)n_values + 1) + a_; kl = b_ - a_; } else { kp = NULL; kl = 0; } } while (0)
#define NUM(i) (data_type == IXO_STRING ? 0 : num_values[i])
    const uint8_t* kp; int kl;
    if (op == 1) {  /* equal */
        KEY(0, kp, kl);
        const int id = search_entry(data_type, entries, ec, NUM(0), kp, kl);
        if (id < 0) result = 0;
        else if (ec == 1) result = 1;
        else for (int i = 0; i < n; i++) if (ld32(ids + ((size_t)i << 2)) == id) bitmap[i >> 6] |= 1ull << (i & 63);
    } else if (op == 2) {  /* in */
        int* found = (int*)malloc(sizeof(int) * (size_t)n_values);
        int m = 0;
        for (int v = 0; v < n_values; v++) {
            KEY(v, kp, kl);
            const int id = search_entry(data_type, entries, ec, NUM(v), kp, kl);
            if (id >= 0) found[m++] = id;
        }
        if (m == 0) result = 0;
        else for (int i = 0; i < n; i++) {
            const int vid = ld32(ids + ((size_t)i << 2));
            for (int e = 0; e < m; e++) if (vid == found[e]) bitmap[i >> 6] |= 1ull << (i & 63);
        }
        free(found);
    } else if (op == 3 || op == 4) {  /* greater, acceptEqual = (op == 4) */
        KEY(0, kp, kl);
        const int id = search_entry(data_type, entries, ec, NUM(0), kp, kl);
        const int start = id >= 0 ? (op == 4 ? id : id + 1) : -id - 1;
        if (start >= ec) result = 0;
        else if (start == 0) result = 1;
        else for (int i = 0; i < n; i++) if (ld32(ids + ((size_t)i << 2)) >= start) bitmap[i >> 6] |= 1ull << (i & 63);
    } else if (op == 5) {  /* between */
        if (n_values < 2) { free(ds); return -1000; }
        KEY(0, kp, kl);
        const int id1 = search_entry(data_type, entries, ec, NUM(0), kp, kl);
        KEY(1, kp, kl);
        const int id2 = search_entry(data_type, entries, ec, NUM(1), kp, kl);
        const int from = id1 >= 0 ? id1 : -id1 - 1, to = id2 >= 0 ? id2 + 1 : -id2 - 1;
        if (!(from >= 0 && from < ec) || !(to >= 0 && to <= ec)) result = IXO_BUG;  /* Preconditions.checkState :165-166 */
        else if (to - from <= 0) result = 0;
        else if (to - from >= ec) result = 1;
        else for (int i = 0; i < n; i++) { const int id = ld32(ids + ((size_t)i << 2)); if (id >= from && id < to) bitmap[i >> 6] |= 1ull << (i & 63); }
    } else if (op == 6) {  /* like (:182-200): SQLLike.match on every row's string; fixBitmapInPack: empty -> NONE */
        if (data_type != IXO_STRING) { free(ds); return -102; }  /* "column type ... is illegal in LIKE" */
        KEY(0, kp, kl);
        const uint8_t* body = entries + 4 * ((size_t)ec + 1);
        int any = 0;
        for (int i = 0; i < n; i++) {
            const int id = ld32(ids + ((size_t)i << 2));
            const int32_t a = ld32(entries + 4 * (size_t)id), b = ld32(entries + 4 * (size_t)(id + 1));
            if (like_match(body + a, b - a, kp, kl, 0, 0)) { bitmap[i >> 6] |= 1ull << (i & 63); any = 1; }
        }
        if (!any) result = 0;
    } else {
        result = -1000;
    }
#undef KEY
#undef NUM
    free(ds);
    return result;
}

static double bits_to_double(int64_t bits) { double d; memcpy(&d, &bits, 8); return d; }  /* Double.longBitsToDouble */

/* ExtIndex_SimpleBits (indexr-segment/src/main/java/io/indexr/segment/index/ExtIndex_SimpleBits.java): equal :25-89, in
 * :91-166, greater :168-234, between :236-311, like :313-331, row after row over the decoded pack.  Keys as in
 * ixo_filter.  Returns 0 (NONE: fixBitmapInPack on an empty bitmap) or 2, -102 for LIKE on numbers. */
int ixo_scan_filter(int data_type, int n, const uint8_t* data, int data_size, int op, int n_values,
                    const int64_t* num_values, const uint8_t* str_values, uint64_t* bitmap) {
    memset(bitmap, 0, 8192);
    if (op < 1 || op > 6 || n_values < 1 || (op == 5 && n_values < 2)) return -1000;
    if (op == 6 && data_type != IXO_STRING) return -102;
    if (data_type == IXO_STRING ? 4 * ((int64_t)n + 1) > data_size : (int64_t)n * ((data_type == IXO_INT || data_type == IXO_FLOAT) ? 4 : 8) > data_size) return -1000;
    int any = 0;
#define SET(i) do { bitmap[(i) >> 6] |= 1ull << ((i) & 63); any = 1; } while (0)
#define SCAN(T, LOAD, KEYOF)                                                                                             \
    do {                                                                                                                 \
        const T k0 = KEYOF(num_values[0]);                                                                               \
        const T k1 = KEYOF(num_values[n_values > 1]);                                                                    \
        for (int i = 0; i < n; i++) {                                                                                    \
            T v; memcpy(&v, data + (size_t)i * sizeof(T), sizeof(T));                                                    \
            int hit = 0;                                                                                                 \
            if (op == 1) hit = v == k0;                                                                                  \
            else if (op == 2) { for (int e = 0; e < n_values && !hit; e++) hit = v == KEYOF(num_values[e]); }            \
            else if (op == 3) hit = v > k0;                                                                              \
            else if (op == 4) hit = v > k0 || v == k0;                                                                   \
            else hit = v >= k0 && v <= k1;                                                                               \
            if (hit) SET(i);                                                                                             \
        }                                                                                                                \
    } while (0)
#define K_INT(x) ((int32_t)(uint32_t)(uint64_t)(x))
#define K_LONG(x) ((int64_t)(x))
#define K_FLOAT(x) ((float)bits_to_double(x))
#define K_DOUBLE(x) (bits_to_double(x))
    switch (data_type) {
        case IXO_INT: SCAN(int32_t, 0, K_INT); break;
        case IXO_LONG: SCAN(int64_t, 0, K_LONG); break;
        case IXO_FLOAT: SCAN(float, 0, K_FLOAT); break;
        case IXO_DOUBLE: SCAN(double, 0, K_DOUBLE); break;
        case IXO_STRING: {
            const uint8_t* body = data + 4 * ((size_t)n + 1);
            const uint8_t* kbody = str_values + 4 * ((size_t)n_values + 1);
            for (int i = 0; i < n; i++) {
                const int32_t a = ld32(data + 4 * (size_t)i), b = ld32(data + 4 * (size_t)(i + 1));
                if (a < 0 || b < a || 4 * ((int64_t)n + 1) + b > data_size) return IXO_DATA_CORRUPTED;
                const uint8_t* sp = body + a;
                const int sl = b - a;
                int hit = 0;
                if (op == 1 || op == 2) {
                    for (int e = 0; e < n_values && !hit; e++) {
                        const int32_t ka = ld32(str_values + 4 * (size_t)e), kb = ld32(str_values + 4 * (size_t)(e + 1));
                        hit = kb - ka == sl && memcmp(sp, kbody + ka, (size_t)sl) == 0;
                    }
                } else if (op == 6) {
                    const int32_t ka = ld32(str_values), kb = ld32(str_values + 4);
                    hit = like_match(sp, sl, kbody + ka, kb - ka, 0, 0);
                } else {
                    const int32_t ka = ld32(str_values), kb = ld32(str_values + 4);
                    const int c1 = compare_bytes(sp, sl, kbody + ka, kb - ka);
                    if (op == 3) hit = c1 > 0;
                    else if (op == 4) hit = c1 >= 0;
                    else {
                        const int32_t kc = ld32(str_values + 8);
                        hit = c1 >= 0 && compare_bytes(sp, sl, kbody + kb, kc - kb) <= 0;
                    }
                }
                if (hit) SET(i);
            }
            break;
        }
        default: return -102;
    }
#undef SCAN
#undef SET
    return any ? 2 : 0;
}

/* DataPack.foreach(BitMap position, XxxSetter) (indexr-segment/src/main/java/io/indexr/segment/pack/DataPack.java: int
 * :273-306, long :324-357, float :375-408, double :426-459, BytePiece :708-762): position == BitMap.ALL (here bitmap ==
 * NULL) visits every row; otherwise DirectBitMapIterator.nextSetBit hands out the set bits in rising order and the loop
 * stops at the first index >= objCount.  Every setter.set(index, value) call is recorded: row_ids[k] = index; numbers:
 * the value's bytes at values + k * width; strings: str_off[k] / str_len[k] and the bytes appended to values.  Returns
 * the number of calls (the reference's return value). */
int ixo_select(int data_type, int n, const uint8_t* data, int data_size, const uint64_t* bitmap, int32_t* row_ids,
               uint8_t* values, int32_t* str_off, int32_t* str_len) {
    if (n < 0 || data_type < IXO_INT || data_type > IXO_STRING) return -1000;
    const int w = (data_type == IXO_INT || data_type == IXO_FLOAT) ? 4 : 8;
    if (data_type == IXO_STRING ? 4 * ((int64_t)n + 1) > data_size : (int64_t)n * w > data_size) return -1000;
    int count = 0;
    int32_t bytes = 0;
    int word = 0;
    uint64_t cur = bitmap ? bitmap[0] : 0;
    int next_all = 0;
    for (;;) {
        int index;
        if (!bitmap) {
            if (next_all >= n) break;
            index = next_all++;
        } else {
            while (cur == 0 && word < 1023) cur = bitmap[++word];
            if (cur == 0) break;  /* NO_MORE */
            index = word * 64 + __builtin_ctzll(cur);
            cur &= cur - 1;
            if (index >= n) break;
        }
        row_ids[count] = index;
        if (data_type == IXO_STRING) {
            const int32_t a = ld32(data + 4 * (size_t)index), b = ld32(data + 4 * (size_t)(index + 1));
            if (a < 0 || b < a || 4 * ((int64_t)n + 1) + b > data_size) return IXO_DATA_CORRUPTED;
            /* the recorded bytes are bounded by the pack's body; overlapping rows (offsets stepping backwards between
             * two picked rows) that exceed it are reported rather than recorded */
            if (4 * ((int64_t)n + 1) + bytes + (b - a) > data_size) return IXO_DATA_CORRUPTED;
            memcpy(values + bytes, data + 4 * ((size_t)n + 1) + a, (size_t)(b - a));
            str_off[count] = bytes;
            str_len[count] = b - a;
            bytes += b - a;
        } else {
            memcpy(values + (size_t)count * w, data + (size_t)index * w, (size_t)w);
        }
        count++;
    }
    return count;
}

/* DataPackCreator_VLT.from (indexr-vlt-segment/.../segment/pack/DataPackCreator_VLT.java:21-163): dpn.minValue / maxValue /
 * sum of a numeric pack.  java.lang.Math.min / max on float and double: a NaN first argument is returned, -0.0 is below
 * 0.0, otherwise <= / >=. */
static float jmin_f(float a, float b) { if (a != a) return a; if (a == 0.0f && b == 0.0f && signbit(b)) return b; return a <= b ? a : b; }
static float jmax_f(float a, float b) { if (a != a) return a; if (a == 0.0f && b == 0.0f && signbit(a)) return b; return a >= b ? a : b; }
static double jmin_d(double a, double b) { if (a != a) return a; if (a == 0.0 && b == 0.0 && signbit(b)) return b; return a <= b ? a : b; }
static double jmax_d(double a, double b) { if (a != a) return a; if (a == 0.0 && b == 0.0 && signbit(a)) return b; return a >= b ? a : b; }
static int64_t double_bits(double d) { int64_t v; memcpy(&v, &d, 8); return v; }
int ixo_pack_stats(int data_type, int n, const uint8_t* data, int64_t* min_value, int64_t* max_value, int64_t* sum) {
    if (n < 1) return -1000;
    switch (data_type) {
        case IXO_INT: {  /* :21-46 */
            int32_t mn, mx, v; uint64_t s = 0;
            memcpy(&mn, data, 4); mx = mn;
            for (int i = 0; i < n; i++) { memcpy(&v, data + 4 * (size_t)i, 4); if (v < mn) mn = v; if (v > mx) mx = v; s += (uint64_t)(int64_t)v; }
            *min_value = mn; *max_value = mx; *sum = (int64_t)s;
            return 0;
        }
        case IXO_LONG: {  /* :56-81 */
            int64_t mn, mx, v; uint64_t s = 0;
            memcpy(&mn, data, 8); mx = mn;
            for (int i = 0; i < n; i++) { memcpy(&v, data + 8 * (size_t)i, 8); if (v < mn) mn = v; if (v > mx) mx = v; s += (uint64_t)v; }
            *min_value = mn; *max_value = mx; *sum = (int64_t)s;
            return 0;
        }
        case IXO_FLOAT: {  /* :91-118: float sum, NumType.floatToLong = raw bits of the widened value */
            float mn, mx, v; volatile float s = 0;
            memcpy(&mn, data, 4); mx = mn;
            for (int i = 0; i < n; i++) { memcpy(&v, data + 4 * (size_t)i, 4); mn = jmin_f(v, mn); mx = jmax_f(v, mx); s = s + v; }
            *min_value = double_bits((double)mn); *max_value = double_bits((double)mx); *sum = double_bits((double)s);
            return 0;
        }
        case IXO_DOUBLE: {  /* :128-155 */
            double mn, mx, v; volatile double s = 0;
            memcpy(&mn, data, 8); mx = mn;
            for (int i = 0; i < n; i++) { memcpy(&v, data + 8 * (size_t)i, 8); mn = jmin_d(v, mn); mx = jmax_d(v, mx); s = s + v; }
            *min_value = double_bits(mn); *max_value = double_bits(mx); *sum = double_bits(s);
            return 0;
        }
        default: return -102;
    }
}
