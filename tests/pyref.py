"""Independent spec-level codec in pure Python (big-int bit strings, no tables, no byte-wise shifts).

Test infrastructure.  It is written from the wire format rather than from the reference's control flow, so that it
can cross-check the C oracle (oracle/ixr_oracle.c), which follows the Java structure.  Small inputs only.

Wire format sources (relative to /root/reference/indexr-vlt-segment/src/main/java/io/indexr/vlt/codec):
  bit packing   pack/PackingUtil.java:14-21, pack/IntPacker.java:35-70 (+ generator indexr-codegen/.../PackingClassGen.java)
  varints       VarInteger.java:49-127
  DELTA         delta/DeltaCodec.java:40-141, 144-245, 268-388
  RLE hybrid    rle/RLEPackingHybridCodec.java:38-128
  DictStruct    dict/DictStructUtil.java:25-142, dict/DictCompressCodec.java:40-85, dict/SimpleDictCodec.java
"""
import struct

INT, LONG, FLOAT, DOUBLE, STRING = 1, 2, 3, 4, 5
PLAIN, DELTA, RLE, SIMPLE_DICT, DICT, DICT_COMPRESS = 1, 2, 3, 4, 5, 6


def _width(t):
    return 32 if t in (INT, FLOAT) else 64


# ------------------------------------------------------------------ bit packing
def pack_bits(vals, bw):
    """n values -> ceil(n/8)*bw bytes; value i sits at bit i*bw of the little-endian bit string."""
    if not vals or bw == 0:
        return b""
    mask = (1 << bw) - 1
    acc = 0
    for i, v in enumerate(vals):
        acc |= (v & mask) << (i * bw)
    return acc.to_bytes(((len(vals) + 7) // 8) * bw, "little")


def unpack_bits(buf, off, n, bw):
    nbytes = ((n + 7) // 8) * bw if (n and bw) else 0
    acc = int.from_bytes(buf[off:off + nbytes], "little")
    mask = (1 << bw) - 1
    return [(acc >> (i * bw)) & mask for i in range(n)], nbytes


# ------------------------------------------------------------------ varints
def put_uvarint(out, v):
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def get_uvarint(buf, pos):
    v = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def zigzag(v, bits):
    m = (1 << bits) - 1
    return ((v << 1) ^ (v >> (bits - 1))) & m


def unzigzag(u, bits):
    v = (u >> 1) ^ -(u & 1)
    return _signed(v, bits)


def _signed(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


# ------------------------------------------------------------------ DELTA numbers
def delta_encode(vals, bits):
    """vals: python ints (signed, `bits` wide)."""
    if not vals:
        return b""
    m = (1 << bits) - 1
    body = bytearray()
    put_uvarint(body, zigzag(vals[0], bits))
    prev = vals[0]
    for b0 in range(0, len(vals), 128):
        blk = vals[b0:b0 + 128]
        deltas = []
        for v in blk:
            deltas.append(_signed(v - prev, bits))
            prev = v
        mn = min(deltas)
        put_uvarint(body, zigzag(mn, bits))
        rel = [(d - mn) & m for d in deltas]
        for m0 in range(0, len(rel), 32):
            mb = rel[m0:m0 + 32]
            orv = 0
            for x in mb:
                orv |= x
            bw = orv.bit_length()
            body.append(bw)
            body += pack_bits(mb, bw)
    return struct.pack("<i", len(body)) + bytes(body)


def delta_decode(buf, off, n, bits):
    """-> (values, consumed)"""
    if n == 0:
        return [], 0
    (enc_len,) = struct.unpack_from("<i", buf, off)
    pos = off + 4
    first, pos = get_uvarint(buf, pos)
    prev = unzigzag(first, bits)
    out = []
    for b0 in range(0, n, 128):
        cnt = min(128, n - b0)
        mn, pos = get_uvarint(buf, pos)
        mn = unzigzag(mn, bits)
        for m0 in range(0, cnt, 32):
            c = min(32, cnt - m0)
            bw = buf[pos]
            pos += 1
            vals, used = unpack_bits(buf, pos, c, bw)
            pos += used
            for x in vals:
                prev = _signed(prev + mn + x, bits)
                out.append(prev)
    assert pos - off - 4 == enc_len, (pos - off - 4, enc_len)
    return out, pos - off


# ------------------------------------------------------------------ strings
def strings_struct(strs):
    offs = [0]
    for s in strs:
        offs.append(offs[-1] + len(s))
    return struct.pack("<%di" % len(offs), *offs) + b"".join(strs)


def parse_strings_struct(buf, off, n):
    offs = struct.unpack_from("<%di" % (n + 1), buf, off)
    body = off + 4 * (n + 1)
    return [bytes(buf[body + offs[i]:body + offs[i + 1]]) for i in range(n)], 4 * (n + 1) + offs[n]


def delta_encode_strings(strs):
    if not strs:
        return b""
    prev = b""
    pre, suf, tail = [], [], bytearray()
    for s in strs:
        p = 0
        while p < min(len(prev), len(s)) and prev[p] == s[p]:
            p += 1
        pre.append(p)
        suf.append(len(s) - p)
        tail += s[p:]
        prev = s
    body = delta_encode(pre, 32) + delta_encode(suf, 32) + bytes(tail)
    return struct.pack("<i", len(body)) + body


def delta_decode_strings(buf, off, n):
    if n == 0:
        return [], 0
    (enc_len,) = struct.unpack_from("<i", buf, off)
    pos = off + 4
    pre, c1 = delta_decode(buf, pos, n, 32)
    pos += c1
    suf, c2 = delta_decode(buf, pos, n, 32)
    pos += c2
    out = []
    prev = b""
    for p, s in zip(pre, suf):
        cur = prev[:p] + bytes(buf[pos:pos + s])
        pos += s
        out.append(cur)
        prev = cur
    assert pos - off - 4 == enc_len
    return out, pos - off


# ------------------------------------------------------------------ RLE / bit-packing hybrid
def rle_encode(ids, bw):
    if not ids:
        return b""
    bytew = (bw + 7) // 8
    body = bytearray()
    n = len(ids)
    # maximal runs
    runs = []
    i = 0
    while i < n:
        j = i
        while j < n and ids[j] == ids[i]:
            j += 1
        runs.append((i, j))
        i = j
    lit_start = 0

    def flush_lit(a, b):
        if a == b:
            return
        put_uvarint(body, ((b - a) << 1) | 1)
        body.extend(pack_bits(ids[a:b], bw))

    for k, (a, b) in enumerate(runs):
        last = k == len(runs) - 1
        if b - a >= 8 and not last:  # a run is only emitted when a different value follows it
            flush_lit(lit_start, a)
            put_uvarint(body, (b - a) << 1)
            body.extend((ids[a] & 0xFFFFFFFF).to_bytes(4, "little")[:bytew])
            lit_start = b
    flush_lit(lit_start, n)
    return struct.pack("<i", len(body)) + bytes(body)


def rle_decode(buf, off, n, bw):
    if n == 0:
        return [], 0
    (enc_len,) = struct.unpack_from("<i", buf, off)
    pos = off + 4
    end = pos + enc_len
    bytew = (bw + 7) // 8
    out = []
    while pos < end:
        h, pos = get_uvarint(buf, pos)
        if h & 1:
            vals, used = unpack_bits(buf, pos, h >> 1, bw)
            pos += used
            out += vals
        else:
            v = int.from_bytes(buf[pos:pos + bytew], "little")
            pos += bytew
            out += [v] * (h >> 1)
    assert len(out) == n and pos == end
    return out, pos - off


# ------------------------------------------------------------------ dictionaries
def _num_fmt(t):
    return {INT: "i", LONG: "q", FLOAT: "f", DOUBLE: "d"}[t]


def _bits_fmt(t):
    return {INT: "i", LONG: "q", FLOAT: "i", DOUBLE: "q"}[t]


def dict_struct_decompress(buf, off, n, t):
    """compressed DictStruct (after the 4-byte size) -> (entries, ids, consumed).
    Numeric entries are returned as raw bit patterns (signed ints)."""
    entry_count, = struct.unpack_from("<i", buf, off)
    compressed = buf[off + 4] != 0
    pos = off + 5
    if t == STRING:
        if compressed:
            entries, c = delta_decode_strings(buf, pos, entry_count)
        else:
            entries, c = parse_strings_struct(buf, pos, entry_count)
    else:
        bits = _width(t)
        if compressed:
            entries, c = delta_decode(buf, pos, entry_count, bits)
        else:
            entries = list(struct.unpack_from("<%d%s" % (entry_count, _bits_fmt(t)), buf, pos))
            c = entry_count * bits // 8
    pos += c
    bw = max(entry_count - 1, 0).bit_length() if entry_count >= 1 else 32
    ids, c = rle_decode(buf, pos, n, bw)
    pos += c
    return entries, ids, pos - off


def dict_struct_bytes(entries, ids, t):
    if t == STRING:
        e = strings_struct(entries)
    else:
        e = struct.pack("<%d%s" % (len(entries), _bits_fmt(t)), *entries)
    return struct.pack("<i", len(entries)) + e + struct.pack("<i", 4 * len(ids)) + struct.pack("<%di" % len(ids), *ids)


def rows_bytes(rows, t):
    if t == STRING:
        return strings_struct(rows)
    return struct.pack("<%d%s" % (len(rows), _bits_fmt(t)), *rows)


def decode_pack(buf, n, t, encode_type):
    """whole pack -> (decoded bytes, consumed)"""
    buf = bytes(buf)
    if encode_type == 0:
        return buf, len(buf)
    if encode_type == DELTA:
        if t == STRING:
            rows, c = delta_decode_strings(buf, 0, n)
        else:
            rows, c = delta_decode(buf, 0, n, _width(t))
        return rows_bytes(rows, t), c
    if encode_type == DICT_COMPRESS:
        entries, ids, c = dict_struct_decompress(buf, 4, n, t)
        assert struct.unpack_from("<i", buf, 0)[0] == len(dict_struct_bytes(entries, ids, t))
        return rows_bytes([entries[i] for i in ids], t), 4 + c
    if encode_type == SIMPLE_DICT:
        entry_count, = struct.unpack_from("<i", buf, 0)
        if t == STRING:
            entries, c = parse_strings_struct(buf, 4, entry_count)
        else:
            entries = list(struct.unpack_from("<%d%s" % (entry_count, _bits_fmt(t)), buf, 4))
            c = entry_count * _width(t) // 8
        pos = 4 + c + 4
        bw = max(entry_count - 1, 0).bit_length()
        ids, c2 = rle_decode(buf, pos, n, bw)
        assert struct.unpack_from("<i", buf, 4 + c)[0] == c2
        return rows_bytes([entries[i] for i in ids], t), pos + c2
    raise ValueError("illegal encode type %r" % encode_type)


def encode_dict_compress(rows, t):
    """rows: raw bit patterns (numbers; no NaN / -0.0) or bytes objects. Sorted dictionary, ratio 1."""
    if t == STRING:
        entries = sorted(set(rows))
    elif t in (FLOAT, DOUBLE):
        fmt, ifmt = _num_fmt(t), _bits_fmt(t)
        as_f = lambda b: struct.unpack("<" + fmt, struct.pack("<" + ifmt, b))[0]
        entries = sorted(set(rows), key=as_f)
    else:
        entries = sorted(set(rows))
    index = {e: i for i, e in enumerate(entries)}
    ids = [index[r] for r in rows]
    ds = dict_struct_bytes(entries, ids, t)
    dict_size = len(ds) - 8 - 4 * len(ids)
    compressed = len(entries) >= 64 and dict_size >= 256
    out = struct.pack("<iiB", len(ds), len(entries), 1 if compressed else 0)
    if compressed:
        out += delta_encode_strings(entries) if t == STRING else delta_encode(entries, _width(t))
    else:
        out += ds[4:4 + dict_size]
    out += rle_encode(ids, max(len(entries) - 1, 0).bit_length())
    return out
