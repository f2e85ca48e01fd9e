"""Where the device decoder answers differently from the reference on packs no writer produces.  One test per entry of
the table of divergences: each builds the pack by hand, states what the reference's code does with it (file:line of the
Java that decides) and what the restated oracle does, and asserts the status code the device reports.  Everywhere the
device is the stricter side: it reports an error where the reference would read or write outside a buffer, or would
silently accept a header field that contradicts the bytes."""
import struct

import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from gpu_util import Case, run_batch
from indexr_amd import lib as L

pytestmark = pytest.mark.gpu


def _oracle(pack, data_type, encode_type, n, data_size):
    dec, consumed = O.decompress_pack(np.frombuffer(pack, np.uint8), data_type, encode_type, n, data_size)
    return dec, consumed


def _gpu(pack, data_type, encode_type, n, data_size):
    outs, status, plan = run_batch([Case(np.frombuffer(pack, np.uint8), data_type, encode_type, n, data_size, None, "divergence")])
    plan.destroy()
    single, consumed = L.decode_pack(np.frombuffer(pack, np.uint8), data_type, encode_type, n, data_size)
    assert consumed == int(status[0]), "batch and single-pack entry points disagree: %d vs %d" % (status[0], consumed)
    return int(status[0])


def test_entry_stream_length_word_larger_than_the_stream():
    """DeltaCodec.decodeInts bounds its ByteBuffer by the length word but returns the position it reached
    (DeltaCodec.java:102-110), and DictStructUtil.decompress finds the id stream behind that position
    (DictStructUtil.java:134-141): a length word that is too large changes nothing there.  The device finds the id stream
    from the length word and requires the walk to end where the word says: DATA_CORRUPTED."""
    entries = [i * 7 for i in range(100)]
    ids = [i % 100 for i in range(500)]
    e = R.delta_encode(entries, 32)
    (l0,) = struct.unpack("<i", e[:4])
    good = struct.pack("<iiB", 4 + 400 + 4 + 2000, 100, 1) + e + R.rle_encode(ids, 7)
    bad = struct.pack("<iiB", 4 + 400 + 4 + 2000, 100, 1) + struct.pack("<i", l0 + 1) + e[4:] + R.rle_encode(ids, 7)
    assert _oracle(good, O.INT, O.DICT_COMPRESS, 500, 2000)[1] == len(good) and _gpu(good, L.INT, L.ENC_DICT_COMPRESS, 500, 2000) == len(good)
    dec, consumed = _oracle(bad, O.INT, O.DICT_COMPRESS, 500, 2000)
    assert consumed == len(bad) and dec.view("<i4").tolist() == [entries[i] for i in ids]  # the reference's answer
    assert _gpu(bad, L.INT, L.ENC_DICT_COMPRESS, 500, 2000) == L.ERR_DATA_CORRUPTED


def test_dictionary_id_beyond_the_entries():
    """DictStruct.decodeInts gathers dict[id] without a check (DictStruct.java:57-65): an id >= entryCount reads the 4n word
    or the ids behind the entries and hands that back as a row.  Oracle and device: DATA_CORRUPTED."""
    pack = struct.pack("<iiB", 4 + 12 + 4 + 20, 3, 0) + struct.pack("<3i", 5, 6, 7) + R.rle_encode([0, 1, 2, 3, 1], 2)
    assert _oracle(pack, O.INT, O.DICT_COMPRESS, 5, 20)[1] == -2
    assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 5, 20) == L.ERR_DATA_CORRUPTED
    # the same through a run segment's value
    body = bytearray()
    R.put_uvarint(body, 9 << 1)
    body.append(3)
    R.put_uvarint(body, (1 << 1) | 1)
    body += R.pack_bits([0], 2)
    ids = struct.pack("<i", len(body)) + bytes(body)
    pack = struct.pack("<iiB", 4 + 12 + 4 + 40, 3, 0) + struct.pack("<3i", 5, 6, 7) + ids
    assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 10, 40) == L.ERR_DATA_CORRUPTED


def test_more_entries_than_rows():
    """DictStructUtil.decompress takes entryCount as it comes (DictStructUtil.java:84-88); a dictionary with more entries
    than the pack has rows decodes there.  No writer produces one (entries are the distinct rows, DictCodec.java:79-131), and
    the device sizes a pack's dictionary memory by its row count: DATA_CORRUPTED."""
    pack = struct.pack("<iiB", 4 + 20 + 4 + 12, 5, 0) + struct.pack("<5i", 1, 2, 3, 4, 5) + R.rle_encode([0, 1, 2], 3)
    dec, consumed = _oracle(pack, O.INT, O.DICT_COMPRESS, 3, 12)
    assert consumed == len(pack) and dec.view("<i4").tolist() == [1, 2, 3]
    assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 3, 12) == L.ERR_DATA_CORRUPTED


def test_numeric_dict_struct_size_that_contradicts_the_pack():
    """DictCompressCodec.decode allocates dictStructSize bytes and decodes into them without looking at the size again
    (DictCompressCodec.java:71-85; the decoders ignore their output limit, DeltaCodec.java:100-110): a word that is too large is
    harmless there, one that is too small makes the JVM write past a direct buffer.  For numbers the size follows from entryCount
    and objCount, and the device requires the word to say exactly that: DATA_CORRUPTED either way."""
    entries, ids = struct.pack("<3i", 5, 6, 7), R.rle_encode([0, 1, 2, 2, 1], 2)
    right = 4 + 12 + 4 + 20
    for size, java_ok in ((right + 64, True), (right - 4, False)):
        pack = struct.pack("<iiB", size, 3, 0) + entries + ids
        if java_ok:
            dec, consumed = _oracle(pack, O.INT, O.DICT_COMPRESS, 5, 20)
            assert consumed == len(pack) and dec.view("<i4").tolist() == [5, 6, 7, 7, 6]
        assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 5, 20) == L.ERR_DATA_CORRUPTED
    pack = struct.pack("<iiB", right, 3, 0) + entries + ids
    assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 5, 20) == len(pack)


def test_front_coded_prefix_longer_than_the_previous_string():
    """DeltaCodec.decodeStrings hands prefixLen to StringsStruct.Builder.addString, which copies that many bytes from the
    previous string's address (DeltaCodec.java:368-383, StringsStruct.java:120-131): a prefix longer than the previous string
    takes the bytes behind it, whatever they are.  Oracle and device: DATA_CORRUPTED."""
    pre, suf, tail = [0, 3, 3], [2, 1, 1], b"ab" + b"d" + b"e"
    body = R.delta_encode(pre, 32) + R.delta_encode(suf, 32) + tail
    pack = struct.pack("<i", len(body)) + body
    assert _oracle(pack, O.STRING, O.DELTA, 3, 16 + 10)[1] == -2
    assert _gpu(pack, L.STRING, L.ENC_DELTA, 3, 16 + 10) == L.ERR_DATA_CORRUPTED


def test_id_stream_with_more_rows_than_the_pack():
    """RLEPackingHybridCodec.decodeInts writes a segment's rows before it looks at the total and fails afterwards
    (RLEPackingHybridCodec.java:102-126: setInt in the loop, checkState(valIndex == valCount) at the end): the JVM has
    written past the id array by then.  The device checks every segment against the rows that are left: DATA_CORRUPTED, nothing
    written outside the pack's output (run_batch checks the bytes around it)."""
    body = bytearray()
    R.put_uvarint(body, 200 << 1)  # a run of 200 in a pack of 5 rows
    body.append(1)
    ids = struct.pack("<i", len(body)) + bytes(body)
    pack = struct.pack("<iiB", 4 + 12 + 4 + 20, 3, 0) + struct.pack("<3i", 5, 6, 7) + ids
    assert _oracle(pack, O.INT, O.DICT_COMPRESS, 5, 20)[1] < 0
    assert _gpu(pack, L.INT, L.ENC_DICT_COMPRESS, 5, 20) == L.ERR_DATA_CORRUPTED
