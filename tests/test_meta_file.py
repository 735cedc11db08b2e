"""The plugin-owned segment metadata file "<segment>_<suffix>.meta-jvector" (the verifiable part):
writer + parser of the host mirror against bytes laid out here, independently, from the reference's format
(J/JVectorWriter.java:299,512-563,573-577; J/JVectorReader.java:52-81,255-262; J/GraphNodeIdToDocMap.java:39-59,169-176;
Lucene CodecUtil index header / footer).  The jvector-owned blobs of the .data-jvector file stay out of reach (no real
segment exists to verify a reader against)."""
import importlib
import struct
import zlib

import numpy as np
import pytest


def vint(v):
    v &= 0xFFFFFFFF
    out = b""
    while v & ~0x7F:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def vlong(v):
    out = b""
    while v & ~0x7F:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def expected_file(seg_id, suffix, version, fields):
    b = struct.pack(">I", 0x3fd76c17) + vint(len("JVectorVectorsFormatMeta")) + b"JVectorVectorsFormatMeta" + struct.pack(">i", version)
    b += seg_id + bytes([len(suffix)]) + suffix.encode()
    for f in fields:
        b += struct.pack("<i", f["fieldNumber"]) * 2                       # written by writeField AND by toOutput
        b += struct.pack("<i", f["vectorEncoding"]) + struct.pack("<i", f["similarityOrd"]) + vint(f["vectorDimension"])
        b += vlong(f["vectorIndexOffset"]) + vlong(f["vectorIndexLength"]) + vlong(f["compressedVectorsOffset"]) + vlong(f["compressedVectorsLength"])
        if version >= 1:
            b += bytes([f["quantizationType"]])
        b += struct.pack("<f", f["degreeOverflow"])
        b += struct.pack("<i", 1) + vint(len(f["ord2doc"])) + vint(f["mapMaxDoc"])
        for doc in f["ord2doc"]:
            b += vint(int(doc))                                                 # -1 = five bytes ff ff ff ff 0f
    b += struct.pack("<i", -1)
    b += struct.pack(">I", 0xc02893e8) + struct.pack(">I", 0)
    return b + struct.pack(">Q", zlib.crc32(b) & 0xFFFFFFFF)


def fields_fixture():
    rng = np.random.default_rng(4)
    o2d_a = rng.permutation(300)[:200].astype(np.int32)
    o2d_a[[3, 77]] = -1                                                         # deleted ordinals
    return [
        dict(fieldNumber=2, vectorEncoding=1, similarityOrd=0, vectorDimension=768, vectorIndexOffset=60, vectorIndexLength=123456789012,
             compressedVectorsOffset=123456789072, compressedVectorsLength=320000, quantizationType=1, degreeOverflow=0.0,
             mapMaxDoc=300, ord2doc=o2d_a),
        dict(fieldNumber=5, vectorEncoding=1, similarityOrd=2, vectorDimension=16, vectorIndexOffset=60, vectorIndexLength=4096,
             compressedVectorsOffset=0, compressedVectorsLength=0, quantizationType=2, degreeOverflow=1.2000000476837158,
             mapMaxDoc=3, ord2doc=np.array([2, 0, 1], np.int32)),
    ]


def test_meta_file_bytes_and_round_trip(pkg):
    host = importlib.import_module("opensearch_jvector_amd.host")
    seg_id = bytes(range(16, 32))
    fields = fields_fixture()
    for version, suffix in ((1, "JVector_0"), (0, "")):
        fs = [dict(f) for f in fields]
        if version == 0:
            fs[1]["quantizationType"] = 0                                   # v0 has no type byte: PQ iff compressed vectors present
        got = host.meta_write(seg_id, suffix, version, fs)
        assert got == expected_file(seg_id, suffix, version, fs), "byte layout differs from the reference's format"
        ver, back = host.meta_read(got, seg_id, suffix)
        assert ver == version and len(back) == 2
        for f, g in zip(fs, back):
            for k in ("fieldNumber", "vectorEncoding", "similarityOrd", "vectorDimension", "vectorIndexOffset", "vectorIndexLength",
                      "compressedVectorsOffset", "compressedVectorsLength", "quantizationType", "mapMaxDoc"):
                assert g[k] == f[k], (k, g[k], f[k])
            assert np.float32(g["degreeOverflow"]) == np.float32(f["degreeOverflow"])
            assert np.array_equal(g["ord2doc"], f["ord2doc"])
    # empty segment metadata: header + end marker + footer
    empty = host.meta_write(seg_id, "s", 1, [])
    assert empty == expected_file(seg_id, "s", 1, []) and host.meta_read(empty, seg_id, "s") == (1, [])


def test_meta_file_rejects_corruption(pkg):
    host = importlib.import_module("opensearch_jvector_amd.host")
    seg_id = bytes(range(16))
    good = host.meta_write(seg_id, "sfx", 1, fields_fixture())
    for mutate in (lambda b: b[:40] + bytes([b[40] ^ 1]) + b[41:],             # a flipped bit in the body -> checksum
                   lambda b: b[:-3],                                             # truncated footer
                   lambda b: b"\x00" + b[1:],                                    # bad magic
                   lambda b: b + b"\x00"):                                       # trailing garbage
        with pytest.raises(host.HostError) as ei:
            host.meta_read(mutate(good), seg_id, "sfx")
        assert ei.value.code == -3                                               # IOException
    with pytest.raises(host.HostError):
        host.meta_read(good, bytes(16), "sfx")                                   # another segment's id
    with pytest.raises(host.HostError):
        host.meta_read(good, seg_id, "other")                                    # another suffix
    too_new = bytearray(expected_file(seg_id, "sfx", 1, []))
    too_new[4 + 1 + 24:4 + 1 + 24 + 4] = struct.pack(">i", 7)                   # version beyond VERSION_CURRENT
    with pytest.raises(host.HostError):
        host.meta_read(bytes(too_new), seg_id, "sfx")
