"""DEFLATE on the device (gs_inflate_members, genestrip_amd/csrc/gs_inflate_dev.hip): every member must come out byte for byte as
zlib inflates it -- stored, fixed and dynamic blocks, every compression level, several deflate blocks per member, long matches,
distances up to 32 KiB, empty members, incompressible and fuzzed text -- and a damaged member must be reported, not decoded
(ISIZE / CRC-32 as java.util.zip.GZIPInputStream checks them, B/io/StreamProvider.java:92-100).  Run with -m gpu."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import genestrip_amd as ga
from conftest import GOLDEN, bgzf

pytestmark = pytest.mark.gpu


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, pieces=1):
    """one BGZF member over `data` (<= 64 KiB), optionally several deflate blocks (full flushes between pieces)"""
    z = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    body = b""
    step = max(1, (len(data) + pieces - 1) // pieces)
    for a in range(0, len(data), step):
        body += z.compress(data[a:a + step])
        if a + step < len(data):
            body += z.flush(zlib.Z_FULL_FLUSH)
    body += z.flush()
    bsize = 18 + len(body) + 8 - 1
    assert bsize < 65536, bsize
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body + struct.pack("<II", zlib.crc32(data), len(data)))


def _roundtrip(file_bytes, want):
    members, reached = ga.bgzf_members(file_bytes)
    assert reached == len(file_bytes)
    got, st = ga.inflate_members(file_bytes, members)
    assert not st.any(), st[st != 0]
    assert got.tobytes() == want


def test_sample_fastq_as_bgzf_every_level():
    text = gzip.open(os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")).read()
    for level in (1, 4, 6, 9):
        _roundtrip(bgzf(text, level=level), text)
    _roundtrip(bgzf(text, block=4096, level=6), text)       # many small members
    _roundtrip(bgzf(text, level=0, block=60000), text)      # stored blocks


def test_block_kinds_and_shapes():
    rng = np.random.default_rng(5)
    acgt = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 65000))
    noise = bytes(rng.integers(0, 256, 60000, dtype=np.uint8))     # incompressible: stored or near-flat codes
    runs = (b"A" * 300 + b"CG" * 200 + b"\n") * 80                 # overlapping copies (distance 1 and 2), length 258 matches
    far = acgt[:32768] + acgt[:32768]                              # distances of exactly 32 KiB
    parts = []
    want = b""
    for data, kw in ((acgt, {}), (noise, dict(level=1)), (runs, dict(level=9)), (far[:65536 - 8], dict(level=9)), (b"", {}), (b"x", {}),
                     (acgt[:3000], dict(strategy=zlib.Z_FIXED)), (runs[:5000], dict(strategy=zlib.Z_FIXED)),
                     (acgt[:40000], dict(pieces=7)), (runs[:20000], dict(pieces=3, level=1)), (noise[:5000], dict(level=0)),
                     (acgt[:777], dict(strategy=zlib.Z_HUFFMAN_ONLY)), (runs[:9999], dict(strategy=zlib.Z_RLE))):
        parts.append(_member(data, **kw))
        want += data
    _roundtrip(b"".join(parts), want)


def test_fuzzed_texts():
    rng = np.random.default_rng(11)
    parts, want = [], b""
    for i in range(300):
        n = int(rng.integers(0, 20000))
        kind = i % 4
        if kind == 0:
            d = bytes(rng.choice(np.frombuffer(b"ACGTN\n@+I", dtype=np.uint8), n))
        elif kind == 1:
            d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 2:
            unit = bytes(rng.integers(65, 70, int(rng.integers(1, 40)), dtype=np.uint8))
            d = (unit * (n // max(1, len(unit)) + 1))[:n]
        else:
            d = bytes(rng.integers(0, 4, n, dtype=np.uint8))
        parts.append(_member(d, level=int(rng.integers(0, 10)), pieces=int(rng.integers(1, 4))))
        want += d
    _roundtrip(b"".join(parts), want)


def test_damaged_members_are_reported():
    text = (b"@r1\nACGTACGTAGCTAGCTAGCATCGATCGATCAGCTAGCTAGCTAGCTACGATCGATCGATCGATCGATCAGC\n+\n" + b"I" * 70 + b"\n") * 300
    good = _member(text)
    members, _ = ga.bgzf_members(good)
    ga.inflate_members(good, members)
    po, pl, isz, crc = members[0]
    for name, data, mem in (("crc", good, [(po, pl, isz, crc ^ 1)]), ("isize", good, [(po, pl, isz - 1, crc)]), ("short", good, [(po, pl - 40, isz, crc)])):
        with pytest.raises(ga.GsError) as e:
            ga.inflate_members(data, mem)
        assert e.value.code == -1 and e.value.status[0] != 0, name
    rng = np.random.default_rng(3)
    bad = 0
    for _ in range(60):  # flipped payload bits: wrong text, wrong length or a broken code -- never a silent success with other text
        d = bytearray(good)
        d[po + int(rng.integers(0, pl))] ^= 1 << int(rng.integers(0, 8))
        try:
            got, st = ga.inflate_members(bytes(d), members)
            assert got.tobytes() == text
        except ga.GsError as e:
            assert e.value.code == -1 if hasattr(e, "value") else e.code == -1
            bad += 1
    assert bad >= 55
    # one bad member among good ones: the others still come out
    three = good + _member(b"second member\n" * 50) + good
    mem3, _ = ga.bgzf_members(three)
    mem3[1] = (mem3[1][0], mem3[1][1], mem3[1][2], mem3[1][3] ^ 0x80)
    with pytest.raises(ga.GsError) as e:
        ga.inflate_members(three, mem3)
    assert list(e.value.status != 0) == [False, True, False]
