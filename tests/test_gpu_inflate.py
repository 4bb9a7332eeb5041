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


def test_distances_around_the_ring_and_groups_of_short_matches():
    """The token decoder keeps the last 2 KiB of text in LDS and copies short matches side by side: distances on both sides of the
    ring limit, sources that end inside the group they belong to, matches of every length, groups that end in an end-of-block."""
    rng = np.random.default_rng(77)
    parts, want = [], b""
    for dist0 in (1500, 1800, 1900, 1984, 2000, 2047, 2048, 2049, 2100, 2500, 4096, 9000):
        text = bytearray(rng.integers(0, 256, dist0 + 300, dtype=np.uint8).tobytes())
        at = len(text)
        for k in range(400):  # copies of 3..258 bytes from `dist0 +- a little` back, separated by a few fresh bytes
            ln = int(rng.integers(3, 259)) if k % 7 == 0 else int(rng.integers(3, 12))
            d = dist0 + int(rng.integers(-40, 41))
            if d > at or at + ln > 65000:
                break
            text += text[at - d:at - d + ln] if d >= ln else (text[at - d:at] * (ln // d + 1))[:ln]
            text += rng.integers(0, 256, int(rng.integers(0, 4)), dtype=np.uint8).tobytes()
            at = len(text)
        data = bytes(text[:65000])
        for level in (1, 9):
            parts.append(_member(data, level=level))
            want += data
    acgt = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 64000))  # level 1 on four letters: almost only matches of 3..8 bytes
    for level in (1, 2, 3):
        parts.append(_member(acgt, level=level))
        want += acgt
    tiny = b"".join(_member(acgt[a:a + n], level=6) for a, n in ((0, 1), (1, 2), (3, 3), (6, 7), (13, 64), (77, 65), (142, 300)))  # short last groups
    parts.append(tiny)
    want += acgt[0:1] + acgt[1:3] + acgt[3:6] + acgt[6:13] + acgt[13:77] + acgt[77:142] + acgt[142:442]
    _roundtrip(b"".join(parts), want)


def test_a_literal_behind_more_run_than_the_ring_holds():
    """a handful of maximal matches in 64 bits of input (a run of one byte) produce more text than the LDS ring holds; a literal behind
    them in the same group must still be what a later match copies from the ring (found by tools/gunzip_fuzz.py: the newline between
    a run of 'A' and the first FASTQ record came back as 'A' in every record that copied it)"""
    rng = np.random.default_rng(31)
    for run in (1500, 2300, 4000, 9000):
        for level in (1, 6, 9):
            rec = b"".join(b"@r%07d/1\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 100)) + b"\n+\n" + b"F" * 100 + b"\n" for i in range(40))
            data = (b"A" * run + b"\n" + rec) * 4
            data = data[:65000]
            _roundtrip(_member(data, level), data)
            z = gzip.compress((b"A" * run + b"\n" + rec) * 300, compresslevel=level, mtime=0)
            want = (b"A" * run + b"\n" + rec) * 300
            got, _ = ga.gunzip_device(z, len(want))
            assert got.tobytes() == want, (run, level)


def test_chains_of_copies_inside_one_group():
    """The text of a group of up to 256 symbols is written one lane per symbol: a symbol whose source lies among the same 64 takes it from
    that lane, after as many rounds of pointer jumping as the chain of copies is deep.  Texts made of periods of 1 .. 70 bytes (overlapping
    copies, runs), of copies of copies a few bytes back, of groups of 65 .. 256 symbols with a long match in the middle, and of far
    sources (text, not ring) next to near ones -- as BGZF members (byte ring of 2 KiB) and as single-member streams (symbol ring of 1 Ki,
    markers in front of every segment)."""
    rng = np.random.default_rng(2024)
    texts = []
    t = bytearray()
    for period in list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 70]:  # a b a b a b ...: one literal period, then a copy that overlaps itself
        unit = bytes(rng.integers(65, 91, period, dtype=np.uint8))
        for reps in (2, 3, 5, 17, 40):
            t += unit * reps + bytes(rng.integers(97, 123, int(rng.integers(1, 4)), dtype=np.uint8))
    texts.append(bytes(t))
    t = bytearray(rng.integers(65, 69, 40, dtype=np.uint8).tobytes())
    while len(t) < 60000:  # every piece a copy of something 3 .. 60 bytes back, itself a copy: chains of three, four, five inside 64 symbols
        d = int(rng.integers(3, 61))
        ln = int(rng.integers(3, 10))
        src = len(t) - d
        for i in range(ln):
            t.append(t[src + i])
        if rng.integers(0, 5) == 0:
            t += bytes(rng.integers(65, 69, 1, dtype=np.uint8))
    texts.append(bytes(t))
    base = bytes(rng.integers(65, 69, 5000, dtype=np.uint8))
    t = bytearray(base)
    while len(t) < 62000:  # short near copies, a long copy (70 .. 250 symbols: two to four rounds of 64), far copies beyond both rings
        kind = int(rng.integers(0, 4))
        d = (int(rng.integers(3, 40)), int(rng.integers(900, 1200)), int(rng.integers(1900, 2300)), int(rng.integers(3000, 4900)))[kind]
        ln = int(rng.integers(70, 251)) if rng.integers(0, 6) == 0 else int(rng.integers(3, 9))
        src = len(t) - d
        for i in range(ln):
            t.append(t[src + i])
        t += bytes(rng.integers(65, 69, int(rng.integers(0, 3)), dtype=np.uint8))
    texts.append(bytes(t))
    parts, want = [], b""
    for text in texts:
        for level in (1, 6, 9):
            data = text[:65000]
            parts.append(_member(data, level=level))
            want += data
    _roundtrip(b"".join(parts), want)
    for level in (1, 6, 9):
        whole = b"".join(texts) * 6
        for chunk in ("8192", "1048576"):
            os.environ["GS_GUNZIP_CHUNK"] = chunk
            try:
                got, _ = ga.gunzip_device(gzip.compress(whole, compresslevel=level, mtime=0), len(whole))
            finally:
                del os.environ["GS_GUNZIP_CHUNK"]
            assert got.tobytes() == whole, (level, chunk)


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


def test_canonical_decoder_gives_the_same_text(monkeypatch):
    """the table-free decoder that takes over when a Huffman code needs more sub-table room than a wave has (no stream zlib writes
    does), forced onto everything"""
    monkeypatch.setenv("GS_INFLATE_FORCE_SLOW", "1")
    text = gzip.open(os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")).read()
    _roundtrip(bgzf(text, level=6), text)
    _roundtrip(bgzf(text[:200000], level=9, block=20000), text[:200000])
    test_block_kinds_and_shapes()


def _fastq(seq, off, start=0, multiline=False, final_newline=True):
    out = []
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])].tobytes()
        q = b"F" * len(s)
        if multiline:
            h = len(s) // 2
            out.append(b"@r%d\n%s\n%s\n+\n%s\n%s\n" % (start + i, s[:h], s[h:], q[:h], q[h:]))
        else:
            out.append(b"@r%d some text\n%s\n+\n%s\n" % (start + i, s, q))
    t = b"".join(out)
    return t if final_newline else t[:-1]


@pytest.mark.parametrize("shape", ["plain", "no_final_newline", "truncated", "multiline", "small_feeds"])
def test_bgzf_files_through_the_device_inflater(tmp_path, monkeypatch, shape):
    """gs_host_match_files on BGZF input: the members are inflated on the device (no per-read outputs asked for); the table and
    the totals must equal the host decoder's and the oracle's -- also when the file ends in the middle of a record or without a
    newline (the leftover goes to the reference-exact parser) and when it is not four-line FASTQ (the device scan refuses, the
    general path takes over)"""
    from genestrip_amd import host, synth
    from oracle import gs_oracle as orc
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=3)
    seq, off = synth.reads_host(db.genomes, 30000, read_len=150, seed=9)
    text = _fastq(seq, off, multiline=shape == "multiline", final_newline=shape != "no_final_newline")
    if shape == "truncated":
        text = text[:-200]
    if shape == "small_feeds":
        monkeypatch.setenv("GS_HOST_BGZF_TEXT", "300000")  # many feeds: tails carried from feed to feed on the device
    path = tmp_path / "reads.fastq.gz"
    path.write_bytes(bgzf(text, level=4))
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    monkeypatch.setenv("GS_DEVICE_INFLATE", "1")
    t_dev, _, tot_dev = host.match_files(store, [str(path)])
    monkeypatch.setenv("GS_DEVICE_INFLATE", "0")
    t_host, _, tot_host = host.match_files(store, [str(path)])
    assert np.array_equal(t_dev, t_host)
    assert (tot_dev.reads, tot_dev.kmers, tot_dev.bps) == (tot_host.reads, tot_host.kmers, tot_host.bps)
    if shape in ("plain", "small_feeds"):
        orun = orc.MatchRun(orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi))
        orun.submit(seq, off, threads=8, per_read=False)
        assert np.array_equal(t_dev, orun.finish()[0]) and tot_dev.reads == 30000
    store.close()


@pytest.mark.parametrize("shape", ["plain", "no_final_newline", "multiline", "small_feeds"])
def test_bgzf_files_with_per_read_outputs_through_the_device_inflater(tmp_path, monkeypatch, shape):
    """gs_host_match_files on BGZF input WITH per-read outputs (Kraken-style lines, the filtered FASTQ): the text of a feed comes back
    once for the writers; both files, the table and the totals must equal the host-decoder run byte for byte"""
    from genestrip_amd import host, synth
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=3)
    seq, off = synth.reads_host(db.genomes, 20000, read_len=150, seed=19)
    text = _fastq(seq, off, multiline=shape == "multiline", final_newline=shape != "no_final_newline")
    if shape == "small_feeds":
        monkeypatch.setenv("GS_HOST_BGZF_TEXT", "300000")
    path = tmp_path / "reads.fastq.gz"
    path.write_bytes(bgzf(text, level=1))
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    cfg = ga.MatchConfig(classify=True)
    taxids = [f"t{i}" for i in range(db.n_values)]
    got = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("GS_DEVICE_INFLATE", dev)
        fo, ko = tmp_path / f"filtered{dev}.fastq", tmp_path / f"kraken{dev}.txt"
        t, _, tot = host.match_files(store, [str(path)], config=cfg, filtered_path=fo, kraken_out_path=ko, taxids=taxids)
        got[dev] = (t, (tot.reads, tot.kmers, tot.bps), fo.read_bytes(), ko.read_bytes())
    assert np.array_equal(got["1"][0], got["0"][0]) and got["1"][1] == got["0"][1]
    assert got["1"][2] == got["0"][2] and got["1"][3] == got["0"][3]
    assert len(got["1"][3].splitlines()) == 20000 and len(got["1"][2]) > 0
    store.close()


def test_a_corrupt_bgzf_member_fails_the_file(tmp_path, monkeypatch):
    from genestrip_amd import host, synth
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=3)
    seq, off = synth.reads_host(db.genomes, 5000, read_len=150, seed=9)
    data = bytearray(bgzf(_fastq(seq, off), level=6))
    members, _ = ga.bgzf_members(bytes(data))
    po, pl, _, _ = members[len(members) // 2]
    data[po + pl // 2] ^= 0x55
    path = tmp_path / "bad.fastq.gz"
    path.write_bytes(bytes(data))
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    monkeypatch.setenv("GS_DEVICE_INFLATE", "1")
    with pytest.raises(Exception) as e:
        host.match_files(store, [str(path)])
    assert "corrupt" in str(e.value)
    store.close()


# ---- single-member gzip streams (gzip, pigz) on the device: gs_gunzip_device
def _gz(data, level=6, **kw):
    return gzip.compress(data, compresslevel=level, mtime=0)


def _fastq_like(n, seed):
    rng = np.random.default_rng(seed)
    acgt = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n * 150).tobytes()
    qual = rng.choice(np.frombuffer(b"FFFFFFF:,#", dtype=np.uint8), n * 150).tobytes()
    return b"".join(b"@read%07d/1\n" % i + acgt[i * 150:(i + 1) * 150] + b"\n+\n" + qual[i * 150:(i + 1) * 150] + b"\n" for i in range(n))


@pytest.mark.parametrize("level", [1, 6, 9])
def test_single_member_gzip_equals_zlib(level):
    """block starts found speculatively, segments decoded side by side with window markers, windows resolved in a second pass: the text
    must be what zlib gives, for every level, with many segments (small chunks) and with one"""
    text = _fastq_like(20000, 7 + level)
    z = _gz(text, level)
    for chunk in ("4096", "16384", "1048576"):
        os.environ["GS_GUNZIP_CHUNK"] = chunk
        try:
            got, info = ga.gunzip_device(z, len(text))
        finally:
            del os.environ["GS_GUNZIP_CHUNK"]
        assert got.tobytes() == text, (level, chunk)
        assert info[0] >= 1 and (chunk != "4096" or info[0] > 10), info


def test_golden_sample_fastq_gz_on_the_device(monkeypatch):
    """the reference's own fixture (sample.fastq.gz of the human_virus project) as it lies on disk -- whatever wrote it -- through the device
    decoder, whole and in small batches"""
    raw = open(os.path.join(GOLDEN, "human_virus", "sample.fastq.gz"), "rb").read()
    text = gzip.decompress(raw)
    got, info = ga.gunzip_device(raw, len(text))
    assert got.tobytes() == text
    monkeypatch.setenv("GS_GUNZIP_SLOTS", "4")
    monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
    got, info = ga.gunzip_device(raw, len(text))
    assert got.tobytes() == text


@pytest.mark.parametrize("upload", ["whole", "per_batch"])
@pytest.mark.parametrize("slots", ["3", "17"])
def test_single_member_gzip_in_many_batches(monkeypatch, slots, upload):
    """the stream in batches of a few chunks: every batch starts at the block the one before stopped at, its windows are resolved through
    the window the batch before left, the CRC-32 is folded into a running register.  The compressed bytes: uploaded whole by the
    decoder's own thread while the batches run (the default up to 16 GiB), or span by span with every batch"""
    if upload == "per_batch":
        monkeypatch.setenv("GS_GUNZIP_WHOLE_MAX", "0")
    monkeypatch.setenv("GS_GUNZIP_SLOTS", slots)
    monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
    for level in (1, 6):
        text = _fastq_like(12000, 40 + level)
        got, info = ga.gunzip_device(_gz(text, level), len(text))
        assert got.tobytes() == text
        assert info[2] > 5, info  # batches


def test_a_decoder_given_up_half_way_is_used_again(monkeypatch):
    """gs_gunzipper_park stops the upload thread of a stream that is not read to its end; the object then takes the next stream"""
    import ctypes
    from genestrip_amd import binding
    lib = binding.lib()
    monkeypatch.setenv("GS_GUNZIP_SLOTS", "4")
    monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
    a, b = _fastq_like(20000, 71), _fastq_like(5000, 72)
    za, zb = _gz(a, 1), _gz(b, 6)
    buf_a, buf_b = ctypes.create_string_buffer(za, len(za)), ctypes.create_string_buffer(zb, len(zb))
    g = ctypes.c_void_p()
    assert lib.gs_gunzipper_open(ctypes.byref(g), 0, buf_a, len(za)) == 0
    d, n, last = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int()
    assert lib.gs_gunzipper_next(g, 0, ctypes.byref(d), ctypes.byref(n), ctypes.byref(last)) == 0 and last.value == 0  # (one batch of many)
    assert lib.gs_gunzipper_park(g) == 0
    del buf_a  # (the first stream's bytes are gone)
    assert lib.gs_gunzipper_reopen(g, buf_b, len(zb)) == 0
    got = bytearray()
    keep = 0
    while True:
        assert lib.gs_gunzipper_next(g, 0, ctypes.byref(d), ctypes.byref(n), ctypes.byref(last)) == 0
        out = ctypes.create_string_buffer(max(1, n.value))
        assert lib.gs_device_fetch(0, d, out, n.value) == 0
        got += out.raw[:n.value]
        if last.value:
            break
    assert bytes(got) == b
    assert lib.gs_gunzipper_close(g) == 0


def test_a_small_first_batch_for_callers_with_writers():
    """gs_gunzipper_first_span: the first batch takes at most that many compressed bytes (first text early), the rest as usual"""
    import ctypes
    from genestrip_amd import binding
    lib = binding.lib()
    text = _fastq_like(20000, 81)
    z = _gz(text, 6)
    buf = ctypes.create_string_buffer(z, len(z))
    g = ctypes.c_void_p()
    assert lib.gs_gunzipper_open(ctypes.byref(g), 0, buf, len(z)) == 0
    assert lib.gs_gunzipper_first_span(g, 65536) == 0
    d, n, last = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int()
    got, sizes = bytearray(), []
    while True:
        assert lib.gs_gunzipper_next(g, 0, ctypes.byref(d), ctypes.byref(n), ctypes.byref(last)) == 0
        out = ctypes.create_string_buffer(max(1, n.value))
        assert lib.gs_device_fetch(0, d, out, n.value) == 0
        got += out.raw[:n.value]
        sizes.append(n.value)
        if last.value:
            break
    assert bytes(got) == text
    assert len(sizes) == 2 and sizes[0] < sizes[1] and sizes[0] < 16 * 65536, sizes
    assert lib.gs_gunzipper_close(g) == 0


def test_single_member_gzip_with_more_blocks_than_wave_slots(monkeypatch):
    """every block start is found; with more blocks than wave slots a segment takes floor(blocks / slots) of them and the blocks left
    over wait for the next batch"""
    monkeypatch.setenv("GS_GUNZIP_SLOTS", "4")
    monkeypatch.setenv("GS_GUNZIP_CHUNK", "1048576")
    for level in (1, 6):
        text = _fastq_like(30000, 50 + level)
        got, info = ga.gunzip_device(_gz(text, level), len(text))
        assert got.tobytes() == text
        assert 4 <= info[0] <= 4 * info[2], info  # at most four segments in each batch


def test_single_member_gzip_shapes():
    rng = np.random.default_rng(3)
    for name, data in (("empty", b""), ("one byte", b"x"), ("runs", (b"A" * 1000 + b"\n") * 3000),
                       ("far copies", bytes(rng.integers(65, 91, 40000, dtype=np.uint8)) * 30),
                       ("header fields", _fastq_like(3000, 5))):
        z = _gz(data)
        if name == "header fields":  # FNAME + FCOMMENT + FEXTRA in front of the deflate stream
            body = z[10:]
            z = z[:3] + bytes([4 | 8 | 16]) + z[4:10] + struct.pack("<H", 5) + b"extra" + b"name.fastq\0" + b"a comment\0" + body
        got, info = ga.gunzip_device(z, len(data))
        assert got.tobytes() == data, name


def test_single_member_gzip_what_the_device_path_refuses_or_reports():
    text = _fastq_like(5000, 11)
    z = _gz(text)
    # members behind one another (cat a.gz b.gz): the text of one behind the other, every member's CRC-32 and ISIZE checked; what is
    # not a member behind the last trailer is ignored, as java.util.zip.GZIPInputStream does
    other = _fastq_like(3000, 12)
    got, info = ga.gunzip_device(z + _gz(other, 1) + _gz(b"") + z, 2 * len(text) + len(other))
    assert got.tobytes() == text + other + text
    got, _ = ga.gunzip_device(z + b"\0" * 100, len(text))
    assert got.tobytes() == text
    two = bytearray(z + z)
    two[-6] ^= 1  # the second member's CRC-32
    with pytest.raises(ga.GsError) as e:
        ga.gunzip_device(bytes(two), 2 * len(text))
    assert e.value.code == -1 and "CRC" in str(e.value)
    bad = bytearray(z)
    bad[-6] ^= 1  # the CRC-32 of the trailer
    with pytest.raises(ga.GsError) as e:
        ga.gunzip_device(bytes(bad), len(text))
    assert e.value.code == -1 and "CRC" in str(e.value)
    rng = np.random.default_rng(5)
    n_bad = 0
    for _ in range(20):  # a flipped payload bit: never a silent success with other text
        d = bytearray(z)
        d[20 + int(rng.integers(0, len(z) - 40))] ^= 1 << int(rng.integers(0, 8))
        try:
            got, _ = ga.gunzip_device(bytes(d), len(text) + 100000)
            assert got.tobytes() == text
        except ga.GsError as ex:
            assert ex.code in (-1, -4)
            n_bad += 1
    assert n_bad >= 18


def test_a_segment_that_outgrows_its_room_is_decoded_again(monkeypatch):
    """long runs inside ordinary FASTQ: the segments that hold them expand a hundred times more than their neighbours"""
    monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
    monkeypatch.setenv("GS_GUNZIP_RATIO", "6")
    body = _fastq_like(6000, 21)
    data = body[:700000] + b"@poly\n" + b"A" * 3_000_000 + b"\n+\n" + b"#" * 3_000_000 + b"\n" + body[700000:]
    z = _gz(data, 6)
    got, info = ga.gunzip_device(z, len(data))
    assert got.tobytes() == data and info[0] > 10


def test_binary_data_and_mirages(monkeypatch):
    """random bytes compress to stored / near-flat blocks, and with GS_GUNZIP_ANY_BYTES=1 every parsable header counts as a block start:
    a start that was a mirage makes the segment in front of it run past it, which is decoded again"""
    rng = np.random.default_rng(9)
    data = bytes(rng.integers(0, 256, 300000, dtype=np.uint8)) + _fastq_like(4000, 2) + bytes(rng.integers(0, 64, 500000, dtype=np.uint8))
    z = _gz(data)
    for any_bytes in ("0", "1"):
        monkeypatch.setenv("GS_GUNZIP_ANY_BYTES", any_bytes)
        monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
        got, info = ga.gunzip_device(z, len(data))
        assert got.tobytes() == data, any_bytes


@pytest.mark.parametrize("batches", [False, True])
@pytest.mark.parametrize("outputs", [False, True])
def test_gzip_files_through_the_device_gunzip(tmp_path, monkeypatch, outputs, batches):
    """gs_host_match_files on a plain (single-member) .gz: inflated on the device as a whole; table, totals and per-read files must
    equal the host-decoder run, also when the file ends in the middle of a record"""
    from genestrip_amd import host, synth
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=3)
    seq, off = synth.reads_host(db.genomes, 20000, read_len=150, seed=29)
    text = _fastq(seq, off)[:-77]
    path = tmp_path / "reads.fastq.gz"
    path.write_bytes(_gz(text, 6))
    monkeypatch.setenv("GS_HOST_BGZF_TEXT", "700000")  # several slices
    if batches:  # ... and several batches of the stream: records that straddle a batch are carried on the device
        monkeypatch.setenv("GS_GUNZIP_SLOTS", "7")
        monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    got = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("GS_DEVICE_GUNZIP", dev)
        kw = {}
        if outputs:
            kw = dict(filtered_path=tmp_path / f"f{dev}.fastq", kraken_out_path=tmp_path / f"k{dev}.txt", taxids=[f"t{i}" for i in range(db.n_values)])
        t, _, tot = host.match_files(store, [str(path)], **kw)
        files = tuple(open(p, "rb").read() for p in (kw.get("filtered_path"), kw.get("kraken_out_path")) if p)
        got[dev] = (t, (tot.reads, tot.kmers, tot.bps), files)
    assert np.array_equal(got["1"][0], got["0"][0]) and got["1"][1] == got["0"][1] and got["1"][2] == got["0"][2]
    assert got["1"][1][0] in (19999, 20000)  # (the last record is cut short: what becomes of it is the reference parser's business)
    store.close()


def test_two_gzip_members_in_one_file(tmp_path, monkeypatch):
    """`cat a.gz b.gz`: member behind member on the device (each with its own CRC-32 and ISIZE, no window across the seam) -- table and
    totals as the host-decoder run over the whole file"""
    from genestrip_amd import host, synth
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=3)
    seq, off = synth.reads_host(db.genomes, 9000, read_len=150, seed=31)
    text = _fastq(seq, off)
    cut = text.index(b"\n@", len(text) // 2) + 1
    path = tmp_path / "two.fastq.gz"
    path.write_bytes(_gz(text[:cut], 6) + _gz(text[cut:], 1))
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    got = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("GS_DEVICE_GUNZIP", dev)
        t, _, tot = host.match_files(store, [str(path)])
        got[dev] = (t, (tot.reads, tot.kmers, tot.bps))
    assert np.array_equal(got["1"][0], got["0"][0]) and got["1"][1] == got["0"][1] and got["1"][1][0] == 9000
    store.close()


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("mode", ["", "bgzf"])
def test_randomised_streams_equal_zlib(mode):
    """a short run of tools/gunzip_fuzz.py (texts of several kinds, every level / strategy, flushes, several members, the decoder's
    geometry drawn at random): zlib's text, or a refusal to the host decoders -- never other text"""
    assert _tool("gunzip_fuzz").main(40 if mode == "" else 20, 11, mode) == 0


def test_randomised_compressed_files_equal_the_plain_file():
    """a short run of tools/host_gz_fuzz.py: FASTQ of random shape as one gzip member / several / BGZF through `match` (with and
    without per-read lines) and `filter` with writeback, random batch / slice / chunk sizes, against the plain file"""
    assert _tool("host_gz_fuzz").main(4, 5) == 0
