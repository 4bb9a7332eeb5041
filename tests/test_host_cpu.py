"""CPU tests of the C++ host layer (libgshost.so): parser semantics against the oracle's restatement and the
reference fixtures, Double.toString formatting, completeResults + CSV integer columns."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN
from genestrip_amd import host
from oracle import gs_oracle as orc


def _read_all(path, k, fasta=None, max_reads=1 << 20):
    r = host.FastqReader(path, k=k, fasta=fasta)
    seqs, descs, quals = [], [], []
    first = 0
    while True:
        b = r.next_batch(max_reads=max_reads)
        if b is None:
            break
        assert b["first_read_no"] == first
        first += b["n_reads"]
        for i in range(b["n_reads"]):
            seqs.append(b["seq"][int(b["seq_off"][i]):int(b["seq_off"][i + 1])].tobytes())
            descs.append(b["desc"][int(b["desc_off"][i]):int(b["desc_off"][i + 1])].tobytes())
            quals.append(b["qual"][int(b["qual_off"][i]):int(b["qual_off"][i + 1])].tobytes())
    tot = r.totals()
    r.close()
    return seqs, descs, quals, tot


def _oracle_all(data, k, fasta=False):
    rd = orc.parse_fastq(data, fasta=fasta, k=k)

    def col(name):
        o = rd[name + "_off"]
        return [rd[name][int(o[i]):int(o[i + 1])].tobytes() for i in range(rd["n_reads"])]
    return col("seq"), col("desc"), col("qual"), (rd["n_reads"], rd["total_kmers"], rd["total_bps"])


def test_k6_sample_fastq_gz_totals_and_records():
    path = os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")
    got = _read_all(path, 31, max_reads=1000)  # several batches
    assert got[3] == (6565, 461305, 658255)
    want = _oracle_all(gzip.open(path).read(), 31)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[3] == want[3]


def test_k7_simple_fixture():
    seqs, descs, quals, tot = _read_all(os.path.join(GOLDEN, "fastq", "SimpleTest.fastq"), 2)
    assert descs == [b"@S", b"@T"]
    assert seqs == [b"GATTTGGGGTTCAAAGCAGTATCGATCAAATAGTAAATCCATTTGTTCAACTCACAGTTT", b"CGAT"]
    assert quals == [b"!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65", b"!**>"]


@pytest.mark.parametrize("name,data,fasta", [
    ("a.fastq", b"@r1 x\nAC\0GT\r\n+\n!!!!!\n@r2\nACGT\n+\nIIII", False),
    ("b.fastq", b"@r1\nACGT\nAC\n+r1\nII\nII\nII\n@r2\n\n+\n\n@r3\nNNNN\n+\n!!!!\n", False),
    ("c.fasta", b">s1 d\nACGT\nAC\n>s2\nGG\n", True),
    ("d.fa", b">only\nACGTACGT", True),
    ("e.fastq", b"", False),
])
def test_parser_edge_cases_match_oracle(tmp_path, name, data, fasta):
    p = tmp_path / name
    p.write_bytes(data)
    got = _read_all(str(p), 2)  # type by suffix
    want = _oracle_all(data, 2, fasta)
    assert got == want


def test_gzip_by_content_and_multi_batch(tmp_path):
    rng = np.random.default_rng(1)
    recs = []
    for i in range(5000):
        L = int(rng.integers(1, 300))
        s = bytes(rng.choice(list(b"ACGTN"), L).tolist())
        recs.append(b"@read%d extra\n%s\n+\n%s\n" % (i, s, b"I" * L))
    data = b"".join(recs)
    p = tmp_path / "x.fastq.gz"
    with gzip.open(p, "wb") as f:
        f.write(data)
    got = _read_all(str(p), 31, max_reads=777)
    assert got == _oracle_all(data, 31)


def test_java_double_to_string():
    cases = {1.0: "1.0", 0.001: "0.001", 1e-4: "1.0E-4", 1e7: "1.0E7", 123456.789: "123456.789",
             9999999.0: "9999999.0", 1e10: "1.0E10", 0.1 + 0.2: "0.30000000000000004", 100.0: "100.0",
             1.5e-5: "1.5E-5", -2.5: "-2.5", 0.0: "0.0", 1234567.0: "1234567.0", 12345678.0: "1.2345678E7",
             2.0 / 3.0: "0.6666666666666666", 150.0: "150.0"}
    # (double columns are outside the bit-exact contract, SURVEY 8c: e.g. Double.MIN_VALUE prints 4.9E-324 in Java)
    for v, s in cases.items():
        assert host.java_double(v) == s, (v, host.java_double(v), s)
    assert host.java_double(float("nan")) == "NaN"


def test_csv_complete_results(tmp_path):
    # tree: 0 root -> 1 genus -> {2, 3} species ; 4 = second genus without hits; 5 = value without node
    parent = [-1, 0, 1, 1, 0, -2]
    taxids = ["1", "10", "100", "101", "20", "999"]
    names = ["root", "G", "S0", "S1", "H", None]
    ranks = ["no rank", "genus", "species", "species", "genus", None]
    dbk = [5, 50, 500, 400, 70, 0]
    t = np.zeros((6, 10), dtype=np.int64)
    d = np.zeros((6, 4))
    #        reads rk   kmers uniq contigs sq   maxc r1k bps  readno
    t[2] = [3, 300, 310, 120, 4, 30000, 100, 5, 450, 7]
    t[3] = [1, 90, 95, 60, 2, 5000, 60, 2, 150, 9]
    t[1] = [0, 0, 12, 10, 3, 60, 5, 3, 0, 11]
    t[5] = [9, 9, 9, 9, 9, 9, 9, 9, 9, 9]  # no tree node: never reported
    d[2] = [0.3, 0.05, 0.6, 0.2]
    tot = host.Totals(1000, 120000, 150000, 0, 0, 0, 0)
    out = tmp_path / "r.csv"
    host.write_csv(out, parent, taxids, dbk, 1025, t, d, tot, names=names, ranks=ranks)
    lines = out.read_text().split("\n")
    head = lines[0].split(";")
    assert head[:9] == ["pos", "level", "name", "rank", "taxid", "reads", "kmers from reads", "kmers", "unique kmers"]
    assert lines[0].endswith("acc. class error std. dev.;")
    rows = [l.split(";") for l in lines[1:] if l]
    assert all(len(r) == len(head) for r in rows)
    col = {n: i for i, n in enumerate(head)}
    total = rows[0]
    assert total[col["name"]] == "TOTAL" and total[col["reads"]] == "1000" and total[col["kmers"]] == "120000"
    assert total[col["reads bps"]] == "150000" and total[col["db kmers"]] == "1025" and total[col["avg. read length"]] == "150.0"
    assert total[col["taxid"]] == "" and total[col["acc. reads"]] == ""
    # rows in tree order: root (added as missing ancestor), genus 10, species 100, 101; genus 20 and 999 absent
    assert [r[col["taxid"]] for r in rows[1:]] == ["1", "10", "100", "101"]
    assert [r[col["pos"]] for r in rows] == ["0", "1", "2", "3", "4"]
    assert [r[col["level"]] for r in rows[1:]] == ["0", "1", "2", "2"]
    root, genus, s0, s1 = rows[1:]
    assert root[col["reads"]] == "0" and root[col["kmers"]] == "0" and root[col["parent taxid"]] == ""
    assert genus[col["kmers"]] == "12" and genus[col["contigs"]] == "3" and genus[col["parent taxid"]] == "1"
    assert s0[col["reads"]] == "3" and s0[col["unique kmers"]] == "120" and s0[col["max contig length"]] == "100"
    # accumulated = own + descendants
    assert root[col["acc. reads"]] == "4" and genus[col["acc. reads"]] == "4" and s0[col["acc. reads"]] == "3"
    assert root[col["acc. kmers"]] == str(310 + 95 + 12) and genus[col["acc. kmers"]] == str(310 + 95 + 12)
    assert root[col["acc. reads bps"]] == "600" and s1[col["acc. read >=1 kmer"]] == "2"
    assert root[col["acc. reads kmers"]] == "390"
    # doubles follow the reference formulas and Double.toString
    assert s0[col["average contig length"]] == host.java_double(310 / 4)
    assert s0[col["avg. read length"]] == "150.0" and s0[col["db coverage"]] == host.java_double(120 / 500)
    assert s0[col["mean error"]] == host.java_double(0.3 / 3)
    assert s0[col["norm. kmers"]] == host.java_double(310 / 500)
    assert genus[col["acc. norm. kmers"]] == host.java_double(12 / 50 + 310 / 500 + 95 / 400)
    assert root[col["average contig length"]] == "" and root[col["mean error"]] == ""  # NaN -> blank
    # experimental "max kmer counts" column (maxKMerResCounts > 0): N values per row, TOTAL row = overall
    mc = np.zeros((7, 3), dtype=np.int16)
    mc[2] = [9, 4, 1]
    mc[6] = [9, 7, 4]
    out2 = tmp_path / "r2.csv"
    host.write_csv(out2, parent, taxids, dbk, 1025, t, d, tot, names=names, ranks=ranks, max_kmer_counts=mc)
    l2 = out2.read_text().split("\n")
    assert l2[0].endswith("acc. class error std. dev.;max kmer counts;")
    assert l2[1].endswith(";9;7;4;") and l2[4].endswith(";9;4;1;") and l2[2].endswith(";;")


# ------------------------------------------------------------------ the ingest path's gzip decoder (gs_inflate.h)
def _gz(data, level=6, strategy=0, wbits=31):
    import zlib
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return c.compress(data) + c.flush()


def _gz_inputs():
    import random
    rnd = random.Random(7)
    fastq = b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, bytes(rnd.choice(b"ACGT") for _ in range(150)),
                                                bytes(rnd.choice(b"F:,#") for _ in range(150))) for i in range(1500))
    return {
        "empty": b"", "one byte": b"A", "fastq": fastq, "random": bytes(rnd.getrandbits(8) for _ in range(100_000)),
        "run": b"A" * 300_000, "period 7": (b"ACGTTGC" * 50_000), "mixed": fastq[:50_000] + bytes(rnd.getrandbits(8) for _ in range(70_000)) + fastq[:90_000],
    }


@pytest.mark.parametrize("block", [1, 7, 4096, 1 << 20])
def test_gunzip_equals_zlib_on_all_block_types(block):
    """stored / fixed / dynamic blocks, every compression level, output cut into blocks of any size (the decoder must
    resume inside a DEFLATE block and inside a match)"""
    import zlib
    for name, data in _gz_inputs().items():
        if block < 4096 and len(data) > 120_000:
            data = data[:120_000]
        for level, strategy in ((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (4, zlib.Z_RLE)):
            got = host.gunzip(_gz(data, level, strategy), len(data), block)
            assert got == data, (name, level, strategy, block)


def test_gunzip_members_header_fields_and_corruption():
    import gzip
    import io
    d = _gz_inputs()
    a, b = d["fastq"], d["mixed"]
    assert host.gunzip(_gz(a, 6) + _gz(b, 1) + _gz(b"", 9), len(a) + len(b), 5000) == a + b  # concatenated members
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads.fastq", mode="wb", fileobj=buf, mtime=123) as f:  # FNAME header field
        f.write(a)
    assert host.gunzip(buf.getvalue(), len(a), 333) == a
    raw = bytearray(_gz(a, 6))
    for cut in (len(raw) // 2, len(raw) - 3, 5):
        with pytest.raises(RuntimeError):
            host.gunzip(bytes(raw[:cut]), len(a))       # truncated
    flip = bytearray(raw)
    flip[len(flip) // 2] ^= 0x10
    with pytest.raises(RuntimeError):
        host.gunzip(bytes(flip), len(a) + 1000)          # damaged data: bad code or CRC mismatch
    crc = bytearray(raw)
    crc[-6] ^= 1
    with pytest.raises(RuntimeError):
        host.gunzip(bytes(crc), len(a))                  # trailer CRC
    with pytest.raises(RuntimeError):
        host.gunzip(b"not a gzip stream at all", 100)


@pytest.mark.parametrize("threads,chunk", [(1, 65536), (3, 65536), (4, 200_000), (8, 1 << 20)])
def test_parallel_gunzip_equals_zlib(threads, chunk):
    """speculative mid-stream starts, marker resolution and the in-order repair of wrong or missing guesses: the
    multi-threaded decoder must deliver exactly zlib's bytes (and check CRC-32 / ISIZE) whatever the chunking"""
    import zlib
    d = _gz_inputs()
    big = d["fastq"] * 6 + d["random"] * 3 + d["run"] + d["mixed"] * 4 + d["period 7"]
    for level, strategy in ((1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (0, 0), (6, zlib.Z_HUFFMAN_ONLY)):
        assert host.gunzip_parallel(_gz(big, level, strategy), len(big), threads, chunk, 77_777) == big, (level, strategy)
    members = _gz(d["fastq"] * 3, 6) + _gz(d["mixed"], 1) + _gz(b"", 6) + _gz(d["fastq"], 9)
    assert host.gunzip_parallel(members, len(d["fastq"]) * 4 + len(d["mixed"]), threads, chunk) == d["fastq"] * 3 + d["mixed"] + d["fastq"]
    raw = bytearray(_gz(big, 6))
    raw[len(raw) // 3] ^= 4
    with pytest.raises(RuntimeError):
        host.gunzip_parallel(bytes(raw), len(big) + 100_000, threads, chunk)
    with pytest.raises(RuntimeError):
        host.gunzip_parallel(bytes(_gz(big, 6)[:-20]), len(big) + 100_000, threads, chunk)


@pytest.mark.parametrize("threads", [1, 4])
def test_bgzf_blocks_are_inflated_side_by_side(threads):
    from conftest import bgzf as _bgzf
    """BGZF input (blocks that say how long they are) takes the block-parallel reader; ordinary members behind the blocks,
    trailing garbage, odd buffer sizes and damage must behave exactly as with the general decoder"""
    import gzip as gz
    d = _gz_inputs()
    data = d["fastq"] * 5 + d["random"] + d["mixed"] * 2
    for kw in (dict(), dict(block=4000, level=1), dict(eof_marker=False), dict(extra_subfield=True), dict(block=60000, level=0)):
        raw = _bgzf(data, **kw)
        assert gz.decompress(raw) == data
        for room in (1 << 20, 77_777, 65_536, 10_007):
            assert host.gunzip_parallel(raw, len(data), threads, 1 << 16, room) == data, (kw, room)
    assert host.gunzip_parallel(_bgzf(b""), 10, threads) == b""
    # ordinary members behind the blocks (the general decoder takes over), then trailing garbage
    tail = d["mixed"] * 30
    mixed = _bgzf(data, eof_marker=False) + _gz(tail, 6) + _bgzf(d["fastq"])
    assert host.gunzip_parallel(mixed, len(data) + len(tail) + len(d["fastq"]), threads, 1 << 16, 50_000) == data + tail + d["fastq"]
    assert host.gunzip_parallel(_bgzf(data) + b"\0" * 100, len(data), threads) == data
    # damage: a flipped bit in the text of a block, a wrong CRC, a wrong ISIZE, a truncated last block
    raw = bytearray(_bgzf(data))
    for pos in (len(raw) // 2, 40):
        bad = bytearray(raw)
        bad[pos] ^= 0x10
        with pytest.raises(RuntimeError):
            host.gunzip_parallel(bytes(bad), len(data) + 70_000, threads)
    first_len = int.from_bytes(raw[16:18], "little") + 1
    bad = bytearray(raw)
    bad[first_len - 8] ^= 1       # CRC-32 of the first block
    with pytest.raises(RuntimeError):
        host.gunzip_parallel(bytes(bad), len(data) + 70_000, threads)
    bad = bytearray(raw)
    bad[first_len - 4] ^= 1       # its ISIZE
    with pytest.raises(RuntimeError):
        host.gunzip_parallel(bytes(bad), len(data) + 70_000, threads)
    with pytest.raises(RuntimeError):
        host.gunzip_parallel(bytes(raw[:len(raw) - 40]), len(data) + 70_000, threads)


def test_gunzip_survives_random_damage():
    """random damage to the compressed bytes must end in an error (or, for bytes that do not matter such as the header's
    timestamp, in the exact data) -- never in a crash, a hang or silently different data"""
    import random
    d = _gz_inputs()
    data = d["fastq"][:200_000] + d["random"][:20_000] + d["fastq"][:100_000]
    good = _gz(data, 6)
    rnd = random.Random(99)
    outcomes = {"error": 0, "same": 0}
    for case in range(400):
        raw = bytearray(good)
        for _ in range(rnd.choice([1, 1, 1, 2, 5])):
            pos = rnd.randrange(len(raw))
            raw[pos] ^= 1 << rnd.randrange(8)
        if rnd.random() < 0.1:
            raw = raw[:rnd.randrange(len(raw))]
        for fn in ((lambda b: host.gunzip(b, len(data) + 65536, 10_000)),
                   (lambda b: host.gunzip_parallel(b, len(data) + 65536, 3, 65536, 50_000))):
            try:
                out = fn(bytes(raw))
            except RuntimeError:
                outcomes["error"] += 1
                continue
            assert out == data, case
            outcomes["same"] += 1
    assert outcomes["error"] > 600


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_gunzip_under_sanitizers(tmp_path, sanitizer):
    """the multi-threaded decoder under ThreadSanitizer and AddressSanitizer (CPU build; GPU sanitizers are not available)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(__file__), "native", "gunzip_sanitize.cpp")
    exe = str(tmp_path / "gunzip_sanitize")
    flags = ["-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize=alignment", "-std=c++17", "-pthread"]
    b = subprocess.run(["g++", *flags, "-o", exe, src, "-lz"], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fails 0" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_writers_under_sanitizers(tmp_path, sanitizer):
    """the output side of the host layer (formatting pool + writer threads) under ThreadSanitizer / AddressSanitizer"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(__file__), "native", "writers_sanitize.cpp")
    exe = str(tmp_path / "writers_sanitize")
    flags = ["-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize=alignment", "-std=c++17", "-pthread"]
    b = subprocess.run(["g++", *flags, "-o", exe, src, "-lz"], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fails 0" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_gunzipper_upload_threads_under_sanitizers(tmp_path, sanitizer):
    """the host threads of the DEVICE gunzip (gs_gunzipper_*: the staged copy and the upload thread, genestrip_amd/csrc/gs_upload.h)
    against a mock device under ThreadSanitizer / AddressSanitizer: batches following the upload, park() with the source freed,
    reopen, a failing copy helper, an uploader destroyed while it runs (ADVICE r03 high was a use-after-unmap
    of exactly this thread)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(os.path.dirname(__file__), "native", "uploader_sanitize.cpp")
    exe = str(tmp_path / "uploader_sanitize")
    flags = ["-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize=alignment", "-std=c++17", "-pthread"]
    b = subprocess.run(["g++", *flags, "-o", exe, src, "-lz"], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fails 0" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


# ------------------------------------------------------------------ f4: the report writer against rows the reference prints
def test_csv_writer_reproduces_the_readme_rows(tmp_path):
    """README.md:168-181 prints the header and 13 rows of `human_virus_match_sample.csv` (fixture
    tests/golden/readme_csv/).  Rows 1-12 form one root-to-leaf chain of the tax tree.  Feeding their own integer columns
    (and the double sums behind the four error columns) to gs_host_write_csv must give back: the header byte for byte,
    every integer column, every double column that derives from integers as the SAME string (formula order +
    Double.toString), and -- for rows 5-12, whose only descendant with counts is the leaf -- the accumulated columns.
    (acc. columns of rows 1-4 include descendants the README does not print.)"""
    import math
    lines = open(os.path.join(GOLDEN, "readme_csv", "human_virus_match_sample_rows.csv")).read().split("\n")
    head = lines[0].split(";")
    col = {n: i for i, n in enumerate(head)}
    want = [l.split(";") for l in lines[1:] if l]
    assert len(want) == 13 and all(len(r) == len(head) for r in want)
    total, rows = want[0], want[1:]
    nv = len(rows)
    parent = [-1] + list(range(nv - 1))
    assert [r[col["parent taxid"]] for r in rows] == [""] + [r[col["taxid"]] for r in rows[:-1]]  # it is a chain
    t = np.zeros((nv, 10), dtype=np.int64)
    d = np.zeros((nv, 4))
    desc = [None] * nv
    icols = ["reads", "kmers from reads", "kmers", "unique kmers", "contigs", None, "max contig length", "reads >=1 kmer", "reads bps"]
    for v, r in enumerate(rows):
        for j, name in enumerate(icols):
            if name:
                t[v, j] = int(r[col[name]])
        t[v, 9] = -1
        contigs, kmers, reads = int(r[col["contigs"]]), int(r[col["kmers"]]), int(r[col["reads"]])
        if r[col["contig len std. dev."]]:  # contigLenSquaredSum is an integer: invert getContigLenStdDev (:475-477)
            sd = float(r[col["contig len std. dev."]])
            t[v, 5] = round(sd * sd * (contigs - 1) + kmers * kmers / contigs)
        if reads:
            for j, (mean, sd) in enumerate([("mean error", "kmer error std. dev."), ("mean class error", "class error std. dev.")]):
                s = float(r[col[mean]]) * reads
                d[v, 2 * j] = s
                d[v, 2 * j + 1] = float(r[col[sd]]) ** 2 * (reads - 1) + s * s / reads
        if r[col["max contig desc."]]:
            desc[v] = r[col["max contig desc."]]
            t[v, 9] = v  # any read number: the caller maps it to the descriptor
    tot = host.Totals(int(total[col["reads"]]), int(total[col["kmers"]]), int(total[col["reads bps"]]), 0, 0, 0, 0)
    out = tmp_path / "readme.csv"
    host.write_csv(out, parent, [r[col["taxid"]] for r in rows], [int(r[col["db kmers"]]) for r in rows],
                   int(total[col["db kmers"]]), t, d, tot, names=[r[col["name"]] for r in rows],
                   ranks=[r[col["rank"]] for r in rows], max_contig_desc=desc)
    got_lines = out.read_text().split("\n")
    # all 46 header columns, separators included.  One name differs between the README (printed by an older release) and
    # the source the writer follows: README.md:168 "acc. class mean error", C/match/CountsPerTaxid.java:563
    # @MDCDescription(pos = 1003, name = "acc. mean class error") -- the source wins.
    assert lines[0].count("acc. class mean error;") == 1
    assert got_lines[0] == lines[0].replace("acc. class mean error;", "acc. mean class error;")
    col["acc. mean class error"] = col["acc. class mean error"]
    head = [("acc. mean class error" if n == "acc. class mean error" else n) for n in head]
    got = [l.split(";") for l in got_lines[1:] if l]
    assert got[0] == total  # the TOTAL row, every column
    assert len(got) == 13
    exact = ["pos", "level", "name", "rank", "taxid", "reads", "kmers from reads", "kmers", "unique kmers", "contigs",
             "average contig length", "max contig length", "reads >=1 kmer", "reads bps", "avg. read length", "db coverage",
             "exp. unique kmers", "unique kmers / exp.", "db kmers", "parent taxid", "norm. reads", "norm. kmers",
             "norm. reads bps", "norm. read >=1 kmer", "norm. reads kmers", "max contig desc.", "contig len std. dev."]
    acc = [n for n in head if n.startswith("acc. ") and "error" not in n]
    approx = ["mean error", "kmer error std. dev.", "mean class error", "class error std. dev."]
    acc_approx = [n for n in head if n.startswith("acc. ") and "error" in n]
    for g, w in zip(got[1:], rows):
        for n in exact:
            assert g[col[n]] == w[col[n]], (w[col["name"]], n, g[col[n]], w[col[n]])
        for n in approx:
            assert (g[col[n]] == "") == (w[col[n]] == "")
            if w[col[n]]:
                assert math.isclose(float(g[col[n]]), float(w[col[n]]), rel_tol=1e-9), (w[col["name"]], n)
        if int(w[col["pos"]]) >= 5:  # below Orthornavirae the printed rows hold the whole subtree
            for n in acc:
                assert g[col[n]] == w[col[n]], (w[col["name"]], n, g[col[n]], w[col[n]])
            for n in acc_approx:
                assert math.isclose(float(g[col[n]]), float(w[col[n]]), rel_tol=1e-9), (w[col["name"]], n)


def test_csv_writer_reports_a_failed_write():
    """a short write must not pass for a result (/dev/full accepts the open and fails the flush)"""
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    t = np.zeros((1, 10), dtype=np.int64)
    t[0] = [1, 1, 1, 1, 1, 1, 1, 1, 1, 0]
    from genestrip_amd import binding
    with pytest.raises(binding.GsError) as e:
        host.write_csv("/dev/full", [-1], ["1"], [5], 5, t, np.zeros((1, 4)), host.Totals(1, 1, 1, 0, 0, 0, 0))
    assert e.value.code == -7
