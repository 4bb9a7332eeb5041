"""Pins the CPU oracle (oracle/) against the reference's own known-answer tests and fixtures
(SURVEY.md section 8c, K1..K10).  CPU only."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import gs_oracle as orc


def _tree3():
    """R/taxtree/{nodes,names}.dmp: root 1 with children 2 and 3 (value indices 0,1,2 pre-order)."""
    parent = {}
    with open(os.path.join(GOLDEN, "taxtree", "nodes.dmp")) as f:
        for line in f:
            p = [x.strip() for x in line.split("|")]
            parent[p[0]] = p[1]
    assert parent == {"1": "1", "2": "1", "3": "1"}
    taxids = ["1", "2", "3"]
    return taxids, np.array([-1, 0, 0], dtype=np.int32)


def _db_k2(pairs, parent_vi=None, bloom_gate=False):
    """k=2 store; pairs = [(2-mer string, value index)] inserted via the canonical k-mer
    (T/match/FastqKMerMatcherTest.java:101-110)."""
    d = {}
    for s, vi in pairs:
        d.setdefault(orc.kmer_canonical(s), vi)  # putLong ignores duplicates
    keys = sorted(d)
    return orc.DB(2, keys, [d[x] for x in keys], 3, parent_vi, bloom_gate)


# ---------------------------------------------------------------- K10
def test_k10_java_random_constants():
    r = orc.JRandom(42)
    assert [r.next_long() for _ in range(3)] == [-5025562857975149833, -5843495416241995736, 5694868678511409995]
    # java.util.Random(42).nextInt(4) is a power-of-two bound
    r = orc.JRandom(42)
    v = [r.next_int(4) for _ in range(1000)]
    assert min(v) == 0 and max(v) == 3


# ---------------------------------------------------------------- K1
def test_k1_k2_codec():
    assert orc.kmer_canonical("CC") == 5 and orc.kmer_canonical("GG") == 5
    assert orc.kmer_canonical("TT") == 15 and orc.kmer_canonical("AA") == 15
    assert orc.kmer_canonical("AG") == 9 and orc.kmer_canonical("CT") == 9
    assert orc.kmer_canonical("CG") == 1
    assert orc.kmer_canonical("GA") == 12
    assert orc.kmer_straight("CNGT", 0, 4) == (-1, 1)
    assert orc.kmer_reverse("CNGN", 0, 4) == (-1, 3)  # scans right to left
    assert orc.kmer_straight("acgt", 0, 4)[0] == -1  # lower case is invalid


# ---------------------------------------------------------------- K9
@pytest.mark.parametrize("k", [1, 2, 5, 15, 16, 30, 31])
def test_k9_rolling_equals_from_scratch(k):
    L = orc.lib()
    rnd = orc.JRandom(10)
    seq = bytes(b"CGAT"[rnd.next_int(4)] for _ in range(3000))
    a = np.frombuffer(seq, dtype=np.uint8)
    ks, _ = orc.kmer_straight(seq, 0, k)
    kr, _ = orc.kmer_reverse(seq, 0, k)
    for i in range(1, len(seq) - k + 1):
        ks = L.orc_next_straight(ks, int(a[i + k - 1]), k)
        kr = L.orc_next_reverse(kr, int(a[i + k - 1]), k)
        assert ks == orc.kmer_straight(seq, i, k)[0]
        assert kr == orc.kmer_reverse(seq, i, k)[0]
        if i > 200:
            break
    # vectorised helper agrees with the scalar codec
    can = orc.canonical_kmers(seq[:400], k)
    ref = [orc.kmer_canonical(seq, i, k) for i in range(400 - k + 1)]
    assert can.tolist() == ref


# ---------------------------------------------------------------- K8
def test_k8_lca_table():
    # tree of T/tax/SmallTaxTreeLCATest.java:50-56; value index = id-1
    edges = {1: 1, 2: 1, 3: 2, 4: 2, 5: 3, 6: 5, 7: 1}
    parent = np.array([-1 if c == p else p - 1 for c, p in sorted(edges.items())], dtype=np.int32)
    db = orc.DB(2, [5], [0], 7, parent)
    n = lambda s: s - 1
    assert db.lca(n(5), n(6)) == n(5) and db.lca(n(6), n(5)) == n(5)
    assert db.lca(n(1), n(6)) == n(1)
    assert db.lca(n(6), n(4)) == n(2) and db.lca(n(3), n(4)) == n(2)
    assert db.lca(n(6), n(7)) == n(1)
    assert db.lca(n(6), n(6)) == n(6)
    assert db.lca(-1, n(6)) == -1 and db.lca(n(6), -1) == -1

    def anc(x):
        out = [x]
        while parent[x] >= 0:
            x = parent[x]
            out.append(x)
        return out
    for x in range(7):
        for y in range(7):
            brute = next(a for a in anc(x) if a in anc(y))
            assert db.lca(x, y) == brute


# ---------------------------------------------------------------- K2
K2_CASES = [
    (0.0, [("CCCC", "1"), ("GAGAGA", None), ("CCCG", "3"), ("AGGGG", "2"), ("CCCCCCT", "2")]),
    (1.0, [("CTCCT", "2"), ("CTCTCCT", None), ("TAGGGG", "2"), ("TAGGGGT", None)]),
    (0.5, [("CCA", "1"), ("CCAA", None)]),
    (0.1, [("CC", "1"), ("CCA", None), ("CCAA", None)]),
    (0.99, [("TTTT", None), ("CTTT", "2")]),
]


@pytest.mark.parametrize("gate", [False, True])
def test_k2_read_classification(gate):
    taxids, parent = _tree3()
    db = _db_k2([("CC", 0), ("CT", 1), ("CG", 2)], parent, bloom_gate=gate)
    read_no = 0
    for err, cases in K2_CASES:
        run = orc.MatchRun(db, classify=True, max_paths=4, threshold=1, max_read_tax_err=err,
                           max_read_class_err=-1.0)
        for read, want in cases:
            read_no += 1
            cv, fl = run.submit_reads([read], first_read_no=read_no)
            got = None if cv[0] < 0 else taxids[cv[0]]
            assert got == want, (err, read, got, want)


# ---------------------------------------------------------------- K3
def _k3_recount(read):
    """the in-test recount of T/match/FastqKMerMatcherTest.java:137-181"""
    counters, contigs, maxlen = [0] * 3, [0] * 3, [0] * 3
    t, contig_len = -1, 0
    for j in range(len(read)):
        if j > 0:
            last_t = t
            pair = read[j - 1:j + 1]
            if pair in (b"CC", b"GG"):
                t = 0
            elif pair in (b"AA", b"TT"):
                t = 1
            elif pair in (b"AG", b"CT"):
                t = 2
            else:
                t = -1
            if t >= 0:
                counters[t] += 1
            if last_t != t and last_t != -1:
                contigs[last_t] += 1
                maxlen[last_t] = max(maxlen[last_t], contig_len)
                contig_len = 0
        if t != -1:
            contig_len += 1
    if t != -1:
        contigs[t] += 1
        maxlen[t] = max(maxlen[t], contig_len)
    return counters, contigs, maxlen


def test_k3_match_read_recount():
    db = _db_k2([("CC", 0), ("GG", 1), ("TT", 1), ("AG", 2)])
    rnd = orc.JRandom(42)
    for i in range(2000):  # all 2000 reads of T/match/FastqKMerMatcherTest.java:98
        read = bytes(b"CGAT"[rnd.next_int(4)] for _ in range(500))
        counters, contigs, maxlen = _k3_recount(read)
        # MyFastqMatcher: taxTree == null, maxReadTaxErrorCount 0, maxReadClassErrorCount 0
        run = orc.MatchRun(db, classify=False, max_paths=4, max_read_tax_err=0.0, max_read_class_err=0.0)
        run.submit_reads([read], first_read_no=0)
        t, _ = run.finish()
        for j in range(3):
            if counters[j] == 0:
                assert t[j, orc.C_READS_1KMER] == 0
            else:
                assert t[j, orc.C_KMERS] == counters[j]
                assert t[j, orc.C_UNIQUE_KMERS] == 1
                assert t[j, orc.C_CONTIGS] == contigs[j]
                assert t[j, orc.C_MAX_CONTIG_LEN] == maxlen[j]


# ---------------------------------------------------------------- K4
def test_k4_thread_invariance():
    db = _db_k2([("CC", 0), ("GG", 1), ("TT", 1), ("AG", 2)])
    rnd = orc.JRandom(4242)
    reads = [bytes(b"CGAT"[rnd.next_int(4)] for _ in range(200)) for _ in range(4000)]
    res = []
    for threads in (1, 8):
        run = orc.MatchRun(db, classify=False, max_paths=4, max_read_tax_err=0.0, max_read_class_err=0.0)
        run.submit_reads(reads, first_read_no=1, threads=threads)
        res.append(run.finish()[0])
    assert np.array_equal(res[0], res[1])
    assert np.all(res[0][:, orc.C_KMERS] > 0)


# ---------------------------------------------------------------- K5
def _dengue_db():
    lines = open(os.path.join(GOLDEN, "dengue1", "dengue1.fasta")).read().split("\n")
    genome = "".join(l.strip() for l in lines if not l.startswith(">")).upper()
    assert len(genome) == 10735
    keys = np.unique(orc.canonical_kmers(genome, 31))
    return genome, keys


def test_k5_dengue_kraken_line():
    genome, keys = _dengue_db()
    assert len(keys) == 10705
    db = orc.DB(31, keys, np.zeros(len(keys), np.int32), 1, np.array([-1], np.int32))
    rd = orc.parse_fastq(open(os.path.join(GOLDEN, "dengue1", "test.fastq"), "rb").read(), k=31)
    assert rd["n_reads"] == 1
    read = rd["seq"].tobytes()
    desc = rd["desc"].tobytes()
    segs = db.segments(read)
    run = orc.MatchRun(db, classify=True)
    cv, fl = run.submit(rd["seq"], rd["seq_off"])
    names = {0: "1", -1: "0", -2: "A"}
    line = ("C" if cv[0] >= 0 else "U") + "\t" + desc[1:].split(b" ")[0].decode() + "\t" + \
        ("1" if cv[0] == 0 else "0") + "\t" + str(len(read)) + "\t" + " ".join(f"{names[c]}:{n}" for c, n in segs)
    golden = open(os.path.join(GOLDEN, "dengue1", "test.out")).read().rstrip("\n")
    assert line == golden == "C\ttest\t1\t41\t0:2 1:7 0:2"


# ---------------------------------------------------------------- K6
def test_k6_sample_fastq_totals():
    data = gzip.open(os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")).read()
    rd = orc.parse_fastq(data, k=31)
    assert (rd["n_reads"], rd["total_bps"], rd["total_kmers"]) == (6565, 658255, 461305)


# ---------------------------------------------------------------- K7
def test_k7_parser_fixture():
    data = open(os.path.join(GOLDEN, "fastq", "SimpleTest.fastq"), "rb").read()
    rd = orc.parse_fastq(data, k=2)
    assert rd["n_reads"] == 2

    def field(name, i):
        o = rd[name + "_off"]
        return rd[name][int(o[i]):int(o[i + 1])].tobytes().decode()
    assert field("desc", 0) == "@S"
    assert field("seq", 0) == "GATTTGGGGTTCAAAGCAGTATCGATCAAATAGTAAATCCATTTGTTCAACTCACAGTTT"
    assert field("qual", 0) == "!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65"
    assert (field("desc", 1), field("seq", 1), field("qual", 1)) == ("@T", "CGAT", "!**>")


def test_parser_edge_cases():
    # no trailing newline => the last line loses its final byte; \r is kept; NUL bytes are dropped
    rd = orc.parse_fastq(b"@r1 x\nAC\0GT\r\n+\n!!!!!\n@r2\nACGT\n+\nIIII", k=2)
    assert rd["n_reads"] == 2
    assert rd["seq"][:5].tobytes() == b"ACGT\r"
    # "IIII" at EOF: nextLine()-1 = 3 < readSize 4, the retry at EOF returns startPos => 3-1 = 2
    # (AbstractFastqReader.java:329-337)
    assert int(rd["qual_off"][2] - rd["qual_off"][1]) == 2
    fa = orc.parse_fastq(b">s1 d\nACGT\nAC\n>s2\nGG\n", fasta=True, k=2)
    assert fa["n_reads"] == 2
    assert fa["seq"].tobytes() == b"ACGTACGG" and fa["desc"].tobytes() == b"@s1 d@s2"


# ---------------------------------------------------------------- Bloom filters (section 8a rows a5, a10, a11)
def test_bloom_sizing_and_membership():
    # p = 1e-8 => 38.34 bits/key and 27 hashes
    b = orc.Bloom(orc.BLOOM_XOR, 1_000_000, 1e-8)
    assert b.hashes == 27 and 38_300_000 < b.bits < 38_400_000
    assert b.hash_factors[:3].tolist() == [-5025562857975149833, -5843495416241995736, 5694868678511409995]
    for kind, fpp in ((orc.BLOOM_XOR, 1e-4), (orc.BLOOM_MURMUR, 1e-4), (orc.BLOOM_BLOCKED, 0.01)):
        f = orc.Bloom(kind, 5000, fpp)
        rnd = np.random.default_rng(1)
        keys = rnd.integers(0, 2 ** 62, 5000, dtype=np.int64)
        f.put_many(keys)
        assert all(f.contains(x) for x in keys.tolist())
        other = rnd.integers(0, 2 ** 62, 20000, dtype=np.int64)
        fp = sum(f.contains(x) for x in other.tolist()) / 20000
        assert fp <= (0.03 if kind == orc.BLOOM_BLOCKED else 1.1e-3)
    blk = orc.Bloom(orc.BLOOM_BLOCKED, 1000, 0.01)
    assert blk.hash_factors[0] == -5025562857975149833 and blk.bits == (1000 * 10 + 63) // 64


def test_filter_accept_read_thresholds():
    genome, keys = _dengue_db()
    f = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    f.put_many(keys)
    read = genome[100:250]
    assert f.accept_read(31, 1, 0.2, read)
    assert f.accept_read(31, 0, 0.2, read)
    junk = "ACGT" * 40
    assert not f.accept_read(31, 1, 0.2, junk)
    # one hit k-mer only: accepted with minPosCount=1, rejected by ratio 0.2 (24 of 120 needed)
    one = genome[500:531] + junk[:119]
    assert f.accept_read(31, 1, 0.2, one)
    assert not f.accept_read(31, 0, 0.2, one)
    # an N inside the only matching window kills it
    assert not f.accept_read(31, 1, 0.2, one[:10] + "N" + one[11:])
    assert not f.accept_read(31, 1, 0.2, "ACGT")  # shorter than k
