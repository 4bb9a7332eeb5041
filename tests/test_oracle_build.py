"""The DB-construction restatement (oracle/gs_oracle.c: orc_build_*, FillDBGoal + DBGoal) against what the reference's own
tests and fixtures pin: the LCA known answers of T/tax/TaxTreeLCATest.java, the dengue1 genome whose k-mer set the Kraken
golden line (R/projects/dengue1/test.out) depends on, and a dictionary implementation written from the Java sources'
description in plain Python.  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import gs_oracle as orc

# T/tax/TaxTreeLCATest.java:51 edges {child, parent}: {1,1},{2,1},{3,2},{4,2},{5,3},{6,5},{7,1}; value index = taxid - 1
PARENT = np.array([-1, 0, 1, 1, 2, 4, 0], dtype=np.int32)


def test_taxtree_lca_known_answers():
    def lca(a, b):
        return orc.taxtree_lca(PARENT, a - 1 if a else -1, b - 1 if b else -1) + 1

    assert lca(6, 6) == 6                                    # equal nodes
    assert lca(5, 6) == 5 and lca(6, 5) == 5 and lca(1, 6) == 1  # ancestor / descendant
    assert lca(6, 4) == 2 and lca(4, 6) == 2 and lca(3, 4) == 2  # diverging within a subtree
    assert lca(6, 7) == 1                                    # different top-level branches
    assert lca(None, 6) == 0 and lca(6, None) == 0           # null -> null
    # the brute-force cross check of the reference's test, over all ordered pairs
    def anc(x):
        out = []
        while x >= 0:
            out.append(x)
            x = PARENT[x]
        return out
    for a in range(7):
        for b in range(7):
            want = next(x for x in anc(a) if x in anc(b))
            assert orc.taxtree_lca(PARENT, a, b) == want


FIB = [0, 1, 2]
for _i in range(3, 40):
    FIB.append(FIB[-1] + FIB[-2])


def _naive_dust(w):
    """the window form of the low-complexity score as T/util/CGATLongBufferTest.java:280-313 states it"""
    d, srl, last = 0, [0, 0, 0], [None, None, None]
    for ch in w:
        for p in range(3):
            if ch == last[p]:
                srl[p] += 1
            else:
                d += FIB[srl[p]]
                srl[p] = 0
        last = [ch, last[0], last[1]]
    return d + FIB[srl[0]] + FIB[srl[1]] + FIB[srl[2]]


def test_dust_known_answers_and_the_window_form():
    """T/util/CGATLongBufferTest.java:56-108 (known answers) and :119-145 (the streaming score equals the window form on every
    window of a random sequence, also right after a reset by a non-base)"""
    assert orc.dust_value("TTTCGCGA") == FIB[2] + FIB[1] + FIB[2]
    assert [orc.dust_value(x) for x in ("ACAT", "AAAT", "AAAA", "ACAA")] == [FIB[1], FIB[2] + FIB[1], FIB[3] + FIB[2] + FIB[1], 3 * FIB[1]]
    assert [orc.dust_value(x) for x in ("ACATA", "AAAAA", "AATAA", "TATAT")] == [2 * FIB[1], FIB[4] + FIB[3] + FIB[2], 3 * FIB[1] + FIB[2], FIB[3]]
    assert orc.dust_value("AC" * 15 + "A") == FIB[29] and orc.dust_value("GC" * 15 + "G") == FIB[29]
    assert orc.dust_value("C" * 31) == FIB[30] + FIB[29] + FIB[28] == orc.dust_value("T" * 31)
    rng = np.random.default_rng(10)
    for k in (4, 9, 31):
        # sequences with long runs and short periods, cut by N: every window's k-mer passes the filter iff its window score does
        parts = []
        for _ in range(30):
            parts.append(rng.choice(list("ACGT"), int(rng.integers(1, 60))))
            unit = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 4))))
            parts.append(list((unit * 40)[:int(rng.integers(0, 50))]))
            if rng.random() < 0.3:
                parts.append(["N"])
        s = "".join("".join(p) for p in parts)
        for max_dust in (0, 3, 12, 100):
            want = sum(1 for i in range(len(s) - k + 1) if "N" not in s[i:i + k] and _naive_dust(s[i:i + k]) <= max_dust)
            got = orc.lib().orc_dust_passed(k, max_dust, orc._p(orc._seq(s)), len(s))
            assert got == want, (k, max_dust)


def _py_build(k, regions, parent, lower=True, step=1, max_dust=-1):
    """dictionary form of FillDBGoal + DBGoal: regions = [(bytes, node, is_update)] in order"""
    comp = {"C": "G", "G": "C", "A": "T", "T": "A"}
    code = {"C": 0, "G": 1, "A": 2, "T": 3}

    def enc(s):
        v = 0
        for ch in s:
            v = (v << 2) | code[ch]
        return v

    def lca(a, b):
        pa = []
        while a >= 0:
            pa.append(a)
            a = parent[a]
        while b >= 0:
            if b in pa:
                return b
            b = parent[b]
        return -1

    def kmers(seq):
        s = seq.decode("latin1")
        if lower:
            s = "".join({"a": "A", "c": "C", "g": "G", "t": "T"}.get(ch, ch) for ch in s)
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if (i + k) % step == 0 and all(ch in code for ch in w) and (max_dust < 0 or _naive_dust(w) <= max_dust):
                yield max(enc(w), enc("".join(comp[ch] for ch in reversed(w))))

    store = {}
    for seq, node, upd in regions:
        if not upd:
            for x in kmers(seq):
                store.setdefault(x, node)
    for seq, node, upd in regions:
        if upd:
            for x in kmers(seq):
                if x in store:
                    l = lca(store[x], node)
                    if l >= 0:
                        store[x] = l
    keys = np.array(sorted(store), dtype=np.int64)
    return keys, np.array([store[x] for x in keys], dtype=np.int32)


def _orc_build(k, regions, parent, lower=True, step=1, max_dust=-1):
    b = orc.DbBuild(k, len(parent), parent, lower, step, max_dust)
    for upd in (False, True):
        part = [(s, n) for s, n, u in regions if u == upd]
        if upd:
            b.optimize()
        if part:
            seq = np.frombuffer(b"".join(s for s, _ in part), dtype=np.uint8)
            off = np.cumsum([0] + [len(s) for s, _ in part]).astype(np.uint64)
            (b.update if upd else b.fill)(seq, off, np.array([n for _, n in part], dtype=np.int32))
    out = b.fetch()
    b.close()
    return out


@pytest.mark.parametrize("k,lower,step,max_dust", [(5, True, 1, -1), (5, False, 1, -1), (7, True, 3, -1), (31, True, 1, -1), (2, True, 1, -1),
                                                   (9, True, 1, 6), (31, True, 1, 20), (7, True, 2, 5)])
def test_build_restatement_equals_the_dictionary_form(k, lower, step, max_dust):
    rng = np.random.default_rng(k * 31 + step)
    alphabet = np.frombuffer(b"ACGTacgtN\r", dtype=np.uint8)
    p = np.array([0.23, 0.23, 0.23, 0.23, 0.02, 0.02, 0.01, 0.01, 0.015, 0.005])
    core = rng.choice(alphabet[:4], 400).tobytes()
    regions = []
    for r in range(14):
        body = bytearray(rng.choice(alphabet, int(rng.integers(0, 300)), p=p).tobytes())
        if r % 3 == 0 and len(body) > 120:  # shared material: the same k-mers under different nodes
            a = int(rng.integers(0, 250))
            body[20:120] = core[a:a + 100]
        if r % 4 == 1 and len(body) > 200:  # low complexity
            body[130:190] = (b"AT" * 30) if r % 8 == 1 else (b"A" * 60)
        regions.append((bytes(body), int(rng.integers(0, 7)), False))
    regions += [(s, n, True) for s, n, _ in regions]
    regions += [(core[50:350], 6, True), (b"", 3, True), (b"ACG", 2, False)]
    wk, wv = _py_build(k, regions, PARENT, lower, step, max_dust)
    gk, gv = _orc_build(k, regions, PARENT, lower, step, max_dust)
    assert np.array_equal(wk, gk) and np.array_equal(wv, gv)
    assert len(wk) > 50 or k == 2
    if max_dust >= 0:  # the filter did take something out
        assert len(wk) < len(_py_build(k, regions, PARENT, lower, step, -1)[0])


def test_first_region_wins_without_an_update_pass_and_update_only_touches_stored_kmers():
    tree = np.array([-1, 0, 0], dtype=np.int32)
    a, b = b"ACGTACGGTTCA", b"TTACGTACGGAA"  # share ACGTACGG and what lies inside
    k1, v1 = _orc_build(5, [(a, 1, False), (b, 2, False)], tree)
    k2, v2 = _orc_build(5, [(b, 2, False), (a, 1, False)], tree)
    assert np.array_equal(k1, k2) and not np.array_equal(v1, v2)  # putLong: the first writer's node stays
    shared = set(orc.canonical_kmers(a.decode(), 5).tolist()) & set(orc.canonical_kmers(b.decode(), 5).tolist())
    assert len(shared) >= 3
    assert all(v1[list(k1).index(x)] == 1 for x in shared) and all(v2[list(k2).index(x)] == 2 for x in shared)
    # DBGoal: b as an update-only region moves the shared k-mers of a's store to the root and adds nothing
    k3, v3 = _orc_build(5, [(a, 1, False), (a, 1, True), (b, 2, True)], tree)
    assert np.array_equal(k3, np.unique(orc.canonical_kmers(a.decode(), 5)))
    assert all((v3[i] == 0) == (int(k3[i]) in shared) for i in range(len(k3)))


def test_dengue1_genome_gives_the_kmer_set_behind_the_kraken_golden_line():
    raw = open(os.path.join(GOLDEN, "dengue1", "dengue1.fasta"), "rb").read()
    rd = orc.parse_fastq(raw, fasta=True, k=31)
    assert rd["n_reads"] == 1
    seq, off = rd["seq"], rd["seq_off"]
    b = orc.DbBuild(31, 1, np.array([-1], np.int32))
    b.fill(seq, off, np.array([0], np.int32))
    n = b.optimize()
    keys, vals = b.fetch()
    b.close()
    genome = bytes(seq).decode().upper()
    want = np.unique(orc.canonical_kmers(genome, 31))
    assert n == len(want) and np.array_equal(keys, want) and not vals.any()
    assert len(genome) == 10735  # (tests/test_gpu_host.py: the same store reproduces R/projects/dengue1/test.out byte for byte)


def test_build_restatement_under_sanitizers(tmp_path):
    """orc_build_* compiled with AddressSanitizer + UBSan (CPU only: GPU sanitizers are not available on this pool)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_build_sanitize")
    b = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=alignment", "-fopenmp", "-o", exe,
                        os.path.join(root, "tests", "native", "oracle_build_sanitize.c"), os.path.join(root, "oracle", "gs_oracle.c"), "-lm"],
                       capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "k=31 stored=" in r.stdout and "ERROR" not in r.stderr
