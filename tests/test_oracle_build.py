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


def _py_build(k, regions, parent, lower=True, step=1):
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
            if (i + k) % step == 0 and all(ch in code for ch in w):
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


def _orc_build(k, regions, parent, lower=True, step=1):
    b = orc.DbBuild(k, len(parent), parent, lower, step)
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


@pytest.mark.parametrize("k,lower,step", [(5, True, 1), (5, False, 1), (7, True, 3), (31, True, 1), (2, True, 1)])
def test_build_restatement_equals_the_dictionary_form(k, lower, step):
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
        regions.append((bytes(body), int(rng.integers(0, 7)), False))
    regions += [(s, n, True) for s, n, _ in regions]
    regions += [(core[50:350], 6, True), (b"", 3, True), (b"ACG", 2, False)]
    wk, wv = _py_build(k, regions, PARENT, lower, step)
    gk, gv = _orc_build(k, regions, PARENT, lower, step)
    assert np.array_equal(wk, gk) and np.array_equal(wv, gv)
    assert len(wk) > 50 or k == 2


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
