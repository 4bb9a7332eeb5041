"""Parity of the HIP filter path (gs_filter_submit) against the oracle's FastqBloomFilter.isAcceptRead.
The Bloom bit arrays and hash factors are replicated exactly, so false positives must agree too."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _filters(keys, kind, fpp):
    ob = orc.Bloom(kind, len(keys), fpp)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(kind, ob.bits, ob.hash_factors, ob.words)
    return ob, gb


def _reads(sdb, n=12000):
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=123)
    rng = np.random.default_rng(3)
    seq = seq.copy()
    for r in rng.choice(n, n // 50, replace=False):  # sprinkle N's and lower case (invalid bases)
        p = int(off[r]) + int(rng.integers(0, 150))
        seq[p] = ord("N") if rng.random() < 0.7 else ord("a")
    return seq, off


@pytest.mark.parametrize("kind,fpp", [(ga.BLOOM_XOR, 1e-8), (ga.BLOOM_XOR, 0.05), (ga.BLOOM_MURMUR, 1e-4),
                                      (ga.BLOOM_BLOCKED, 0.01)])
@pytest.mark.parametrize("min_pos,ratio", [(1, 0.2), (0, 0.2), (0, 0.9), (5, 0.2), (200, 0.2)])
def test_filter_matches_oracle(sdb, kind, fpp, min_pos, ratio):
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:5])]  # "requested" taxa only (BloomIndexGoal.java:70-110)
    ob, gb = _filters(keys, kind, fpp)
    seq, off = _reads(sdb)
    want = ob.filter_batch(31, min_pos, ratio, seq, off, threads=4)
    got = ga.FastqBloomFilter(31, gb, min_pos, ratio).accept_reads(seq, off)
    assert np.array_equal(want, got), np.flatnonzero(want != got)[:10]
    if min_pos == 1:
        assert 0 < got.sum() < len(got)
    gb.close()


def test_filter_ragged_and_k_variants(sdb):
    rng = np.random.default_rng(9)
    g0 = sdb.genomes
    reads = [b"", b"ACGT", bytes(g0[0][:30]), bytes(g0[0][:31])]
    for _ in range(600):
        s = int(rng.integers(0, g0.shape[0]))
        L = int(rng.integers(10, 900))
        p = int(rng.integers(0, g0.shape[1] - L))
        r = bytearray(g0[s][p:p + L].tobytes())
        if rng.random() < 0.5:
            r[int(rng.integers(0, L))] = ord("N")
        reads.append(bytes(r))
    seq, off = orc.pack_reads(reads)
    for k in (15, 21, 31):
        keys = np.unique(np.concatenate([orc.canonical_kmers(bytes(g0[i]), k) for i in range(3)]))
        ob, gb = _filters(keys, ga.BLOOM_XOR, 1e-6)
        for min_pos, ratio in ((1, 0.2), (0, 0.5)):
            want = ob.filter_batch(k, min_pos, ratio, seq, off)
            got = ga.FastqBloomFilter(k, gb, min_pos, ratio).accept_reads(seq, off)
            assert np.array_equal(want, got), (k, min_pos, np.flatnonzero(want != got)[:10])
        gb.close()


def test_filter_consistent_with_match(sdb):
    """ComprehensiveFilterTest's property (T/goals/refseq/ComprehensiveFilterTest.java:81-168): with
    minPosCountFilter = 1 the filter accepts every read that `match` finds a requested-taxon k-mer in."""
    req = sdb.species_vi
    keys = sdb.kmers[np.isin(sdb.value_idx, req)]
    ob, gb = _filters(keys, ga.BLOOM_XOR, 1e-8)
    seq, off = synth.reads_host(sdb.genomes, 8000, read_len=150, seed=77)
    acc = ga.FastqBloomFilter(31, gb, 1, 0.2).accept_reads(seq, off)
    # match against a store restricted to the requested taxa
    store = ga.DeviceKMerStore(31, keys, sdb.value_idx[np.isin(sdb.value_idx, req)], sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    _, fl = m.match_reads(seq, off)
    found = (fl & ga.F_FOUND) != 0
    assert np.all(acc[found] == 1)           # no false negatives
    assert (acc[~found] == 1).mean() < 1e-3  # false positives only at the Bloom rate
    m.close()
    store.close()
    gb.close()


# ------------------------------------------------------------------ text mode (records found on the device)
def _fastq(seq, off, nl=b"\n"):
    recs = []
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])].tobytes()
        if i % 9 == 4:
            s = s[:i % 37]  # short and empty reads
        recs.append(b"@q%d" % i + nl + s + nl + b"+" + nl + b"#" * len(s) + nl)
    return recs


@pytest.mark.parametrize("kind,fpp,min_pos,ratio", [(ga.BLOOM_XOR, 1e-8, 1, 0.2), (ga.BLOOM_XOR, 0.05, 0, 0.5),
                                                   (ga.BLOOM_MURMUR, 1e-4, 3, 0.2), (ga.BLOOM_BLOCKED, 0.01, 1, 0.2)])
def test_filter_text_mode_matches_oracle(sdb, kind, fpp, min_pos, ratio):
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:5])]
    ob, gb = _filters(keys, kind, fpp)
    seq, off = _reads(sdb, 3000)
    for nl in (b"\n", b"\r\n"):
        text = b"".join(_fastq(seq, off, nl))
        p = orc.parse_fastq(text, k=31)  # what the reference parser makes of it ('\r' stays in the read)
        pseq = p["seq"] if len(p["seq"]) else np.zeros(1, dtype=np.uint8)
        want = ob.filter_batch(31, min_pos, ratio, pseq, p["seq_off"], threads=4)
        f = ga.FastqBloomFilter(31, gb, min_pos, ratio)
        acc = np.full(3000, 7, dtype=np.uint8)
        nls = np.zeros(12000, dtype=np.uint32)
        f.submit_text(text, acc, newlines=nls)
        f.sync()
        assert f.text_status()[0] == -1
        assert np.array_equal(acc, want), np.flatnonzero(acc != want)[:10]
        assert np.array_equal(nls, np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10).astype(np.uint32))
        f.text_reset(True)
    gb.close()


def test_filter_text_mode_refusal(sdb):
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:5])]
    ob, gb = _filters(keys, ga.BLOOM_XOR, 1e-8)
    seq, off = _reads(sdb, 40)
    recs = _fastq(seq, off)
    bad = b"".join(recs[:3]) + b"@x\nACGT\nACGT\n+\nIIIIIIII\n@y\nAC\n+\n"  # multi-line record: 12 + 8 lines
    f = ga.FastqBloomFilter(31, gb, 1, 0.2)
    acc = np.full(5, 9, dtype=np.uint8)
    f.submit_text(bad, acc)
    f.sync()
    failed, first_bad, tot = f.text_status()
    assert failed == 0 and first_bad == 3 and tot == (0, 0, 0)
    assert not acc.any()  # a refused chunk never reports stale flags
    f.text_reset()
    acc2 = np.zeros(40, dtype=np.uint8)
    f.submit_text(b"".join(recs), acc2)
    f.sync()
    assert f.text_status()[0] == -1
    p = orc.parse_fastq(b"".join(recs), k=31)
    assert np.array_equal(acc2, ob.filter_batch(31, 1, 0.2, p["seq"], p["seq_off"], threads=1))
    gb.close()
