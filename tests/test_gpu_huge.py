"""Reads of tens of thousands of k-mer positions and more -- assembled contigs, chromosomes in a FASTA; the reference takes them like
any other read after growing its buffer (C/fastq/AbstractFastqReader.java:593-604) -- go over MANY waves (gs_match_huge_kernel: chunks
of whole iterations; gs_match_huge_finish_kernel: seams, distinct tax ids in order of first appearance, classification).  The result must
be what matchRead (C/match/FastqKMerMatcher.java:330-531) computes in one pass over the read: table, class and flags equal the
oracle's, read by read."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc
from test_gpu_match import _assert_same, _both

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _chimera(g0, rng, L, lo=40, hi=900, n_frac=0.0):
    parts, left = [], L
    while left > 0:
        s = int(rng.integers(0, g0.shape[0]))
        n = min(left, int(rng.integers(lo, hi)))
        p = int(rng.integers(0, g0.shape[1] - n))
        parts.append(g0[s][p:p + n].tobytes())
        left -= n
    r = bytearray(b"".join(parts))
    for _ in range(int(L * n_frac)):
        r[int(rng.integers(0, L))] = ord("N")
    return bytes(r)


@pytest.mark.parametrize("chunk", ["128", "384"])
def test_many_reads_over_small_chunks(sdb, monkeypatch, chunk):
    """the hand-over pulled down to 300 positions and the chunks to one and three iterations: 6000 reads of 20 .. 6000 bases go every way
    there is -- one iteration, the one-wave long path, the chunked path (its slots run out: the rest stays on one wave) --, with seams
    every 128 positions inside runs, at their ends, next to N windows"""
    monkeypatch.setenv("GS_HUGE_MIN", "300")
    monkeypatch.setenv("GS_HUGE_CHUNK", chunk)
    rng = np.random.default_rng(int(chunk))
    g0 = sdb.genomes
    reads = []
    for i in range(6000):
        L = int(rng.choice([int(rng.integers(20, 330)), 330, 331, 458, 459, int(rng.integers(331, 1500)), int(rng.integers(1500, 6000))],
                           p=[0.5, 0.01, 0.01, 0.01, 0.01, 0.4, 0.06]))
        reads.append(_chimera(g0, rng, L, n_frac=0.002 if i % 5 == 0 else 0.0))
    reads.append(bytes(g0[3][:9000]))            # one species end to end: chunks that are one run
    reads.append(b"N" * 700 + bytes(g0[1][:400]))  # chunks without a k-mer
    reads.append(b"ACGT" * 500)                   # nothing found
    seq, off = orc.pack_reads(reads)
    for cfg in (dict(), dict(threshold=4, max_paths=3), dict(classify=False), dict(max_read_tax_err=0.3, max_read_class_err=0.6)):
        o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, (seq, off), first_read_no=17, **cfg)
        _assert_same(o, g)
    # several submits of one run: the rows are clean again after every batch
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    ocv, ofl = orun.submit(seq, off)
    ot, _ = orun.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    at = 0
    for cut in (150, 151, 2000, len(reads)):
        cv, fl = m.match_reads(seq[int(off[at]):int(off[cut])], off[at:cut + 1] - off[at], at)
        assert np.array_equal(cv, ocv[at:cut]) and np.array_equal(fl, ofl[at:cut])
        at = cut
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    m.close()
    store.close()


def test_counters_in_device_memory(sdb, monkeypatch):
    """more value indices than the LDS variant takes: contigs and votes through global atomics, the tree from device memory"""
    monkeypatch.setenv("GS_HUGE_MIN", "300")
    monkeypatch.setenv("GS_HUGE_CHUNK", "256")
    nv = 3000
    parent = np.full(nv, 0, dtype=np.int32)
    parent[0] = -1
    parent[1:sdb.n_values] = sdb.parent_vi[1:]
    vidx = sdb.value_idx.copy()
    vidx[::7] = 300 + (np.arange(len(vidx[::7])) % 2690)  # hundreds of distinct tax ids per read: the 64 per chunk run out too
    rng = np.random.default_rng(4)
    reads = [_chimera(sdb.genomes, rng, int(rng.integers(200, 9000)), n_frac=0.001) for _ in range(900)]
    o, g = _both(31, sdb.kmers, vidx, nv, parent, reads, first_read_no=3)
    _assert_same(o, g)


def test_megabase_records(sdb):
    """the sizes as they are (hand-over at 32768 positions, chunks of 1024 and more): records of 30 kbp .. 9 Mbp between short reads, the
    longest one with more than 8192 x 1024 positions (its chunks grow instead of their number)"""
    rng = np.random.default_rng(23)
    g0 = sdb.genomes
    reads = [_chimera(g0, rng, 150) for _ in range(500)]
    for L in (32797, 32798, 70001, 400000, 1500000, 9000000):
        reads.insert(int(rng.integers(0, len(reads))), _chimera(g0, rng, L, lo=40, hi=9000, n_frac=2e-5))
    reads.append(bytes(np.tile(g0[5], 30)))  # 600 kbp of one species
    for cfg in (dict(), dict(threshold=50)):
        o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, reads, **cfg)
        _assert_same(o, g)
    assert (o[2] & ga.F_FOUND).sum() > 400


def _gpu_segments(m, reads):
    seq, off = orc.pack_reads(reads)
    seg_off, codes, starts, lens = m.segments(seq, off)
    return [list(zip(codes[int(a):int(b)].tolist(), lens[int(a):int(b)].tolist())) for a, b in zip(seg_off[:-1], seg_off[1:])]


@pytest.mark.parametrize("huge_min", ["300", None])
def test_kraken_style_segments_of_long_records(sdb, monkeypatch, huge_min):
    """printKrakenStyleOut (C/match/FastqKMerMatcher.java:597-611) for records that are cut into pieces of 32 iterations: the runs are
    those of one pass over the record -- a run that crosses a seam is one run, one that ends at a seam is two"""
    if huge_min:
        monkeypatch.setenv("GS_HUGE_MIN", huge_min)
    rng = np.random.default_rng(31)
    g0 = sdb.genomes
    reads = [_chimera(g0, rng, int(rng.integers(20, 20000)), n_frac=0.0005) for _ in range(300)]
    reads.append(bytes(np.tile(g0[5], 3)))                                 # runs across many seams
    reads.append(bytes(g0[2][:4096 + 30]) + bytes(g0[7][:4096]))            # a change exactly at a seam
    reads.append(_chimera(g0, rng, 3_000_000, lo=40, hi=9000, n_frac=2e-5))
    reads.append(b"ACGT" * 20000)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    got = _gpu_segments(m, reads)
    for i, r in enumerate(reads):
        want = odb.segments(r, cap=1 << 20)
        assert got[i] == want, (i, len(r), got[i][:6], want[:6])
        assert sum(n for _, n in got[i]) == len(r) - 30
    m.close()
    store.close()


@pytest.mark.parametrize("gz", [False, True])
def test_chromosome_sized_record_in_a_fasta_file(sdb, tmp_path, gz):
    """gs_host_match_files on a FASTA with a 3 Mbp record between contigs (60 bases per line; the record is longer than the pipeline's
    blocks): table, totals and the Kraken-style lines are the oracle's"""
    import gzip
    from genestrip_amd import host
    rng = np.random.default_rng(41)
    g0 = sdb.genomes
    recs = [_chimera(g0, rng, int(rng.integers(100, 3000))) for _ in range(40)]
    recs.insert(17, _chimera(g0, rng, 3_000_000, lo=40, hi=9000, n_frac=2e-5))
    recs.append(_chimera(g0, rng, 200_000, lo=40, hi=9000))
    text = b"".join(b">rec%d len=%d\n" % (i, len(r)) + b"\n".join(r[j:j + 60] for j in range(0, len(r), 60)) + b"\n" for i, r in enumerate(recs))
    p = str(tmp_path / ("asm.fa.gz" if gz else "asm.fasta"))
    with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as f:
        f.write(text)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    seq, off = orc.pack_reads(recs)
    ocv, _ = orun.submit(seq, off)
    want, _ = orun.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    kr = str(tmp_path / "kraken.out")
    table, _, tot = host.match_files(store, [p], kraken_out_path=kr, taxids=sdb.taxids)
    assert np.array_equal(table, want) and tot.reads == len(recs) and tot.bps == sum(map(len, recs))
    names = {-1: "0", -2: "A"}
    lines = []
    for i, r in enumerate(recs):
        segs = odb.segments(r, cap=1 << 20)
        segtxt = " ".join(f"{sdb.taxids[c] if c >= 0 else names[c]}:{n}" for c, n in segs)
        lines.append(f"{'C' if ocv[i] >= 0 else 'U'}\trec{i}\t{sdb.taxids[ocv[i]] if ocv[i] >= 0 else '0'}\t{len(r)}\t{segtxt}")
    assert open(kr).read().rstrip("\n").split("\n") == lines
    store.close()
