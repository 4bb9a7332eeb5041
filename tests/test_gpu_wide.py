"""Reads of 129 .. 256 k-mer positions go through gs_match_wide_kernel: the read's three or four sub-rounds of 64 positions in ONE
trip instead of two iterations of the long-read kernel -- a whole batch when its reads are of one length (gs_match_submit_fixed),
else the reads gs_match_kernel puts into the two queues.  Same contract as every match path: table, class and flags
equal matchRead's (C/match/FastqKMerMatcher.java:330-531) as the oracle restates it."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc
from test_gpu_huge import _chimera

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _fixed(store_args, seq, L, n, first=0, **cfg):
    k, kmers, vidx, nv, parent = store_args
    off = np.arange(n + 1, dtype=np.uint64) * L
    orun = orc.MatchRun(orc.DB(k, kmers, vidx, nv, parent), **cfg)
    ocv, ofl = orun.submit(seq, off, first)
    ot, _ = orun.finish()
    store = ga.DeviceKMerStore(k, kmers, vidx, nv, parent)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    cv, fl = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.uint8)
    m.submit_fixed(seq, L, n, first, class_vi=cv, flags=fl)
    gt, _ = m.finish()
    m.close()
    store.close()
    bad = np.argwhere(ot != gt)
    assert bad.size == 0, (L, cfg, bad[:8].tolist())
    assert np.array_equal(cv, ocv), (L, cfg, np.flatnonzero(cv != ocv)[:8])
    assert np.array_equal(fl, ofl), (L, cfg, np.flatnonzero(fl != ofl)[:8])
    return ofl


@pytest.mark.parametrize("L", [159, 160, 190, 221, 222, 223, 250, 285, 286, 287])
def test_reads_of_one_length_in_one_trip(sdb, L):
    rng = np.random.default_rng(L)
    g0 = sdb.genomes
    n = 5000
    reads = []
    for i in range(n):
        kind = i % 4
        if kind == 0:    # one species
            s, p = int(rng.integers(0, g0.shape[0])), int(rng.integers(0, g0.shape[1] - L))
            r = bytearray(g0[s][p:p + L].tobytes())
        elif kind == 1:  # chimeras of short pieces: many runs, several tax ids, ties
            r = bytearray(_chimera(g0, rng, L, lo=31, hi=70))
        elif kind == 2:  # two or three pieces
            r = bytearray(_chimera(g0, rng, L, lo=40, hi=150))
        else:            # nothing from the store
            r = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L).tobytes())
        if i % 7 == 0:
            for _ in range(int(rng.integers(1, 4))):
                r[int(rng.integers(0, L))] = ord("N") if rng.random() < 0.6 else ord("a")
        reads.append(bytes(r))
    seq = np.frombuffer(b"".join(reads), dtype=np.uint8)
    args = (31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    for cfg in (dict(), dict(threshold=4, max_paths=3), dict(classify=False), dict(max_read_tax_err=0.3, max_read_class_err=0.6), dict(count_unique=False, max_paths=1)):
        ofl = _fixed(args, seq, L, n, first=11, **cfg)
    assert (ofl & ga.F_FOUND).sum() > 3000


def test_counters_in_device_memory_and_many_tax_ids(sdb):
    nv = 3000
    parent = np.full(nv, 0, dtype=np.int32)
    parent[0] = -1
    parent[1:sdb.n_values] = sdb.parent_vi[1:]
    vidx = sdb.value_idx.copy()
    vidx[::3] = 300 + (np.arange(len(vidx[::3])) % 2690)  # up to ~ 60 distinct tax ids per read, more than 64 in some
    rng = np.random.default_rng(9)
    L, n = 222, 3000
    seq = np.frombuffer(b"".join(_chimera(sdb.genomes, rng, L, lo=31, hi=400) for _ in range(n)), dtype=np.uint8)
    _fixed((31, sdb.kmers, vidx, nv, parent), seq, L, n)
    _fixed((31, sdb.kmers, vidx, nv, parent), seq, L, n, threshold=3)


def test_mixed_lengths_through_the_queues(sdb):
    """an offsets array with reads of 20 .. 400 bases in random order: one iteration, the two wide queues, the long-read queue"""
    rng = np.random.default_rng(3)
    g0 = sdb.genomes
    reads = []
    for i in range(12000):
        L = int(rng.choice([int(rng.integers(20, 158)), 158, 159, 222, 223, 286, 287, int(rng.integers(159, 287)), int(rng.integers(287, 400))],
                           p=[0.2, 0.03, 0.03, 0.03, 0.03, 0.03, 0.03, 0.5, 0.12]))
        r = bytearray(_chimera(g0, rng, L, lo=31, hi=int(rng.choice([70, 300]))))
        if i % 6 == 0:
            r[int(rng.integers(0, L))] = ord("N")
        reads.append(bytes(r))
    seq, off = orc.pack_reads(reads)
    from test_gpu_match import _assert_same, _both
    for cfg in (dict(), dict(threshold=4, max_paths=3), dict(classify=False)):
        o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, (seq, off), first_read_no=5, **cfg)
        _assert_same(o, g)
    # more classification paths than the wide kernels hold: everything above 128 positions on the long-read path again
    o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, (seq, off), max_paths=100)
    _assert_same(o, g)


@pytest.mark.parametrize("k", [16, 21, 25])
def test_other_k(k):
    """k < 19: a store without records (the ordinary table); 150-bp reads have 135 / 130 / 126 positions"""
    db = synth.SynthDB(k=k, genera=2, species_per_genus=3, genome_len=8000, seed=k)
    rng = np.random.default_rng(k)
    for L in (150, 180, 230):
        n = 2500
        seq = np.frombuffer(b"".join(_chimera(db.genomes, rng, L, lo=k, hi=int(rng.choice([60, 300])), n_frac=0.002) for _ in range(n)), dtype=np.uint8)
        _fixed((k, db.kmers, db.value_idx, db.n_values, db.parent_vi), seq, L, n)


@pytest.mark.parametrize("gz", [False, True])
def test_paired_end_lengths_from_a_fastq_file(sdb, tmp_path, gz):
    """2 x 250-like reads (trimmed: 120 .. 251 bases) in a FASTQ file through gs_host_match_files: the device's record scan hands
    (start, end) pairs to the match kernel, which spreads the reads over its queues; table and totals are the oracle's"""
    import gzip
    from genestrip_amd import host
    rng = np.random.default_rng(8)
    reads = [_chimera(sdb.genomes, rng, int(rng.integers(120, 252)), lo=60, hi=400, n_frac=0.001) for _ in range(20000)]
    text = b"".join(b"@m%d/1\n%s\n+\n%s\n" % (i, r, b"F" * len(r)) for i, r in enumerate(reads))
    p = str(tmp_path / ("r.fastq.gz" if gz else "r.fastq"))
    with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as f:
        f.write(text)
    seq, off = orc.pack_reads(reads)
    orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    orun.submit(seq, off, threads=4, per_read=False)
    want, _ = orun.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    table, _, tot = host.match_files(store, [p])
    assert np.array_equal(table, want) and tot.reads == len(reads) and tot.bps == sum(map(len, reads))
    store.close()
